import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
n=int(sys.argv[2]) if len(sys.argv)>2 else 22
steps=float(sys.argv[3]) if len(sys.argv)>3 else 1
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over all steps; per step {tot/1e6/steps:.2f} ms")
for r in rows[:n]:
    print(f"{r['Name'][:84]:84s} calls={r['Calls']:>6s} ms/step={float(r['TotalDurationNs'])/1e6/steps:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
