"""Timeline of ONE graph-replayed train step from a rocprofv3 kernel trace (steps delimited by the Adam kernel): start offset,
duration, queue, and how much of each kernel ran beside another one.  python tools/step_timeline.py <kernel_trace.csv> [step_index]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 14
seg = rows[idx[k] + 1:idx[k + 1] + 1]
qk = 'Queue_Id' if 'Queue_Id' in seg[0] else None
t0 = int(seg[0]['Start_Timestamp'])
iv = [(int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0) for r in seg]


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'at::native::', '', n)[:70]


qs = {}
cover = 0
last_end = 0
for i, (r, (s, e)) in enumerate(zip(seg, iv)):
    q = r[qk] if qk else '?'
    qs.setdefault(q, len(qs))
    ov = sum(max(0, min(e, e2) - max(s, s2)) for j, (s2, e2) in enumerate(iv) if j != i)
    gap = s - last_end
    print(f'{i:3d} q{qs[q]} +{s / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  gap {gap / 1e3:6.1f}  ov {ov / 1e3:6.1f}  {short(r["Kernel_Name"])}')
    if e > last_end:
        cover += e - max(s, last_end)
        last_end = e
span = max(e for _, e in iv)
busy = sum(e - s for s, e in iv)
print(f'{len(seg)} kernels on {len(qs)} queues, span {span / 1e3:.1f} us, covered {cover / 1e3:.1f} us (idle {(span - cover) / 1e3:.1f}), sum of durations {busy / 1e3:.1f} us')
