"""Kernel sequence of ONE graph-replayed inference pass from a rocprofv3 kernel trace (passes are delimited by the
bound_mask_apply kernel): python tools/infer_sequence.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'bound_mask_apply' in r['Kernel_Name']]
seg = rows[idx[-2] + 1:idx[-1] + 1]
for i, r in enumerate(seg):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f'{i:3d} {d:7.1f} g={r["Grid_Size_X"]:>8s} {n[:100]}')
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
print(len(seg), 'kernels, span', (t1 - t0) / 1e3, 'us')
