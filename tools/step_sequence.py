"""Kernel sequence of ONE graph-replayed train step from a rocprofv3 kernel trace (steps are delimited by the Adam kernel):
python tools/step_sequence.py <kernel_trace.csv> [step_index]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 14
seg = rows[idx[k] + 1:idx[k + 1] + 1]


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'at::native::', '', n)[:110]


for i, r in enumerate(seg):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f'{i:3d} {d:7.1f} g={r["Grid_Size_X"]:>8s} {short(r["Kernel_Name"])}')
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3
print(f'{len(seg)} kernels, span {(t1 - t0) / 1e3:.1f} us, busy {busy:.1f} us')
