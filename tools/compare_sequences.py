"""Two step sequences (tools/step_sequence.py output) side by side, per kernel name: python tools/compare_sequences.py a.txt b.txt
Boxes differ by a few percent in clock: the ratio of families that did not change between the two builds (e.g. the
weight-gradient kernels) tells how much of a difference is the box."""
import re
import sys


def load(f):
    d = {}
    for ln in open(f):
        m = re.match(r'\s*(\d+)\s+([\d.]+) g=\s*(\d+) (.*)', ln)
        if m:
            k = re.split(r'[<(]', m.group(4))[0].strip()
            v = d.setdefault(k, [0.0, 0])
            v[0] += float(m.group(2))
            v[1] += 1
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
ta, tb = sum(v[0] for v in a.values()), sum(v[0] for v in b.values())
for k in sorted(set(a) | set(b), key=lambda k: -max(a.get(k, [0])[0], b.get(k, [0])[0])):
    x, y = a.get(k, [0.0, 0]), b.get(k, [0.0, 0])
    if max(x[0], y[0]) >= 4.0:
        print(f'{k[:44]:44s} {x[0]:8.1f} ({x[1]:3d})  {y[0]:8.1f} ({y[1]:3d})  {y[0] - x[0]:+8.1f}')
print(f'total {ta:.1f} us ({sum(v[1] for v in a.values())} kernels) -> {tb:.1f} us ({sum(v[1] for v in b.values())} kernels)')
