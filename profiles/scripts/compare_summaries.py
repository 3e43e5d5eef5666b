"""Per-kernel comparison of two or more kernel-trace summaries (summarize_trace.py tables): A-files against B-files.
    python3 profiles/scripts/compare_summaries.py A1.md[,A2.md] B1.md[,B2.md] [min ms/step]"""
import re
import sys


def load(paths):
    acc = {}
    for p in paths.split(','):
        for line in open(p):
            m = re.match(r'\| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| `(.*)` \|', line)
            if m:
                acc.setdefault(m[5], []).append((float(m[1]), float(m[3]), float(m[4])))
    return {k: tuple(sum(x[i] for x in v) / len(v) for i in range(3)) for k, v in acc.items()}


a, b = load(sys.argv[1]), load(sys.argv[2])
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
ta = tb = 0.0
print('| kernel | launches | A us | B us | A ms/step | B ms/step | delta ms |\n|---|---:|---:|---:|---:|---:|---:|')
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0,))[0] + b.get(k, (0,))[0])):
    xa, xb = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    ta, tb = ta + xa[0], tb + xb[0]
    if max(xa[0], xb[0]) >= floor:
        print(f'| `{k[:90]}` | {xb[1] or xa[1]:.0f} | {xa[2]:.1f} | {xb[2]:.1f} | {xa[0]:.2f} | {xb[0]:.2f} | {xb[0] - xa[0]:+.3f} |')
print(f'\nall kernels: A {ta:.2f} ms/step, B {tb:.2f} ms/step, delta {tb - ta:+.2f}')
