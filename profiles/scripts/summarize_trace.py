"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py --steps 3 --warmup 2`: per-step kernel time of the timed steps.
usage: python profiles/scripts/summarize_trace.py <kernel_trace.csv> [top_n]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
naive = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('naive_conv')]
rest = rows[(max(naive) + 1) if naive else 0:]
vk = [i for i, r in enumerate(rest) if 'vox_key_kernel' in r['Kernel_Name']]
seg = rest[vk[-8]:vk[-2]]          # 3 timed steps (2 voxelisations each); the last 2 launches belong to the roofline probe
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = re.sub(r'\(.*', '', r['Kernel_Name'])
    n = n.replace('void ', '').replace('at::native::', '').replace('(anonymous namespace)::', '')[:95]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[n][0] += d
    agg[n][1] += 1
busy = sum(v[0] for v in agg.values())
print(f'wall {(t1 - t0) / 3e6:.1f} ms/step, GPU busy {busy / 3e6:.1f} ms/step, {len(seg) / 3:.0f} launches/step')
print('| ms/step | % | launches/step | avg us | kernel |\n|---:|---:|---:|---:|---|')
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f'| {d / 3e6:.2f} | {100 * d / busy:.1f} | {c / 3:.1f} | {d / c / 1e3:.1f} | `{n}` |')

# the roofline probe of bench.py (kernel_roofline): the launches of the stage-1 backward kernels after the last
# voxelisation; the op = one launch of each tile class, its duration = the sum of the three averages
probe = [r for r in rest[vk[-1]:] if 'win_attn_bwd_mfma_kernel<16' in r['Kernel_Name']]
if probe:
    pa = collections.defaultdict(lambda: [0, 0])
    for r in probe:
        n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        pa[n][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        pa[n][1] += 1
    print('\nroofline probe (bench.py kernel_roofline), per launch:')
    tot = 0.0
    for n, (d, c) in sorted(pa.items()):
        print(f'- `{n}`: {c} launches, avg {d / c / 1e3:.1f} us')
        tot += d / c / 1e3
    print(f'- op (sum of the tile classes): {tot:.1f} us')
