"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py --steps 3 --warmup 2`: per-step kernel time of the timed steps.
usage: python profiles/scripts/summarize_trace.py <kernel_trace.csv> [top_n]"""
import collections, csv, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _kernels as K
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
naive = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('naive_conv')]
rest = rows[(max(naive) + 1) if naive else 0:]
vk = [i for i, r in enumerate(rest) if 'vox_key_kernel' in r['Kernel_Name']]
WARM, STEPS = 2, 3                  # bench.py --warmup 2 --steps 3: every step voxelises two frames (2 vox_key launches)
seg = rest[vk[2 * WARM]:vk[2 * (WARM + STEPS)]]      # the 3 timed steps; what follows (FLOP-model forward, probes) is the tail
# box-peak probes (csrc/probe.hip) are never step kernels, wherever a command line places them (the fine-tune profile of round 5
# counted probe_mfma_kernel / probe_copy_kernel as 7 ms "per step")
seg = [r for r in seg if 'probe_mfma_kernel' not in r['Kernel_Name'] and 'probe_copy_kernel' not in r['Kernel_Name']]
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

# idle intervals of the timed steps: where the GPU waited for the host (a host sync drains the queue; the launches behind it arrive one
# by one).  Gaps above 20 us, summed per (kernel before -> kernel after), per step.
ivs = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:50]) for r in seg)
gaps = collections.defaultdict(lambda: [0, 0])
end, last, idle_all, idle_big = ivs[0][1], ivs[0][2], 0, 0
for st, en, nm in ivs[1:]:
    if st > end:
        g = st - end
        idle_all += g
        if g > 20000:
            idle_big += g
            gaps[(last, nm)][0] += g
            gaps[(last, nm)][1] += 1
    if en > end:
        end, last = en, nm
print(f'\nidle: {idle_all / 3e6:.2f} ms/step in all gaps, {idle_big / 3e6:.2f} ms/step in gaps > 20 us:')
for (a, b), (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f'- {g / 3e6:.3f} ms/step, {c / 3:.1f} x: `{a}` -> `{b}`')

# the roofline probes of bench.py (token_gemm_roofline, wgrad_roofline, attention_roofline): the launches after the last voxelisation of
# the training steps (+1: the probe's own forward pass voxelises both frames once more); an op = one launch of each
# kernel of its group, its duration = the sum of the per-kernel averages
tail = rest[vk[2 * (WARM + STEPS)]:]
def grid_of(r):
    return r.get('Grid_Size_X', r.get('Grid_Size', ''))


for title, matchers, last in (('token GEMM op, dual-store GELU2 instance = `roofline` (one launch)', (K.is_tgw_gelu,), 23),
                             ('token GEMM op, plain instance (one launch)', (K.is_tgw_plain,), 23),
                             ('wgrad256 op (kernel + slab reduction)', (K.prefix('wgrad256_kernel'), K.prefix('wgrad_reduce_kernel')), 0),
                             ('stage-1 attention backward op (3 tile classes)', (K.prefix('win_attn_bwd_mfma_kernel<16'),), 0)):
    probe = K.require(title, [r for r in tail if any(m(r['Kernel_Name']) for m in matchers)])
    if last:      # the probe's forward pass launches this kernel too: the probe = its last 23 launches (3 warm-up + 20 timed)
        probe = probe[-last:]
    pa = collections.defaultdict(lambda: [0, 0])
    for r in probe:
        n = K.clean(r['Kernel_Name'])
        pa[n][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        pa[n][1] += 1
    print(f'\nroofline probe, {title}, per launch:')
    tot = 0.0
    for n, (d, c) in sorted(pa.items()):
        print(f'- `{n}`: {c} launches, avg {d / c / 1e3:.1f} us')
        tot += d / c / 1e3
    print(f'- op: {tot:.1f} us')
