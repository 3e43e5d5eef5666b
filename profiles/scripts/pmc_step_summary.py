"""Whole-step HBM traffic per kernel family from two rocprofv3 PMC passes of `bench.py --steps 3 --warmup 2`
(profiles/scripts/pmc_step.sh).  usage: pmc_step_summary.py <out_dir> <json_out>  (markdown on stdout)

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB; FETCH_SIZE
reports half of the bytes of wide coalesced reads on gfx950 and is doubled, WRITE_SIZE is exact.  The correction is
calibrated for 16-byte-per-lane streaming accesses, which is what every heavy kernel of this step issues; narrow-access
kernels (index / scan kernels) are a negligible share of the bytes.  Infinity-Cache hits are counted as traffic (the
counters sit on the L2's memory side), so `traffic` here is an upper bound of the bytes that reached HBM.

The 3 timed steps are the dispatches between the (2*WARM+1)-th and the (2*(WARM+STEPS)+1)-th `vox_key_kernel` launch
(every step voxelises two frames), as in summarize_trace.py; durations come from the kernel trace of the same run."""
import collections, csv, glob, json, re, sys

out_dir, json_out = sys.argv[1], sys.argv[2]
WARM, STEPS = 2, 3


def family(n):
    if n.startswith('wgrad') or n.startswith('dense_wgrad'): return 'weight gradients (wgrad*)'
    if n.startswith('token_gemm'): return 'token GEMMs'
    if n.startswith('win_attn') or n.startswith('dtau') or n.startswith('win_worklist') or n.startswith('win_class'): return 'window attention'
    if 'igemm' in n or n.startswith('dense_conv3x3'): return 'implicit-GEMM convolutions'
    if n.startswith('ln_') or n.startswith('add_ln'): return 'LayerNorm'
    if n.startswith('bn_'): return 'BatchNorm'
    if n.startswith('Cijk'): return 'hipBLASLt GEMMs'
    if n.startswith('deblock') or n.startswith('colsum') or n.startswith('to_dense') or n.startswith('dense_gather'): return 'decoder head (dense <-> sparse)'
    if any(n.startswith(p) for p in ('vox_', 'csr_', 'segmax', 'voxel_mean', 'point_feat', 'mask_', 'group_points', 'chamfer', 'scan_', 'win_count', 'win_level', 'win_emit', 'window_cells', 'nbr_', 'index_grid', 'spconv_')): return 'voxelise / VFE / mask / windows / Chamfer'
    if n.startswith('adam') or n.startswith('multi_cast') or n.startswith('bn_running'): return 'optimizer / parameter copies'
    return 'torch elementwise / copies / fills'


def load(counter):
    f = glob.glob(f'{out_dir}/pmc_{counter}/**/*counter_collection.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows


def clean(name):
    n = re.sub(r'^void ', '', name)
    n = n.replace('at::native::', '').replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*', '', n)[:90]


res = {}
per_counter = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = load(c)
    naive = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('naive_conv')]
    rows = rows[(max(naive) + 1) if naive else 0:]
    vk = [i for i, r in enumerate(rows) if 'vox_key_kernel' in r['Kernel_Name']]
    seg = [r for r in rows[vk[2 * WARM]:vk[2 * (WARM + STEPS)]] if 'probe_mfma_kernel' not in r['Kernel_Name'] and 'probe_copy_kernel' not in r['Kernel_Name']]
    agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for r in seg:
        n = clean(r['Kernel_Name'])
        agg[n][0] += float(r['Counter_Value']) * 1024.0
        agg[n][1] += 1
        if 'Start_Timestamp' in r and 'End_Timestamp' in r:
            agg[n][2] += (int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    per_counter[c] = agg
names = sorted(set(per_counter['FETCH_SIZE']) | set(per_counter['WRITE_SIZE']))
fam = collections.defaultdict(lambda: dict(fetch=0.0, write=0.0, ns=0.0, launches=0))
kern = {}
for n in names:
    f, w = per_counter['FETCH_SIZE'].get(n, [0, 0, 0]), per_counter['WRITE_SIZE'].get(n, [0, 0, 0])
    fetch, write = 2.0 * f[0] / STEPS, w[0] / STEPS
    ns = (f[2] or w[2]) / STEPS
    kern[n] = dict(fetch_bytes=fetch, write_bytes=write, ms=ns / 1e6, launches=f[1] / STEPS)
    d = fam[family(n)]
    d['fetch'] += fetch; d['write'] += write; d['ns'] += ns; d['launches'] += f[1] / STEPS
tot_f = sum(d['fetch'] for d in fam.values())
tot_w = sum(d['write'] for d in fam.values())
tot_ns = sum(d['ns'] for d in fam.values())
out = {'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 '
                  '--warmup 2 --no-cpu-baseline --no-secondary (one pass per counter; the 3 timed steps)',
       'unit_note': 'KB counters; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are counted',
       'step_hbm_bytes': tot_f + tot_w, 'step_fetch_bytes': tot_f, 'step_write_bytes': tot_w,
       'kernel_ms_per_step_under_pmc': tot_ns / 1e6,
       'families': {k: dict(fetch_bytes=v['fetch'], write_bytes=v['write'], bytes=v['fetch'] + v['write'], ms=v['ns'] / 1e6,
                            gbs=(v['fetch'] + v['write']) / max(v['ns'], 1.0), launches=v['launches']) for k, v in fam.items()},
       'kernels': kern}
json.dump(out, open(json_out, 'w'), indent=1)
print(f'whole step: {(tot_f + tot_w) / 1e9:.2f} GB per step ({tot_f / 1e9:.2f} read + {tot_w / 1e9:.2f} written), '
      f'{tot_ns / 1e6:.1f} ms of kernels under the counter pass -> {(tot_f + tot_w) / max(tot_ns, 1):.2f} GB/s average\n')
print('| family | GB/step | read | written | ms/step | GB/s | launches |\n|---|---:|---:|---:|---:|---:|---:|')
for k, v in sorted(fam.items(), key=lambda kv: -(kv[1]['fetch'] + kv[1]['write'])):
    b = v['fetch'] + v['write']
    print(f"| {k} | {b / 1e9:.2f} | {v['fetch'] / 1e9:.2f} | {v['write'] / 1e9:.2f} | {v['ns'] / 1e6:.2f} | {b / max(v['ns'], 1):.0f} | {v['launches']:.0f} |")
print('\n| kernel | MB/launch read | MB/launch written | us/launch | GB/s | launches/step |\n|---|---:|---:|---:|---:|---:|')
for n, v in sorted(kern.items(), key=lambda kv: -(kv[1]['fetch_bytes'] + kv[1]['write_bytes']))[:60]:
    l = max(v['launches'], 1e-9)
    b = v['fetch_bytes'] + v['write_bytes']
    print(f"| `{n}` | {v['fetch_bytes'] / l / 1e6:.1f} | {v['write_bytes'] / l / 1e6:.1f} | {v['ms'] * 1e3 / l:.1f} | {b / max(v['ms'] * 1e6, 1):.0f} | {v['launches']:.1f} |")
