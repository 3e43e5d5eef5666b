"""HBM traffic of the roofline kernels from two rocprofv3 PMC passes of `bench.py --probe-only`.

  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d <out>/pmc_$c -- python3 bench.py --probe-only
  done
  python profiles/scripts/pmc_summary.py <out> > profiles/<round>_pmc.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB;
FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 and is doubled; WRITE_SIZE is exact.
An op = one launch of every kernel of its group (wgrad: the kernel + two slab reductions; attention: the three
tile-class launches NT = 1, 2, 4 of the stage-1 backward; token GEMM: its one launch); bytes per op = sum of the per-launch averages.
The training step itself is not run by --probe-only, so every launch of these kernels belongs to a probe."""
import collections, csv, glob, json, sys

out_dir = sys.argv[1]
TG = 'token_gemm_wreg_kernel<256, 4, 8, false, false, false>'      # round 4 (round 3: five template arguments; rounds 1-2: 'token_gemm_res_kernel<256, 4, false>')
groups = {'token_gemm': (TG,),
          'wgrad': ('wgrad256_kernel', 'wgrad_reduce_kernel'),
          'attention': ('win_attn_bwd_mfma_kernel<16',)}
# the token GEMM also runs in the forward pass that measures the token count (same grid: the kernel is persistent, one
# workgroup per CU): the probe's 3 warm-up + 20 timed launches are the LAST 23 dispatches of that kernel in the process
LAST_N = {TG: 23}
raw = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(f'{out_dir}/pmc_{c}/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == c]
    for r in rows:
        r['_name'] = r['Kernel_Name'].split('(')[0].replace('void ', '')
    keep = {}
    for k, nlast in LAST_N.items():
        ids = sorted(int(r['Dispatch_Id']) for r in rows if r['_name'] == k)
        keep[k] = set(ids[-nlast:])
    for r in rows:
        name = r['_name']
        if name in keep and int(r['Dispatch_Id']) not in keep[name]:
            continue
        agg[name][0] += float(r['Counter_Value'])
        agg[name][1] += 1
    raw[c] = {k: v[0] / v[1] for k, v in agg.items()}
res = {'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py '
                  '--probe-only (one pass per counter)',
       'unit_note': 'counter unit = KB; gfx950: FETCH_SIZE counts half of the bytes of wide coalesced reads -> doubled; '
                    'WRITE_SIZE exact (MI355X_MICROARCH.md, HBM section)'}
for g, pats in groups.items():
    fk = {k: round(v, 1) for k, v in sorted(raw['FETCH_SIZE'].items()) if any(p in k for p in pats)}
    wk = {k: round(v, 1) for k, v in sorted(raw['WRITE_SIZE'].items()) if any(p in k for p in pats)}
    fetch_raw, write = int(sum(fk.values()) * 1024), int(sum(wk.values()) * 1024)
    res[g] = {'FETCH_SIZE_KB_per_launch': fk, 'WRITE_SIZE_KB_per_launch': wk, 'fetch_bytes_raw': fetch_raw,
              'fetch_bytes_corrected': 2 * fetch_raw, 'write_bytes': write,
              'traffic_bytes_per_op': 2 * fetch_raw + write}
# what the priced kernels' sources looked like when the counters were read: bench.py compares these with the tree it runs from
import hashlib, os
_csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd', 'csrc')
res['source_sha16'] = {f: hashlib.sha256(open(os.path.join(_csrc, f), 'rb').read()).hexdigest()[:16]
                       for f in ('token_gemm_wreg.hip', 'wgrad.hip', 'attention_mfma.hip')}
print(json.dumps(res, indent=1))
