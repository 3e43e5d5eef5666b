"""HBM traffic of the roofline kernels from two rocprofv3 PMC passes of `bench.py --probe-hbm-only` (round 6; --probe-only before).

  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d <out>/pmc_$c -- python3 bench.py --probe-hbm-only
  done
  python profiles/scripts/pmc_summary.py <out> > profiles/<round>_pmc.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB;
FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 and is doubled; WRITE_SIZE is exact.
An op = one launch of every kernel of its group (wgrad: the kernel + two slab reductions; attention: the three
tile-class launches NT = 1, 2, 4 of the stage-1 backward; token GEMMs: their one launch); bytes per op = sum of the per-launch
averages.  The training step itself is not run by the probe command, so every launch of these kernels belongs to a probe -- except the
token GEMMs, which also run in the forward pass that measures the token count: their probe is the LAST 23 launches of the instance.

Kernels are matched by base name + identifying template arguments (profiles/scripts/_kernels.py); a group that matches NO kernel
aborts the script (round 4 committed a 0 for the priced kernel because a seventh template argument had made a whole-string
comparison miss)."""
import collections, csv, glob, hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _kernels as K

out_dir = sys.argv[1]
groups = {'token_gemm_gelu': (K.is_tgw_gelu,),              # `roofline`: the dual-store FFN-1 instance, the family's heaviest in the step
          'token_gemm': (K.is_tgw_plain,),                  # `roofline_token_gemm_plain` (the priced kernel of rounds 3-4)
          'wgrad': (K.prefix('wgrad256_kernel'), K.prefix('wgrad_reduce_kernel')),
          'attention': (K.prefix('win_attn_bwd_mfma_kernel<16'),)}
LAST_N = {'token_gemm': 23, 'token_gemm_gelu': 23}        # 3 warm-up + 20 timed launches of the probe
raw = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(f'{out_dir}/pmc_{c}/**/*counter_collection.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == c]
    for r in rows:
        r['_name'] = K.clean(r['Kernel_Name'])
    drop = set()
    for g, nlast in LAST_N.items():
        ids = sorted(int(r['Dispatch_Id']) for r in rows if any(m(r['_name']) for m in groups[g]))
        drop |= set(ids[:-nlast])
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        if int(r['Dispatch_Id']) in drop:
            continue
        agg[r['_name']][0] += float(r['Counter_Value'])
        agg[r['_name']][1] += 1
    raw[c] = {k: v[0] / v[1] for k, v in agg.items()}
res = {'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py '
                  '--probe-hbm-only (one pass per counter)',
       'unit_note': 'counter unit = KB; gfx950: FETCH_SIZE counts half of the bytes of wide coalesced reads -> doubled; '
                    'WRITE_SIZE exact (MI355X_MICROARCH.md, HBM section)'}
for g, matchers in groups.items():
    fk = {k: round(v, 1) for k, v in sorted(raw['FETCH_SIZE'].items()) if any(m(k) for m in matchers)}
    wk = {k: round(v, 1) for k, v in sorted(raw['WRITE_SIZE'].items()) if any(m(k) for m in matchers)}
    K.require(g + ' / FETCH_SIZE', fk)
    K.require(g + ' / WRITE_SIZE', wk)
    fetch_raw, write = int(sum(fk.values()) * 1024), int(sum(wk.values()) * 1024)
    if 2 * fetch_raw + write <= 0:
        sys.exit(f'pmc_summary: group "{g}" summed to 0 bytes')
    res[g] = {'FETCH_SIZE_KB_per_launch': fk, 'WRITE_SIZE_KB_per_launch': wk, 'fetch_bytes_raw': fetch_raw,
              'fetch_bytes_corrected': 2 * fetch_raw, 'write_bytes': write,
              'traffic_bytes_per_op': 2 * fetch_raw + write}
# what the priced kernels' sources looked like when the counters were read: bench.py compares these with the tree it runs from
_csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 't-mae_amd', 'csrc')
res['source_sha16'] = {f: hashlib.sha256(open(os.path.join(_csrc, f), 'rb').read()).hexdigest()[:16]
                       for f in ('token_gemm_wreg.hip', 'wgrad.hip', 'attention_mfma.hip')}
print(json.dumps(res, indent=1))
