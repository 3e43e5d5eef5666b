"""HBM traffic of the roofline kernel from two rocprofv3 PMC passes of `bench.py --probe-only`.

  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d <out>/pmc_$c -- python3 bench.py --probe-only
  done
  python profiles/scripts/pmc_summary.py <out> <algorithmic_bytes_per_op> > profiles/<round>_attn_bwd_pmc.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB;
FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950 and is doubled; WRITE_SIZE is exact.
The op = the three tile-class launches (NT = 1, 2, 4) of win_attn_bwd_mfma_kernel<16, NT>."""
import collections, csv, glob, json, sys

out_dir, alg = sys.argv[1], int(sys.argv[2])
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(f'{out_dir}/pmc_{c}/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'win_attn_bwd_mfma_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c:
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')
            agg[name][0] += float(r['Counter_Value'])
            agg[name][1] += 1
    res[c] = {k: round(v[0] / v[1], 1) for k, v in sorted(agg.items())}
fetch_raw = int(sum(res['FETCH_SIZE'].values()) * 1024)
write = int(sum(res['WRITE_SIZE'].values()) * 1024)
traffic = 2 * fetch_raw + write
print(json.dumps({
    'kernel': 'win_attn_bwd_mfma_kernel<16,NT>, NT=1,2,4 (stage-1 self-attention backward of the previous frame, bf16)',
    'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --probe-only '
               '(one pass per counter)',
    'unit_note': 'counter unit = KB; gfx950: FETCH_SIZE counts half of the bytes of wide coalesced reads -> doubled; '
                 'WRITE_SIZE exact (MI355X_MICROARCH.md, HBM section)',
    'FETCH_SIZE_KB_per_launch': res['FETCH_SIZE'], 'WRITE_SIZE_KB_per_launch': res['WRITE_SIZE'],
    'fetch_bytes_raw': fetch_raw, 'fetch_bytes_corrected': 2 * fetch_raw, 'write_bytes': write,
    'traffic_bytes_per_op': traffic, 'algorithmic_bytes_per_op': alg,
    'traffic_over_algorithmic': round(traffic / alg, 3)}, indent=1))
