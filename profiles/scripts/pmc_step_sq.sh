#!/bin/bash
# SQ counters of EVERY kernel of 3 timed training steps (one rocprofv3 pass, --pmc only beside --kernel-trace), per kernel:
# instructions per class and busy fractions.   bash profiles/scripts/pmc_step_sq.sh <tag>   (on the GPU box)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r4}
OUT=/tmp/pmc_sq_$TAG
rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/p -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-vfe-prefetch > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_sq.log" 2>&1 )
echo "rc $?"
python3 - "$OUT" > gpurun_out/${TAG}_step_sq.md <<'PY'
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/p/**/*counter_collection.csv', recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
    per[k][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, cs in per.items():
    n = len(cs['SQ_BUSY_CYCLES'])
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    busy = m.get('SQ_BUSY_CYCLES', 0) or 1.0
    # SQ_BUSY_CYCLES sums the XCD-level busy cycles (x 8 XCDs x 4 SEs?); use ratios that cancel: per-SIMD cycles = WAVE_CYCLES based
    rows.append((m.get('SQ_INSTS_MFMA', 0) * n, k, n, m))
rows.sort(reverse=True)
print('| kernel | launches | VALU/launch | MFMA/launch | LDS/launch | VALU per MFMA | MFMA busy cyc / BUSY | VALU active (quad-cyc x4) / BUSY | wait-inst / wave-cyc |')
print('|---|---:|---:|---:|---:|---:|---:|---:|---:|')
for _, k, n, m in rows[:60]:
    mf = m.get('SQ_INSTS_MFMA', 0)
    print(f"| `{k[:70]}` | {n} | {m.get('SQ_INSTS_VALU',0):.3g} | {mf:.3g} | {m.get('SQ_INSTS_LDS',0):.3g} | {m.get('SQ_INSTS_VALU',0)/mf if mf else float('nan'):.2f} | "
          f"{m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(m.get('SQ_BUSY_CYCLES',1),1):.3f} | {4*m.get('SQ_ACTIVE_INST_VALU',0)/max(m.get('SQ_BUSY_CYCLES',1),1):.3f} | "
          f"{m.get('SQ_WAIT_INST_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.3f} |")
PY
head -30 gpurun_out/${TAG}_step_sq.md | cut -c1-260
