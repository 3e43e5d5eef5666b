#!/bin/bash
# HBM-traffic passes (FETCH_SIZE, WRITE_SIZE: one rocprofv3 run each) over bench.py --probe-hbm-only -> profiles-ready JSON.
#   bash profiles/scripts/pmc_round6.sh   (on the GPU box; writes gpurun_out/round6_pmc.json, or nothing if a group is empty)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=/tmp/pmc_r6
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 "$GRAFT_REPO_ROOT/bench.py" --probe-hbm-only > "$GRAFT_REPO_ROOT/gpurun_out/pmc_r6_$c.log" 2>&1 )
  rc=$?; echo "$c rc $rc"; [ $rc -eq 0 ] || exit $rc
done
if python3 profiles/scripts/pmc_summary.py $OUT > gpurun_out/round6_pmc.json.tmp; then
  mv gpurun_out/round6_pmc.json.tmp gpurun_out/round6_pmc.json
  python3 -c "import json; d=json.load(open('gpurun_out/round6_pmc.json')); print({k: v['traffic_bytes_per_op'] for k, v in d.items() if isinstance(v, dict) and 'traffic_bytes_per_op' in v})"
else
  echo "pmc_summary FAILED: no round6_pmc.json written"; rm -f gpurun_out/round6_pmc.json.tmp; exit 1
fi
