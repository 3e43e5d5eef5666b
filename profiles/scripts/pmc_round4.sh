#!/bin/bash
# HBM-traffic passes (FETCH_SIZE, WRITE_SIZE: one rocprofv3 run each) over bench.py --probe-only -> profiles-ready JSON.
#   bash profiles/scripts/pmc_round4.sh   (on the GPU box; writes gpurun_out/round4_pmc.json)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=/tmp/pmc_r4
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 "$GRAFT_REPO_ROOT/bench.py" --probe-only > "$GRAFT_REPO_ROOT/gpurun_out/pmc_r4_$c.log" 2>&1 )
  echo "$c rc $?"
done
python3 profiles/scripts/pmc_summary.py $OUT > gpurun_out/round4_pmc.json
head -c 600 gpurun_out/round4_pmc.json
