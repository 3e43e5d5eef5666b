#!/bin/bash
# Whole-step HBM traffic: FETCH_SIZE and WRITE_SIZE of EVERY kernel of 3 timed training steps (one rocprofv3 run per counter,
# --kernel-trace only beside --pmc), summarised per kernel family as bytes/step, GB/s and ratio to the family's algorithmic bytes.
#   bash profiles/scripts/pmc_step.sh <tag>      (on the GPU box; writes gpurun_out/<tag>_step_bytes.{json,md})
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r4}
OUT=/tmp/pmc_step_$TAG
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-vfe-prefetch > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_step_$c.log" 2>&1 )
  echo "$c rc $?"
done
python3 profiles/scripts/pmc_step_summary.py $OUT gpurun_out/${TAG}_step_bytes.json > gpurun_out/${TAG}_step_bytes.md
head -40 gpurun_out/${TAG}_step_bytes.md
