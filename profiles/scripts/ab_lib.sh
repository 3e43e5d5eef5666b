#!/bin/bash
# Same-box A/B of two builds of the library: bench.py (20 timed steps) with TMAE_LIB_PATH = A, B, A, B; prints ms/step of each run.
#   bash profiles/scripts/ab_lib.sh <libA.so> <libB.so> [tag] [extra bench args]      (paths relative to the repo root)
# Boxes differ by up to 3 % in step time; only runs interleaved on ONE box resolve a sub-millisecond change.
cd "$GRAFT_REPO_ROOT" || exit 1
A=$1; B=$2; TAG=${3:-ab}; shift 3
for i in 1 2; do
  for which in A B; do
    lib=$A; [ $which = B ] && lib=$B
    TMAE_LIB_PATH="$GRAFT_REPO_ROOT/$lib" timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" \
      > gpurun_out/${TAG}_${which}${i}.json 2> gpurun_out/${TAG}_${which}${i}.err || { echo "run $which$i failed"; tail -3 gpurun_out/${TAG}_${which}${i}.err; exit 1; }
    python3 -c "
import json,sys; d=json.loads(open('gpurun_out/${TAG}_${which}${i}.json').read().strip().splitlines()[-1]); print('$which$i', '$lib', d['ms_per_step'], 'ms/step', d['value'], d['unit'])"
  done
done
