#!/bin/bash
# Same-box A/B of two TREES: bench.py of ab_base/ (profiles/scripts/make_ab_base.sh) against this tree's, A B A B, 20 timed steps each.
#   bash profiles/scripts/ab_tree.sh [tag] [extra bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-abt}; shift
echo "A = ab_base ($(cat ab_base/COMMIT)), B = this tree"
# one discarded run first: the first process on a fresh box pays for MIOpen's solver search and cold file caches
timeout -k 10 300 python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary "$@" > /dev/null 2>&1
for i in 1 2; do
  for which in A B; do
    b=bench.py; [ $which = A ] && b=ab_base/bench.py
    timeout -k 10 300 python3 $b --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" \
      > gpurun_out/${TAG}_${which}${i}.json 2> gpurun_out/${TAG}_${which}${i}.err || { echo "run $which$i failed"; tail -3 gpurun_out/${TAG}_${which}${i}.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('gpurun_out/${TAG}_${which}${i}.json').read().strip().splitlines()[-1]); print('$which$i', d['ms_per_step'], 'ms/step', d['value'], d['unit'])"
  done
done
