#!/bin/bash
# Kernel-trace profile of 3 timed steps of bench.py on the GPU box: bash profiles/scripts/profile_step.sh <tag> [bench args]
# (BENCH_PY=<another tree's bench.py> profiles that tree instead, e.g. ab_base/bench.py of make_ab_base.sh)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-p}; shift
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python3 "${BENCH_PY:-$GRAFT_REPO_ROOT/bench.py}" --steps 3 --warmup 2 --no-cpu-baseline --no-vfe-prefetch "$@" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}.log" 2>&1 )
TRACE=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1)
STATS=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
echo "trace: $TRACE stats: $STATS"
python3 profiles/scripts/summarize_trace.py "$TRACE" 200 > gpurun_out/${TAG}_summary.md 2>&1
[ -n "$STATS" ] && cp "$STATS" gpurun_out/${TAG}_kernel_stats.csv
head -4 gpurun_out/${TAG}_summary.md
