#!/bin/bash
# A/B of two library builds on one box: kernel trace of 3 steps each, attention kernels summarised.
# bash profiles/scripts/ab_attn.sh <old.so (relative to the repo root)>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for tag in new old; do
  if [ $tag = old ]; then export TMAE_LIB_PATH="$GRAFT_REPO_ROOT/$1"; else unset TMAE_LIB_PATH; fi
  rm -rf /tmp/prof_ab_$tag
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab_$tag -o $tag -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/ab_$tag.log" 2>&1 )
  TRACE=$(find /tmp/prof_ab_$tag -name "*kernel_trace.csv" | head -1)
  python3 profiles/scripts/summarize_trace.py "$TRACE" 200 > gpurun_out/ab_${tag}_summary.md 2>&1
  echo "== $tag"; head -1 gpurun_out/ab_${tag}_summary.md; grep "win_attn" gpurun_out/ab_${tag}_summary.md | head -20
done
