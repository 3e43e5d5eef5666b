#!/bin/bash
# A/B of one environment switch on one box: kernel trace of 3 steps with VAR=A and VAR=B, kernels matching PATTERN listed.
# bash profiles/scripts/ab_env.sh VAR A B PATTERN
# The switches exist only in the -DTMAE_AB debug build (python t-mae_amd/build.py --ab, built here on the box if missing).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
[ -f t-mae_amd/build_ab/libtmae_ab.so ] || python3 t-mae_amd/build.py --ab || exit 1
export TMAE_LIB_PATH="$GRAFT_REPO_ROOT/t-mae_amd/build_ab/libtmae_ab.so"
VAR=$1; PAT=$4
for val in "$2" "$3"; do
  export $VAR=$val
  rm -rf /tmp/prof_abe_$val
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_abe_$val -o t -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/abe_$val.log" 2>&1 )
  TRACE=$(find /tmp/prof_abe_$val -name "*kernel_trace.csv" | head -1)
  python3 profiles/scripts/summarize_trace.py "$TRACE" 200 > gpurun_out/abe_${val}_summary.md 2>&1
  echo "== $VAR=$val"; head -1 gpurun_out/abe_${val}_summary.md; grep -E "$PAT" gpurun_out/abe_${val}_summary.md | head -24
done
