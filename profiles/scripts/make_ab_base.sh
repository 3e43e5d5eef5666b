#!/bin/bash
# Snapshot of a commit as a second, self-contained tree under ab_base/ (git-ignored; it travels to the GPU box with the snapshot)
# for same-box A/B runs of whole trees -- library, Python side and ABI together:  bash profiles/scripts/make_ab_base.sh [commit]
set -e
cd "$(git rev-parse --show-toplevel)"
C=${1:-HEAD}
rm -rf ab_base && mkdir ab_base
git archive "$C" bench.py t-mae_amd include oracle profiles/round5_pmc.json profiles/round4_z_step_bytes.json profiles/round3_pmc_counters.json 2>/dev/null | tar -x -C ab_base \
  || git archive "$C" bench.py t-mae_amd include oracle | tar -x -C ab_base
python3 ab_base/t-mae_amd/build.py > /dev/null
rm -rf ab_base/t-mae_amd/build
git rev-parse --short "$C" > ab_base/COMMIT
echo "ab_base = $(cat ab_base/COMMIT)"
