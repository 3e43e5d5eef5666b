#!/bin/bash
# A/B of the wgrad256 variants (debug build, TMAE_WGRAD_VAR) on ONE box: isolated shapes, then the step.
cd "$GRAFT_REPO_ROOT" || exit 1
[ -f t-mae_amd/build_ab/libtmae_ab.so ] || python3 t-mae_amd/build.py --ab || exit 1
export TMAE_LIB_PATH="$GRAFT_REPO_ROOT/t-mae_amd/build_ab/libtmae_ab.so"
for v in 0 1 2 0 1 2; do TMAE_WGRAD_VAR=$v python3 profiles/scripts/wgrad_var_probe.py; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_wgrad.txt
for v in 0 1 0 1; do
  TMAE_WGRAD_VAR=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step var $v', d['ms_per_step'], d['value'])"
done | tee -a gpurun_out/ab_wgrad.txt
