#!/bin/bash
# dense_wgrad_halo_kernel on the decoder conv's shape with the A/B build's variants (TMAE_DW_VAR bit 0: in-kernel stamps, bit 1: raised
# priority around the MFMA groups): bash profiles/scripts/ab_dense_wgrad_var.sh   (GPU box; needs python t-mae_amd/build.py --ab)
cd "$GRAFT_REPO_ROOT"
for v in ${DW_VARS:-0 2 1 0 2}; do
  echo "== TMAE_DW_VAR=$v"
  TMAE_DW_VAR=$v TMAE_LIB_PATH="$GRAFT_REPO_ROOT/t-mae_amd/build_ab/libtmae_ab.so" timeout -k 10 120 python3 profiles/scripts/dense_wgrad_probe.py 2>&1 | grep -v amdgpu.ids
done
