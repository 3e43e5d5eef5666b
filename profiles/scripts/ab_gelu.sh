#!/bin/bash
# A/B on ONE box: the FFN's first Linear + GELU as one launch (dual-store epilogue, default) vs two launches (TMAE_FFN_GELU=pass).
#   bash profiles/scripts/ab_gelu.sh   -> gpurun_out/ab_gelu.txt
cd "$GRAFT_REPO_ROOT" || exit 1
for v in fused pass fused pass; do
  TMAE_FFN_GELU=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'], d['box_peaks']['hbm_copy_gbs'])"
done | tee gpurun_out/ab_gelu.txt
