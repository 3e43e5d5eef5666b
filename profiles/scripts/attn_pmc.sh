#!/bin/bash
# PMC passes over bench.py --probe-only (token GEMM / wgrad / attention-backward probes); run on the GPU box:
#   bash profiles/scripts/attn_pmc.sh <tag>
# One rocprofv3 run per counter group (SQ has 8 slots, TCC 4); --pmc never combined with tracing.
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r2}
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/g$i" -o p -- python3 bench.py --probe-only --no-cpu-baseline > "$OUT/g$i.log" 2>&1
  echo "group $i ($GROUP) rc $?" >> "$OUT/status.txt"
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SMEM
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum
GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
python3 profiles/scripts/pmc_kernels.py "$OUT" > "$OUT/summary.md" 2>&1
