#!/bin/bash
# SQ / TCC counter passes for the multislice kernel alone (tools/kbench.py).  Each --pmc set is its own run
# (gfx950: 8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE do not fit one pass).  Usage: tools/pmc_sq.sh OUTDIR B [lib]
# The program follows "--" directly (no env/bash hop under the profiler).
set -u
OUT=${1:-gpurun_out/pmc_sq}; B=${2:-32}; LIB=${3:-}
[ -n "$LIB" ] && export ADM_LIB_PATH=$LIB
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_DEP_WAIT SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" \
  "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 tools/kbench.py $B 2 > $OUT/p$i.log 2>&1
  echo "pass $i ($SET): exit $?"
done
python3 tools/pmc_summary.py $OUT
