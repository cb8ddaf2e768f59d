#!/bin/bash
# PMC passes over the Griffin-Lim iteration kernel alone (tools/gl_bench.py), one rocprofv3 run per counter group
# (--pmc never together with a trace domain other than the kernel trace).  Writes a summary to
# gpurun_out/<tag>_gl_pmc.txt; copy it to profiles/ to have it judged.
#   bash tools/gl_pmc.sh r02
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_gl_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
G3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
G4="SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_LEVEL_WAVES"
G5="FETCH_SIZE"
G6="WRITE_SIZE"
G7="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7"; do
  i=$((i+1))
  rm -rf $OUT/p$i
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/p$i -- python3 $R/tools/gl_bench.py --iters 6 --reps 1 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
  echo "pass $i done"
done
python3 $R/tools/pmc_summary.py $OUT gl_stream_kernel > $R/gpurun_out/${TAG}_gl_pmc.txt
cat $R/gpurun_out/${TAG}_gl_pmc.txt
