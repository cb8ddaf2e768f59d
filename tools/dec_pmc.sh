#!/bin/bash
# PMC passes over the persistent decoder alone (tools/dec_bench.py: the weight-stationary kernel; add "streamed" to DEC_ARGS for decoder_persistent.hip): L2 hits / misses and the bytes fetched from and
# written to memory per launch, one rocprofv3 run per counter group.  Summary in gpurun_out/<tag>_dec_pmc.txt.
#   bash tools/dec_pmc.sh r04
set -e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_dec_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
G2="FETCH_SIZE"
G3="WRITE_SIZE"
G4="TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_READ_sum TCC_WRITE_sum"
G5="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5"; do
  i=$((i+1))
  rm -rf $OUT/p$i
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/p$i -- python3 $R/tools/dec_bench.py 64 persistent ${DEC_ARGS:-rows32} > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
  echo "pass $i done"
done
python3 $R/tools/pmc_summary.py $OUT dec_ws_kernel dec_persistent_kernel > $R/gpurun_out/${TAG}_dec_pmc.txt
cat $R/gpurun_out/${TAG}_dec_pmc.txt
