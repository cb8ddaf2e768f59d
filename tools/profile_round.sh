#!/bin/bash
# Everything profiles/ holds for a round:   bash tools/profile_round.sh r06 [A|B]   (two gpurun calls of <= 20 minutes: parts A and B)
# (counter passes never share a run with a trace domain other than the kernel trace)
TAG=${1:-r06}
PART=${2:-AB}   # A: bench, kernel stats, Griffin-Lim counters; B: GEMM counters, timeline, driver's form, decoder counters, stage benchmarks
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [[ $PART == *A* ]]; then
echo "== bench"; python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/bench.err; tail -c 400 $O/${TAG}_bench.json; echo
for PL in 1 0; do
  rm -rf $O/stats$PL
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$PL -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --pipeline $PL > $O/stats$PL.log 2>&1
  cp $(ls $O/stats$PL/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats_pipeline$PL.csv
  echo "== kernel stats pipeline=$PL done"
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipeline 0 > $O/pmc_$C.log 2>&1
done
python3 - <<PY
import csv, glob, json
def mean(counter):
    v = []
    for f in glob.glob('$O/pmc_%s/**/*counter_collection.csv' % counter, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gl_stream_kernel<0' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                v.append(float(r['Counter_Value']))
    return sum(v) / max(1, len(v)), len(v)
f, nf = mean('FETCH_SIZE'); w, nw = mean('WRITE_SIZE')
import sys
sys.path.insert(0, '$R')
import bench
out = {'kernel': 'gl_stream_kernel<0,1102,275,false,3>', 'FETCH_SIZE_KB_mean': f, 'WRITE_SIZE_KB_mean': w, 'dispatches': [nf, nw],
       'kernel_sha16': bench.gl_kernel_sha16(), 'commit': None,
       'hbm_bytes_per_launch': (2.0 * f + w) * 1024.0, 'iterations_per_launch': 3,
       'note': '(2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE counts half the bytes of wide coalesced loads on gfx950 '
               '(MI355X_MICROARCH.md, HBM); separate --pmc passes of bench.py --pipeline 0 --steps 2'}
json.dump(out, open('$O/${TAG}_gl_iter_hbm_bytes_per_launch.json', 'w'), indent=1)
print(out)
PY
echo "== GL SQ counters"; bash $R/tools/gl_pmc.sh $TAG > $O/gl_pmc.log 2>&1; cp $R/gpurun_out/${TAG}_gl_pmc.txt $O/${TAG}_gl_iter_sq_counters.txt
python3 $R/tools/gl_counters_json.py $O/${TAG}_gl_iter_sq_counters.txt $O/${TAG}_gl_iter_valu.json 3
fi
if [[ $PART == *B* ]]; then
echo "== GEMM MFMA counters"
G1="SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
G2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
G3="GRBM_GUI_ACTIVE"
i=0
for G in "$G1" "$G2" "$G3"; do i=$((i+1)); rm -rf $O/mfma/p$i; rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/mfma/p$i -- python3 $R/tools/net_bench.py > $O/mfma_p$i.log 2>&1; done
python3 $R/tools/pmc_summary.py $O/mfma gemm_f32_kernel gemm_f32_pool_kernel cbhg_tail_kernel bigru_kernel > $O/${TAG}_gemm_mfma_counters.txt; head -30 $O/${TAG}_gemm_mfma_counters.txt
echo "== step timeline"; ( cd $R && bash tools/trace_step.sh ${TAG}_trace ) > $O/${TAG}_step_timeline.txt 2> $O/step_timeline.err; wc -l $O/${TAG}_step_timeline.txt
echo "== driver's form"; python3 $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_steps20.json 2>> $O/bench.err; python3 $R/bench.py --steps 20 --warmup 5 --through-facade > $O/${TAG}_bench_steps20_facade.json 2>> $O/bench.err
echo "== decoder L2 counters"; ( cd $R && bash tools/dec_pmc.sh ${TAG} ) > $O/dec_pmc.log 2>&1; cp $R/gpurun_out/${TAG}_dec_pmc.txt $O/${TAG}_decoder_l2_counters.txt
echo "== stage benchmarks"
( echo "# tools/gemm_bench.py"; python3 $R/tools/gemm_bench.py; echo; echo "# tools/net_bench.py"; python3 $R/tools/net_bench.py; echo; echo "# tools/dec_bench.py"; python3 $R/tools/dec_bench.py; echo; echo "# tools/gl_bench.py"; python3 $R/tools/gl_bench.py; echo; echo "# tools/latency_bench.py"; python3 $R/tools/latency_bench.py; echo; echo "# tools/pipeline_sweep.py"; python3 $R/tools/pipeline_sweep.py ) > $O/${TAG}_stage_benchmarks.txt 2>&1
tail -12 $O/${TAG}_stage_benchmarks.txt
fi
