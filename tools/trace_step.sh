# Kernel-trace timeline of two pipelined bench steps (both streams), consecutive launches of the same kernel merged:
#   bash tools/trace_step.sh [tag] > profiles/<tag>_step_timeline.txt   (on the GPU box)
set -e
TAG=${1:-r03}
mkdir -p gpurun_out/$TAG
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/$TAG/steptrace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/steptrace -o st -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/gpurun_out/$TAG/steptrace.log 2>&1
cd $R
python3 - $TAG <<'PY'
import csv,glob,collections,sys
tag=sys.argv[1]
f=glob.glob('gpurun_out/%s/steptrace/**/*kernel_trace.csv'%tag,recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# from the sixth persistent decoder launch (a timed step: 3 warm-up + 6 timed calls come first, the launches after them belong to
# bench.py's trained-spectrum harness) for ~32 ms: two pipelined steps
pd=[i for i,r in enumerate(rows) if 'dec_persistent' in r['Kernel_Name'] or 'dec_ws' in r['Kernel_Name']]
i0=pd[5]; t0=int(rows[i0]['Start_Timestamp'])
last=None
for r in rows[i0:]:
    s=(int(r['Start_Timestamp'])-t0)/1e6; e=(int(r['End_Timestamp'])-t0)/1e6
    if s>32: break
    n=r['Kernel_Name'][:70]
    key=(n,r.get('Queue_Id'))
    if 'gl_stream_kernel<0' in n or 'gl_iter_kernel<0' in n or 'dec_gemm' in n:
        if last and last[0]==key: last[2]=e; last[3]+=1; continue
    if last: print('%8.3f %8.3f x%-3d q%s %s'%(last[1],last[2],last[3],last[0][1],last[0][0]))
    last=[key,s,e,1]
if last: print('%8.3f %8.3f x%-3d q%s %s'%(last[1],last[2],last[3],last[0][1],last[0][0]))
PY
