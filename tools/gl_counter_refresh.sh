TAG=r04
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipeline 0 > $O/pmc_$C.log 2>&1
done
python3 - <<PY
import csv, glob, json, sys
sys.path.insert(0, '$R')
import bench
def mean(counter):
    v = []
    for f in glob.glob('$O/pmc_%s/**/*counter_collection.csv' % counter, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gl_stream_kernel<0' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                v.append(float(r['Counter_Value']))
    return sum(v) / max(1, len(v)), len(v)
f, nf = mean('FETCH_SIZE'); w, nw = mean('WRITE_SIZE')
out = {'kernel': 'gl_stream_kernel<0,1102,275,false,3>', 'FETCH_SIZE_KB_mean': f, 'WRITE_SIZE_KB_mean': w, 'dispatches': [nf, nw],
       'kernel_sha16': bench.gl_kernel_sha16(), 'commit': None,
       'hbm_bytes_per_launch': (2.0 * f + w) * 1024.0, 'iterations_per_launch': 3,
       'note': '(2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE counts half the bytes of wide coalesced loads on gfx950 '
               '(MI355X_MICROARCH.md, HBM); separate --pmc passes of bench.py --pipeline 0 --steps 2'}
json.dump(out, open('$O/${TAG}_gl_iter_hbm_bytes_per_launch.json', 'w'), indent=1)
print(out)
PY
bash $R/tools/gl_pmc.sh $TAG > $O/gl_pmc.log 2>&1; cp $R/gpurun_out/${TAG}_gl_pmc.txt $O/${TAG}_gl_iter_sq_counters.txt
python3 $R/tools/gl_counters_json.py $O/${TAG}_gl_iter_sq_counters.txt $O/${TAG}_gl_iter_valu.json 3
