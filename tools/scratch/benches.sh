set -e
mkdir -p gpurun_out/r05b
python bench.py > gpurun_out/r05b/r05_bench.json 2> gpurun_out/r05b/bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r05b/r05_bench_steps20.json 2>> gpurun_out/r05b/bench.err
python bench.py --steps 20 --warmup 5 --through-facade > gpurun_out/r05b/r05_bench_steps20_facade.json 2>> gpurun_out/r05b/bench.err
