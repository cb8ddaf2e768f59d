set -e
for v in -2 -1 14 15; do
  for form in "--steps 20 --warmup 5" "--steps 60 --warmup 10"; do
  echo "gl_wide_from=$v $form"
  python bench.py $form --no-cpu-baseline --set gl_wide_from=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d.get('stage_ms'))"
  done
done
bash tools/trace_step.sh r05_trace > gpurun_out/timeline_gate.txt 2>/dev/null
