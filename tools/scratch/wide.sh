set -e
for v in -2 -1 14 15 16 17 18; do
  echo "gl_wide_from=$v"
  python bench.py --steps 60 --warmup 10 --set gl_wide_from=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d.get('stage_ms'))"
done
