for a in 0 1 0 1; do timeout 300 python bench.py --zero2-async $a --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('async=$a', r['value'], r['ms_per_step'], r['config']['parallelism'], r.get('loss'))"; done
