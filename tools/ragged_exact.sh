mkdir -p gpurun_out/rx
for cfg in "exact 0" "exact 4" "exact 8" "auto 0"; do
  set -- $cfg
  timeout 420 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing --ragged 0.4 --compact-rows $1 --alloc-roundup $2 2>gpurun_out/rx/err_$1_$2.log | tail -1 > gpurun_out/rx/r40_$1_$2.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/rx/r40_$1_$2.json"))
    print("ragged 0.4 compact $1 roundup $2:", round(d["value"],1), "valid tok/s", d["ms_per_step"], "ms/step peak", d.get("config",{}).get("peak_mem_gb", d.get("peak_mem_gb")))
except Exception as e:
    print("ragged 0.4 compact $1 roundup $2: FAILED", e)
PY
done
