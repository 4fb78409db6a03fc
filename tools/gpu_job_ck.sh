timeout 200 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu -k "recompute_is_bit_identical" 2>&1 < /dev/null | tail -3
timeout 600 python bench.py --batch 24 --grad-checkpointing --ckpt-layers auto --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/ck_b24.err < /dev/null | tail -1 > gpurun_out/ck_b24.json
tail -3 gpurun_out/ck_b24.err; python - <<'PY'
import json
try:
    r=json.load(open('gpurun_out/ck_b24.json'))
    print({k:r[k] for k in ('value','ms_per_step','mfu_vs_bf16_mfma_peak','peak_mem_gb')}, r['config']['parallelism'])
except Exception as e: print("no json", e)
PY
