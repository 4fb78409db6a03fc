timeout 400 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "paired_weight_gradient or batch_of_left_padded or hf_generate or decode" 2>&1 < /dev/null | tail -4
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); ro=r['roofline']; print(r['value'], r['ms_per_step'], ro['frac'], ro['traffic'], ro['traffic_source'], ro['plain']['frac'], ro['fused_mlp']['frac'])"
