timeout 400 python -m pytest tests/test_model_gpu.py -x -q -m gpu -s -k "transposed_weight_cache" 2>&1 < /dev/null | grep -v amdgpu | tail -25
