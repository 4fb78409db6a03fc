timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemv or attn_decode" 2>&1 < /dev/null | tail -5
timeout 300 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "decode" 2>&1 < /dev/null | tail -3
CACHED_ONLY=1 NEW=64 timeout 300 python tools/bench_decode.py 2>&1 < /dev/null | grep -v amdgpu.ids
