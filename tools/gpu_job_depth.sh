timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -s -k "full_depth_22_layers or eight_layers_against_streamed" 2>&1 < /dev/null | grep -v amdgpu.ids | tail -30
timeout 1500 python -m pytest tests/test_fulldepth_gpu.py -x -q -m gpu -s 2>&1 < /dev/null | grep -v amdgpu.ids | tail -60
