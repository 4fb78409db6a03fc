#!/bin/bash
# final check of the round's tree: the one-rank RCCL path of bench.py, then the GPU suite (depth tests: tools/gpu_job_r4.sh / r4_depth_parity.log)
exec < /dev/null
mkdir -p gpurun_out/r4f
MM355_BENCH_FORCE_DIST=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4f/bench_forced_dist.json 2> gpurun_out/r4f/bench_forced_dist.err
echo "forced-dist rc=$?"
(time timeout 1200 python -m pytest tests -m gpu -q --durations=8 --deselect tests/test_fulldepth_gpu.py::test_configs1_full_depth_against_streamed_oracle --deselect tests/test_model_gpu.py::test_configs0_tinyllama_full_depth_22_layers_against_streamed_oracle --deselect tests/test_model_gpu.py::test_configs2_shape_8_frames_seq4096_eight_layers_against_streamed_oracle 2>&1 | tail -16) > gpurun_out/r4f/gpu_tests.log 2>&1
tail -6 gpurun_out/r4f/gpu_tests.log
