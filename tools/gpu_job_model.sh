#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -rf -s > gpurun_out/model_all.log 2>&1
echo "pytest rc=$?" >> gpurun_out/model_all.log
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -120 gpurun_out/model_all.log; tail -20 gpurun_out/smoke.log
