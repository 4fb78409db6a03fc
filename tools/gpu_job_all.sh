#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rf -s > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
grep -v "^$" gpurun_out/gpu_tests.log | tail -${1:-60}; tail -5 gpurun_out/smoke.log
