#!/bin/bash
# GPU-box job: kernel parity tests (whole suite, then per-test isolation if the process died) + micro-benchmarks.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -rf > gpurun_out/kernels_all.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/kernels_all.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then
  for t in $(grep -o "^def test_[a-z_0-9]*" tests/test_kernels_gpu.py | sed 's/def //'); do
    timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -rf -k "$t" > gpurun_out/k_$t.log 2>&1
    echo "$t rc=$?" >> gpurun_out/kernels_iso.log
  done
fi
timeout 900 python tools/bench_kernels.py --tokens 8192 --out gpurun_out/bench_kernels.json > gpurun_out/bench_kernels.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_kernels.log
tail -60 gpurun_out/kernels_all.log
[ -f gpurun_out/kernels_iso.log ] && cat gpurun_out/kernels_iso.log
tail -60 gpurun_out/bench_kernels.log
