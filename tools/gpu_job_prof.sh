#!/bin/bash
# rocprofv3 kernel trace + stats of the headline step; usage: gpu_job_prof.sh <batch> <tag>
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
B=${1:-8}; TAG=${2:-step}
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof/${TAG}_stdout.log 2>&1
echo "rocprof rc=$?"
ls gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name "${TAG}_kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f" | cut -c1-200
find gpurun_out/prof -name "*kernel_trace.csv" -size +30M -delete
grep '"metric"' gpurun_out/prof/${TAG}_stdout.log | cut -c1-400
