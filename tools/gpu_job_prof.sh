#!/bin/bash
# rocprofv3 kernel trace + stats of the headline step (2 timed steps at the given batch)
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1
B=${1:-4}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- python bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof/bench_stdout.log 2>&1
echo "rocprof rc=$?"
ls -la gpurun_out/prof | head; find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f"
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
tail -2 gpurun_out/prof/bench_stdout.log
