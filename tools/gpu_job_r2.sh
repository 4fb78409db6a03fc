#!/bin/bash
# Round-2 evidence run on one MI355X box: GPU test suite, smoke, the default bench line, rocprofv3 kernel stats of the same command,
# and the two --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) behind roofline.traffic.
# usage: tools/gpu_job_r2.sh <tag>      outputs under gpurun_out/<tag>/
TAG=${1:-r2}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
(time python -m pytest tests -m gpu -q 2>&1 | tail -25) > $OUT/gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o step -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/kt_bench.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/pmc_write.log 2>&1
cd $OLDPWD
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/hbm_traffic_summary.py "$F" "$W" $OUT/hbm_traffic.json > $OUT/hbm_traffic.log 2>&1
S=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $OUT/kernel_stats.csv
# the raw traces are large: keep the summaries only
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
ls -la $OUT
