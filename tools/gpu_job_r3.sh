#!/bin/bash
# Round-3 evidence run on one MI355X box: (optional) GPU test suite with durations, the default bench line, rocprofv3 kernel stats of
# the same command, the two --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) behind roofline.traffic, and a
# pass with the memory-side request counters split by destination (if this rocprofv3 exposes them).
# usage: tools/gpu_job_r3.sh <tag> [tests]      outputs under gpurun_out/<tag>/
TAG=${1:-r3}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
if [ "$2" = "tests" ]; then
  (time python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -45) > $OUT/gpu_tests.log 2>&1
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
fi
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
REPO=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA[A-Z_0-9]*\|TCC_[A-Z_0-9]*MALL[A-Z_0-9]*\|[A-Z_0-9]*DRAM[A-Z_0-9]*\|[A-Z_0-9]*MALL[A-Z_0-9]*\|TCC_HIT[A-Z_0-9]*\|TCC_MISS[A-Z_0-9]*\|TCC_REQ[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/memory_side_counters.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o step -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/kt_bench.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/pmc_write.log 2>&1
if grep -q "TCC_EA0_RDREQ_DRAM" $OUT/memory_side_counters.txt; then
  timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_dram -o d -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --layers 4 > $OUT/pmc_dram.log 2>&1
fi
cd $REPO
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/hbm_traffic_summary.py "$F" "$W" $OUT/hbm_traffic.json > $OUT/hbm_traffic.log 2>&1
D=$(find $OUT/pmc_dram -name "*counter_collection.csv" 2>/dev/null | head -1)
[ -n "$D" ] && python - "$D" > $OUT/pmc_dram_summary.log 2>&1 <<'PY'
import csv, collections, sys, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip()[:60]
    a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:14]:
    print(k, {c: (n, round(s / max(n, 1))) for c, (n, s) in d.items()})
PY
S=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
ls -la $OUT
