#!/bin/bash
# The round-5 evidence run (one gpurun call): everything profiles/r5_* is copied from.  Usage on the GPU box, from the repo root:
#   bash tools/gpu_round5_profiles.sh [tests] [bench] [decode] [attn]      (no argument: all four)
# Output: gpurun_out/r5p/*.  rocprofv3 runs from /tmp (TMPDIR=/tmp), kernel traces only (no counters in these passes).
set -u
OUT=gpurun_out/r5p; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
WHAT="${*:-tests bench decode attn}"
R=$(pwd)
for w in $WHAT; do case $w in
tests)
  (time timeout 1500 python -m pytest tests -m gpu -q --durations=12) > $OUT/gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/gpu_tests.log
  (timeout 600 python -m pytest tests/test_attn_hostile_gpu.py -m gpu -q -s) > $OUT/attn_hostile_tests.log 2>&1; echo "rc=$?" >> $OUT/attn_hostile_tests.log ;;
bench)
  timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/step -o step -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$OUT/bench_profiled.json 2> $R/$OUT/bench_profiled.err)
  find $OUT/step -name "*kernel_stats.csv" -exec cp {} $OUT/step_b16_kernel_stats.csv \; ;;
decode)
  timeout 600 python tools/bench_decode.py > $OUT/decode_bench.log 2>&1
  timeout 300 python tools/bench_gemv.py > $OUT/gemv_rows.log 2>&1
  for B in 1 8; do (cd /tmp && NO_GREEDY=1 BATCHES=$B NEW=32 timeout 300 rocprofv3 --kernel-trace -d $R/$OUT/dec_b$B -o dec -- python $R/tools/bench_decode.py > /dev/null 2>&1)
    python tools/rocpd_kernels.py $OUT/dec_b$B/dec_results.db | grep -v "at::native\|rocclr" | head -12 > $OUT/decode_kernels_b$B.txt; done ;;
attn)
  timeout 300 python tools/bench_attn4.py > $OUT/attn4_fwd_bench.log 2>&1
  timeout 300 python tools/bench_attn4_bwd.py > $OUT/attn4_bwd_bench.log 2>&1 ;;
esac; done
ls -la $OUT
