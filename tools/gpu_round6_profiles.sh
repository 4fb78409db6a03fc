#!/bin/bash
# The round-6 evidence run: everything profiles/r6_* is copied from.  Usage on the GPU box, from the repo root:
#   bash tools/gpu_round6_profiles.sh [tests] [bench] [pmc] [modes] [decode] [attn] [swbabl] [vendor]     (no argument: all)
# Output: gpurun_out/r6p/*.  rocprofv3 runs from /tmp (TMPDIR=/tmp).  Counter passes are SEPARATE runs with --kernel-trace only
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"; gpurun refuses --pmc with other trace domains).
set -u
OUT=gpurun_out/r6p; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
WHAT="${*:-tests bench pmc modes decode attn swbabl ragged vendor}"
R=$(pwd)
PMC_CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing"
for w in $WHAT; do case $w in
tests)
  (time timeout 1500 python -m pytest tests -m gpu -q --durations=40) > $OUT/gpu_tests.log 2>&1; echo "rc=$?" >> $OUT/gpu_tests.log ;;
bench)
  timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/step -o step -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$OUT/bench_profiled.json 2> $R/$OUT/bench_profiled.err)
  find $OUT/step -name "*kernel_stats.csv" -exec cp {} $OUT/step_b16_kernel_stats.csv \; ;;
pmc)
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -o f -- $PMC_CMD > $R/$OUT/pmc_fetch.log 2>&1)
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -o w -- $PMC_CMD > $R/$OUT/pmc_write.log 2>&1)
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/pmc_sq -o s -- $PMC_CMD > $R/$OUT/pmc_sq.log 2>&1)
  F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1); S=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1)
  python tools/hbm_traffic_summary.py "$F" "$W" $OUT/step_b16_hbm_traffic.json > $OUT/step_b16_hbm_traffic.txt 2>&1
  python tools/pmc_mfma_busy.py "$S" $OUT/step_b16_mfma_busy.json > $OUT/step_b16_mfma_busy.txt 2>&1
  rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq ;;
modes)
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --seq 4096 --frames 8 --batch 8 > $OUT/bench_seq4096frames8batch8.json 2> $OUT/bench_modes.err
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --all-generation > $OUT/bench_allgeneration.json 2>> $OUT/bench_modes.err
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --llama31-rope > $OUT/bench_llama31_rope.json 2>> $OUT/bench_modes.err ;;
decode)
  timeout 600 python tools/bench_decode.py > $OUT/decode_bench.log 2>&1
  for B in 1 8 32; do (cd /tmp && NO_GREEDY=1 BATCHES=$B NEW=32 timeout 300 rocprofv3 --kernel-trace -d $R/$OUT/dec_b$B -o dec -- python $R/tools/bench_decode.py > /dev/null 2>&1)
    python tools/rocpd_kernels.py $OUT/dec_b$B/dec_results.db | grep -v "at::native\|rocclr" | head -14 > $OUT/decode_kernels_b$B.txt; rm -rf $OUT/dec_b$B; done ;;
attn)
  timeout 300 python tools/bench_attn4.py > $OUT/attn4_fwd_bench.log 2>&1
  timeout 300 python tools/bench_attn4_bwd.py > $OUT/attn4_bwd_bench.log 2>&1 ;;
swbabl)
  # the experiment that bounds a persistent / overlapped gemm_pp_swiglu_bwd_kernel: K loop alone, epilogue alone, and the
  # K loop with one tile's epilogue traffic issued inside it (timing-only builds of tools/build_swb_abl.sh, made in the build container)
  { for rep in 1 2; do python tools/bench_swb_abl.py; for m in 1 2 3; do MM355_LIB_PATH=$R/build/swb_abl$m/libmm355.so python tools/bench_swb_abl.py; done; done; } > $OUT/swiglu_bwd_overlap_bound.log 2>&1 ;;
ragged)
  # padding-free decoder rows on ragged batches (mean padding share 0.4): compact rows on / off, and the dense batch on the same box
  for cr in on off; do timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ragged 0.4 --compact-rows $cr > $OUT/bench_ragged40_compact_$cr.json 2>> $OUT/bench_ragged.err; done
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_dense_same_box.json 2>> $OUT/bench_ragged.err ;;
vendor)
  timeout 900 python tools/bench_vendor_step.py > $OUT/vendor_step.json 2> $OUT/vendor_step.err ;;
esac; done
rm -rf $OUT/step
ls -la $OUT
