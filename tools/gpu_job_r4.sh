#!/bin/bash
# Round-4 evidence run on one MI355X box (every command under its own timeout, stdin closed):
#   tests  : the GPU test suite (the three multi-minute depth tests are in profiles/r4_depth_parity.log and deselected here: `nodepth`)
#   bench  : the default bench line with the driver's arguments, rocprofv3 kernel stats of the same command, the two --pmc passes
#            (FETCH_SIZE / WRITE_SIZE, separate runs) behind roofline.traffic, SQ counters of the attention streams, the one-rank RCCL path
# usage: tools/gpu_job_r4.sh <tag> [tests|nodepth] [bench]      outputs under gpurun_out/<tag>/
TAG=${1:-r4}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
REPO=$PWD
exec < /dev/null
if [ "$2" = "tests" ] || [ "$2" = "nodepth" ]; then
  DES=""
  [ "$2" = "nodepth" ] && DES="--deselect tests/test_fulldepth_gpu.py::test_configs1_full_depth_against_streamed_oracle --deselect tests/test_model_gpu.py::test_configs0_tinyllama_full_depth_22_layers_against_streamed_oracle --deselect tests/test_model_gpu.py::test_configs2_shape_8_frames_seq4096_eight_layers_against_streamed_oracle"
  (time timeout 1500 python -m pytest tests -m gpu -q --durations=12 $DES 2>&1 | tail -40) > $OUT/gpu_tests.log 2>&1
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
fi
if [ "$3" = "bench" ] || [ "$2" = "bench" ]; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o step -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/kt_bench.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/pmc_write.log 2>&1
  cd $REPO
  F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && [ -n "$W" ] && timeout 300 python tools/hbm_traffic_summary.py "$F" "$W" $OUT/hbm_traffic.json > $OUT/hbm_traffic.log 2>&1
  S=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $OUT/kernel_stats.csv
  # SQ counters of the attention streams (attn4 forward / dQ / dK-dV at the bench shape): MFMA busy cycles against wave cycles
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_attn -o a -- python $REPO/tools/prof_attn4.py > $OUT/pmc_attn.log 2>&1
  cd $REPO
  A=$(find $OUT/pmc_attn -name "*counter_collection.csv" | head -1)
  [ -n "$A" ] && timeout 120 python - "$A" > $OUT/pmc_attn_summary.log 2>&1 <<'PY'
import csv, collections, sys, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "")).strip()[:48]
    a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("per launch means; SQ_WAVE_CYCLES / WAIT / ACTIVE in quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs")
for k, d in acc.items():
    if "attn" not in k: continue
    m = {c: s / max(n, 1) for c, (n, s) in d.items()}
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(4.0 * m.get("SQ_WAVE_CYCLES", 1), 1)
    print(k, {c: round(v) for c, v in m.items()}, f"MFMA busy / (4 x wave cycles) = {busy:.3f}")
PY
  # the N > 1 code path of bench.py on one rank (RCCL collectives forced, same-job async A/B)
  MM355_BENCH_FORCE_DIST=1 MM355_BENCH_AB_ASYNC=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_forced_dist.json 2> $OUT/bench_forced_dist.err
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
fi
ls -la $OUT
