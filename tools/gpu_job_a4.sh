mkdir -p gpurun_out/a4
( timeout 120 python tools/bench_attn4_bwd.py --quick ) > gpurun_out/a4/bwd2.log 2>&1
grep -c " OK$" gpurun_out/a4/bwd2.log; grep "MISMATCH\|False\|EXC\|ALL\|SOME" gpurun_out/a4/bwd2.log | cut -c1-200
grep -q "ALL CASES OK" gpurun_out/a4/bwd2.log || exit 1
MM355_LIB_PATH=build/ablate_a4_base/libmm355.so TAG=base timeout 60 python tools/attn4_bwd_timing.py 2>&1 | grep "^\["
REPO=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/a4/prof -o p -- python $REPO/tools/prof_attn4.py > $REPO/gpurun_out/a4/prof.log 2>&1
cd $REPO
S=$(find gpurun_out/a4/prof -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/a4/attn_kernel_stats.csv
find gpurun_out/a4/prof -name "*kernel_trace.csv" -delete; find gpurun_out/a4/prof -name "*.db" -delete
cut -d, -f1-4 gpurun_out/a4/attn_kernel_stats.csv | head -7
