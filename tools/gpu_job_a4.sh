mkdir -p gpurun_out/a4
( timeout 200 python tools/bench_attn4.py --quick | grep -v "OK$" | tail -8
for n in base; do
  MM355_LIB_PATH=build/ablate_a4_$n/libmm355.so SUMMARY=1 TAG=$n CAUSAL=0 B=2 L=4096 timeout 60 python tools/attn4_timing.py 2>&1 | grep "^\["
  MM355_LIB_PATH=build/ablate_a4_$n/libmm355.so SUMMARY=1 TAG=$n-causal timeout 60 python tools/attn4_timing.py 2>&1 | grep "^\["
done
python - <<'PY'
import sys; sys.argv=['x']
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import bench_attn4 as b
b.bench(16, 2048, 32, 8, variants=(3, 4))
b.bench(16, 2048, 32, 8, variants=(3, 4))
b.bench(8, 4096, 32, 8, variants=(3, 4))
b.bench(16, 2048, 64, 8, causal=False, variants=(3, 4))
PY
) > gpurun_out/a4/run.log 2>&1
cat gpurun_out/a4/run.log
