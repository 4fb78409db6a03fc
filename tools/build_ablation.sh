#!/bin/bash
# timing-only ablation builds of libmm355.so: tools/build_ablation.sh <name> <-DMACRO ...>   ->  build/ablate_<name>/libmm355.so
# (run them with MM355_LIB_PATH=...; results are WRONG by construction, only kernel durations mean anything)
set -e
NAME=$1; shift
OUT=build/ablate_$NAME; mkdir -p $OUT
cd metamorph_amd/csrc
for f in gemm_bf16 rowwise elementwise attn attn2 attn3 attn4 decode losses; do
  if [ -f ../lib/$f.o ] && [ $f != attn3 ] && [ $f != attn4 ] && [ $f != attn2 ] && [ $f != attn ]; then cp ../lib/$f.o ../../$OUT/$f.o; else
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c $f.hip -o ../../$OUT/$f.o & fi
done
wait
cd ../../$OUT && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmm355.so *.o
