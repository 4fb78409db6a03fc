#!/bin/bash
# experiment builds: tools/build_swb_stagger.sh <phases> <sleeps per phase (x 127 x 64 cycles)> -> build/swb_p<phases>_s<sleeps>/libmm355.so
set -e
D=build/swb_p$1_s$2
mkdir -p "$D/obj"
for f in metamorph_amd/lib/*.o; do b=$(basename "$f"); if [ "$b" != gemm_bf16.o ]; then cp "$f" "$D/obj/$b"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 -DMM355_SWB_PHASES=$1 -DMM355_SWB_STAGGER=$2 -c metamorph_amd/csrc/gemm_bf16.hip -o "$D/obj/gemm_bf16.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$D/libmm355.so" "$D"/obj/*.o
echo "$D/libmm355.so"
