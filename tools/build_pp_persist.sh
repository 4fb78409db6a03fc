#!/bin/bash
# timing experiment: the ping-pong GEMM as a persistent launch -- <n> workgroups (256 = one per CU) walk the tiles instead of one workgroup per tile:
#   tools/build_pp_persist.sh 256  -> build/pp_persist256/libmm355.so (MM355_LIB_PATH=...); plain gemm_pp_kernel only; results are correct
set -e
D=build/pp_persist$1
mkdir -p "$D/obj"
for f in metamorph_amd/lib/*.o; do b=$(basename "$f"); if [ "$b" != gemm_bf16.o ]; then cp "$f" "$D/obj/$b"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 -Iinclude -DMM355_PP_PERSIST=$1 -c metamorph_amd/csrc/gemm_bf16.hip -o "$D/obj/gemm_bf16.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$D/libmm355.so" "$D"/obj/*.o
echo "$D/libmm355.so"
