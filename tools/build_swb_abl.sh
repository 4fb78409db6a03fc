#!/bin/bash
# timing-only builds that bound a persistent / overlapped form of gemm_pp_swiglu_bwd_kernel:  tools/build_swb_abl.sh <mode>
#   1 = the K loop alone (no epilogue traffic)     2 = the epilogue alone (no K loop)
#   3 = the K loop WITH one tile's epilogue traffic (32 loads + 80 stores of 16 B per thread, at the epilogue's addresses) spread over its phases
# -> build/swb_abl<mode>/libmm355.so   (run with MM355_LIB_PATH=...; results are WRONG by construction, only durations mean anything)
set -e
D=build/swb_abl$1
mkdir -p "$D/obj"
for f in metamorph_amd/lib/*.o; do b=$(basename "$f"); if [ "$b" != gemm_bf16.o ]; then cp "$f" "$D/obj/$b"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 -DMM355_SWB_ABL=$1 -c metamorph_amd/csrc/gemm_bf16.hip -o "$D/obj/gemm_bf16.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$D/libmm355.so" "$D"/obj/*.o
echo "$D/libmm355.so"
