#!/bin/bash
# timing-only builds of the attn4 stream: tools/build_attn4_abl.sh <abl names as given to gen_attn4.py --abl, or "base"> -> build/ablate_a4_<name>/libmm355.so
# BWD_ABL=<names> tools/build_attn4_abl.sh base -> build/ablate_a4b_<names>/libmm355.so: the same for the backward streams (tools/gen_attn4_bwd.py --abl)
# (all with -DMM355_ATTN4_TIMING: phase stamps for tools/attn4_timing.py; results are WRONG by construction except "base")
set -e
NAME=$1
DIR=attn4_gen; [ "$NAME" != base ] && { python tools/gen_attn4.py --abl $NAME > /dev/null; DIR=attn4_gen_${NAME//,/_}; }
OUT=build/ablate_a4_${NAME//,/_}; [ -n "$BWD_ABL" ] && OUT=build/ablate_a4b_${BWD_ABL//,/_}; mkdir -p $OUT
for f in gemm_bf16 rowwise elementwise attn attn2 attn3 decode losses; do cp metamorph_amd/lib/$f.o $OUT/$f.o; done
BDIR=attn4_bwd_gen; [ -n "$BWD_ABL" ] && { python tools/gen_attn4_bwd.py --abl $BWD_ABL > /dev/null; BDIR=attn4_bwd_gen_${BWD_ABL//,/_}; OUT2=$OUT; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-spill-vgpr-to-agpr=0 -DMM355_ATTN4_TIMING -DATTN4B_GEN_DIR=$BDIR -c metamorph_amd/csrc/attn4_bwd.hip -o $OUT/attn4_bwd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-spill-vgpr-to-agpr=0 -DMM355_ATTN4_TIMING -DATTN4_GEN_DIR=$DIR -c metamorph_amd/csrc/attn4.hip -o $OUT/attn4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmm355.so $OUT/*.o
