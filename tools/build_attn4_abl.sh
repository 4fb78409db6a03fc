#!/bin/bash
# timing-only builds of the attn4 stream: tools/build_attn4_abl.sh <abl names as given to gen_attn4.py --abl, or "base"> -> build/ablate_a4_<name>/libmm355.so
# BWD_ABL=<names> tools/build_attn4_abl.sh base -> build/ablate_a4b_<names>/libmm355.so: the same for the backward streams (tools/gen_attn4_bwd.py --abl)
# (all with -DMM355_ATTN4_TIMING: phase stamps for tools/attn4_timing.py; results are WRONG by construction except "base")
# The ablation streams are generated on demand into build/attn4_abl/ (git-ignored); only the product streams live under metamorph_amd/csrc/.
set -e
NAME=$1
ABL=build/attn4_abl; mkdir -p $ABL
INC=metamorph_amd/csrc; DIR=attn4_gen; [ "$NAME" != base ] && { python tools/gen_attn4.py --abl $NAME > /dev/null; DIR=attn4_gen_${NAME//,/_}; INC=$ABL; }
OUT=build/ablate_a4_${NAME//,/_}; [ -n "$BWD_ABL" ] && OUT=build/ablate_a4b_${BWD_ABL//,/_}; mkdir -p $OUT
for f in metamorph_amd/lib/*.o; do b=$(basename $f); [ $b != attn4.o ] && [ $b != attn4_bwd.o ] && cp $f $OUT/$b; done
BINC=metamorph_amd/csrc; BDIR=attn4_bwd_gen; [ -n "$BWD_ABL" ] && { python tools/gen_attn4_bwd.py --abl $BWD_ABL > /dev/null; BDIR=attn4_bwd_gen_${BWD_ABL//,/_}; BINC=$ABL; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-spill-vgpr-to-agpr=0 -DMM355_ATTN4_TIMING -I$BINC -DATTN4B_GEN_DIR=$BDIR -c metamorph_amd/csrc/attn4_bwd.hip -o $OUT/attn4_bwd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-spill-vgpr-to-agpr=0 -DMM355_ATTN4_TIMING -I$INC -DATTN4_GEN_DIR=$DIR -c metamorph_amd/csrc/attn4.hip -o $OUT/attn4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmm355.so $OUT/*.o
