#!/bin/bash
# timing-only builds of the gemm_st stream: tools/build_gemm_st_abl.sh <name> [generator options] -> build/gst_<name>/libmm355.so
#   <name> = "base" (the shipped stream, e.g. with other --b1/--b2/--dma-every) or ablations joined by "+" (nodma+nobar, nolds, nomfma ...:
#   WRONG results by construction).  Everything else of the library is linked from metamorph_amd/lib/*.o (build that first).
set -e
NAME=$1; shift
TAG=${NAME//+/_}; [ -n "$SUFFIX" ] && TAG=${TAG}_$SUFFIX
OUT=build/gst_$TAG; mkdir -p $OUT
if [ "$NAME" = base ]; then python tools/gen_gemm_st.py --out $OUT/gen "$@"; else python tools/gen_gemm_st.py --abl $NAME "$@"; rm -rf $OUT/gen; mv build/gemm_st_abl_${NAME//+/_} $OUT/gen; fi
for f in metamorph_amd/lib/*.o; do b=$(basename $f); [ $b != gemm_st.o ] && cp $f $OUT/$b; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-spill-vgpr-to-agpr=0 $EXTRA_DEFS -I$OUT -DGST_GEN_DIR=gen -c metamorph_amd/csrc/gemm_st.hip -o $OUT/gemm_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmm355.so $OUT/*.o
rm -f $OUT/*.o
echo $OUT/libmm355.so
