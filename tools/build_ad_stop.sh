#!/bin/bash
# timing-only builds of mm355_attn_decode's wide kernel, cut short after a phase:  tools/build_ad_stop.sh <n>
#   0 = return after the kv_lens load   1 = after the score phase (+ V prefetch issue)   2 = after the softmax   3 = after PV (before the merge / store)
# -> build/ad_stop<n>/libmm355.so   (run with MM355_LIB_PATH=...; results are WRONG by construction, only durations mean anything)
set -e
D=build/ad_stop$1
mkdir -p "$D/obj"
for f in metamorph_amd/lib/*.o; do b=$(basename "$f"); if [ "$b" != decode.o ]; then cp "$f" "$D/obj/$b"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 -Iinclude -DMM355_AD_STOP=$1 -c metamorph_amd/csrc/decode.hip -o "$D/obj/decode.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$D/libmm355.so" "$D"/obj/*.o
echo "$D/libmm355.so"
