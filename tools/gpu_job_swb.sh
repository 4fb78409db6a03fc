TAG=base timeout 120 python tools/bench_swb_stagger.py 2>&1 < /dev/null | grep "M="
for n in p4_s4 p3_s6 p2_s8 p4_s9; do TAG=$n MM355_LIB_PATH=build/swb_$n/libmm355.so timeout 120 python tools/bench_swb_stagger.py 2>&1 < /dev/null | grep "M=32768"; done
