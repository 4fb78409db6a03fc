for n in ${GST_LIST}; do echo "== $n"; MM355_LIB_PATH=build/gst_$n/libmm355.so timeout 120 python tools/bench_gemm_st.py --bench-only --no11 2>&1 | grep bench; done
