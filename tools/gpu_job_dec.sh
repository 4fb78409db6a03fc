for pf in 0 64 0 128 32; do echo "== PREFETCH_MIB=$pf"; PREFETCH_MIB=$pf CACHED_ONLY=1 NEW=64 timeout 200 python tools/bench_decode.py 2>&1 < /dev/null | grep "decode step"; done
