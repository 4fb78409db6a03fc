import torch, time, sys
sys.path.insert(0, "/root/repo")
from metamorph_amd import ops
dev = "cuda"
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, N, K in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate|up", 28672, 4096), ("down", 4096, 14336), ("lm_head", 128256, 4096)]:
    ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(8 if N < 100000 else 2)]   # rotate weights: no L2/MALL reuse between calls
    for M in (16, 24, 32, 48, 64):
        x = (torch.randn(M, K, device=dev)).bfloat16()
        i = [0]
        def gv():
            i[0] += 1; ops.gemv(x, ws[i[0] % len(ws)])
        def gm():
            i[0] += 1; ops.gemm_splitk(x, ws[i[0] % len(ws)])
        a = t(gv); b = t(gm) if M > 16 else float("nan")
        print(f"{name:8s} M={M:2d}: gemv {a:7.1f} us ({N*K*2/a/1e6:.2f} TB/s)   split-K gemm {b:7.1f} us", flush=True)
