import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch
from metamorph_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
T = 32768
for name, m, n, k in [("gate_up", T, 28672, 4096), ("down", T, 4096, 14336), ("dW_gate_up", 28672, 4096, T)]:
    a = (torch.randn(m, k, device="cuda", generator=g) * 0.5).bfloat16()
    b = (torch.randn(n, k, device="cuda", generator=g) * 0.5).bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    res = {}
    for v in (11, 91, 92, 93):
        ops.gemm(a, b, out=c, variant=v)
        ts = []
        for _ in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4): ops.gemm(a, b, out=c, variant=v)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 4 * 1e-3)
        res[v] = 2.0 * m * n * k / statistics.median(ts) / 1e12
    print(name, "  ".join(f"v{v}: {t:7.1f} TF" for v, t in res.items()), flush=True)
