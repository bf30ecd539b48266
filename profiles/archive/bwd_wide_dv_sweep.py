import os, sys, statistics, torch
sys.path.insert(0, "/root/repo")
import sparsefactorization_amd as sfa
from sparsefactorization_amd.chord import _launch_bwd
dev = torch.device("cuda:0")
for B, N, L, C in ((32, 2000, 12, 128), (32, 2048, 12, 64), (32, 2048, 12, 128), (16, 4096, 13, 128)):
    g = torch.Generator(device=dev).manual_seed(0)
    sets = 6
    Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(sets)]
    Vs = [torch.randn(B, N, C, device=dev, generator=g) for _ in range(sets)]
    dZ = torch.randn(B, N, C, device=dev, generator=g)
    dWs = [torch.empty_like(Ws[0]) for _ in range(sets)]
    dVs = [torch.empty_like(Vs[0]) for _ in range(sets)]
    def reading(n=60):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(n):
            s = i % sets
            _launch_bwd(dZ, Ws[s], Vs[s], dWs[s], dVs[s], B, N, L, C, N * C, None)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    reading()
    t = {w: [] for w in (0, 1, 2)}
    for r in range(5):
        for w in ((0, 1, 2) if r % 2 == 0 else (2, 1, 0)):
            sfa.set_tuning("fwd_wide", w); t[w].append(reading())
    sfa.set_tuning("fwd_wide", 0)
    print(f"B={B} N={N} L={L} C={C}: backward step (dW + dV), fwd_wide 0 / 1 / 2: " + " / ".join(f"{statistics.median(t[w]):.2f}" for w in (0, 1, 2)), flush=True)
