#!/usr/bin/env python3
"""The drop-in routes of INTEGRATION.md, timed end to end from Python (host launch overhead included), forward
only, no autograd:
  route 1  the reference's loop unchanged: V = spmm(index, W.reshape(B, N*L), N, N, V); V = V + res   (one library call +
           one PyTorch add per step)
  route 2  V = chord_spmm(W, V, residual=res) per step (residual fused, no index tensor)
  route 3  chord_chain(W_list, V0, use_residual)  (one library call for the whole loop; one launch when N <= 2048)
  route 1L the loop of route 1, unchanged, with `from sparsefactorization_amd.lazy import spmm`: recorded and run as route 3
Also under HIP-graph replay (the per-call host cost disappears; what remains is device time).

    python profiles/dropin_routes.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparsefactorization_amd as sfa  # noqa: E402

SHAPES = {  # name: (B, N, M, C, residual)
    "cfg1 adding N=128": (40, 128, 7, 8, True),
    "pathfinder N=1024 C=32": (64, 1024, 11, 32, False),
    "listops N=2000 C=128": (32, 2000, 11, 128, False),
    "N=2048 C=64": (32, 2048, 11, 64, False),
    "imdb N=4097 C=32": (32, 4097, 12, 32, True),
    "cfg2 adding N=16384": (64, 16384, 14, 8, True),
}


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def graphed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def main():
    dev = torch.device("cuda:0")
    print(f"{'shape':26s} {'route 1 spmm+add':>18s} {'route 2 chord_spmm':>19s} {'route 3 chord_chain':>20s} {'route 1L lazy.spmm':>19s}   (us per forward chain; eager | graph replay)")
    for name, (B, N, M, C, res) in SHAPES.items():
        L = M + 1
        g = torch.Generator(device=dev).manual_seed(0)
        Ws = [0.1 * torch.randn(B, N, L, device=dev, generator=g) for _ in range(M)]
        V0 = torch.randn(B, N, C, device=dev, generator=g)
        index = torch.tensor(sfa.get_chord_indices_assym(N, L), device=dev)

        def route1():
            V = V0
            for W in Ws:
                V = sfa.spmm(index, W.reshape(B, N * L), N, N, V)
                if res:
                    V = V + V0
            return V

        from sparsefactorization_amd import lazy

        def route1_lazy():
            V = V0
            for W in Ws:
                V = lazy.spmm(index, W.reshape(B, N * L), N, N, V)
                if res:
                    V = V + V0
            return V.contiguous()  # first use of the values: the recorded chain runs

        def route2():
            V = V0
            for W in Ws:
                V = sfa.chord_spmm(W, V, V0 if res else None)
            return V

        def route3():
            return sfa.chord_chain(Ws, V0, res)

        with torch.no_grad():
            a, b, c = route1(), route2(), route3()
            assert torch.equal(b, c) and torch.equal(route1_lazy(), c)
            assert torch.allclose(a, c, rtol=1e-5, atol=1e-5 * float(c.abs().max()))
            iters = 200 if N <= 4097 else 50
            import statistics
            routes = (route1, route2, route3, route1_lazy)
            samples = [[] for _ in routes]
            for _ in range(5):  # interleaved rounds, median; all eager runs before any capture
                for i, fn in enumerate(routes):
                    samples[i].append(timeit(fn, iters // 4))
            eager = [statistics.median(x) for x in samples]
            replays = [graphed(fn) for fn in routes]
            samples = [[] for _ in routes]
            for _ in range(5):
                for i, fn in enumerate(replays):
                    samples[i].append(timeit(fn, iters // 4))
            replay = [statistics.median(x) for x in samples]
            cells = list(zip(eager, replay))
        print(f"{name:26s} " + " ".join(f"{e:9.1f} | {r:7.1f}" for e, r in cells))


if __name__ == "__main__":
    main()
