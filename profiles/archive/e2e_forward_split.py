#!/usr/bin/env python3
"""End-to-end PSFNet forward at the headline shape (Adding, N=16384, B=64): where does the time go?

Times (HIP events, no_grad): the whole net(x); the producers only (init_linear + g + the M link MLPs); the chain
only (chord_chain on pre-materialised W). The chain is what bench.py measures; the producers are PyTorch-ROCm
(rocBLAS/hipBLASLt GEMMs + GELU) — SURVEY.md §8f row 3 is about fusing them.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import psf_training  # noqa: E402


def time_ms(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    out = {}
    for problem, n_vec, B in (("adding", 16384, 64), ("order", 16384, 64), ("adding", 2048, 64)):
        net = psf_training.build_model(problem, n_vec).to(dev).eval()
        X, _ = psf_training.make_split(problem, B, n_vec, dev, 1)
        with torch.no_grad():
            def embed():
                d = X
                pos_done = False
                if problem == "order":  # as PSFNet.forward: lookup + positional add in one pass
                    from sparsefactorization_amd.token_linear import embed_tokens
                    pos_done = net.use_pos_embedding and not net.add_init_linear_layer
                    d = embed_tokens(d.squeeze(-1), net.embedding, net.pos_embedding.weight if pos_done else None)
                if net.add_init_linear_layer:
                    d = net.init_linear(d)
                if net.use_pos_embedding and not pos_done:
                    d = d + net.pos_embedding.weight.unsqueeze(0)
                return d
            data = embed()
            V0 = net.g(data)
            Ws = net.link_weights(data)
            t_all = time_ms(lambda: net(X))
            # the same forward captured once in a HIP graph and replayed (static input buffer)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                net(X)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                y_static = net(X)
            t_graph = time_ms(graph.replay)
            assert torch.equal(y_static, net(X))
            t_prod = time_ms(lambda: (embed(), net.g(data), net.link_weights(data)))
            t_chain = time_ms(lambda: sfa.chord_chain(Ws, V0, True))
            t_embed = time_ms(embed)
            from sparsefactorization_amd import fused_mlp
            blocks = [net.g] + list(net.fs)
            t_mlp_fused = time_ms(lambda: fused_mlp.fused_mlp_forward(data, blocks)) if fused_mlp.eligible(data, blocks) else None
            sfa.set_tuning("mlp_variant", 1)  # streaming weights (what the h = 128 networks use)
            t_mlp_stream = time_ms(lambda: fused_mlp.fused_mlp_forward(data, blocks))
            sfa.set_tuning("mlp_variant", 0)
            fused_mlp.enabled = False
            t_mlp_torch = time_ms(lambda: [blk(data) for blk in blocks])
            fused_mlp.enabled = True
            from sparsefactorization_amd.psfnet import _flat_head
            t_head = time_ms(lambda: _flat_head(net.final, V0.reshape(B, -1)))
        key = f"{problem}_N{n_vec}_B{B}"
        out[key] = {"forward_ms": t_all, "forward_graph_replay_ms": t_graph,"producers_ms": t_prod, "embed_ms": t_embed, "mlps_fused_ms": t_mlp_fused, "mlps_fused_streaming_ms": t_mlp_stream,
                    "mlps_pytorch_ms": t_mlp_torch, "chain_ms": t_chain, "head_ms": t_head,
                    "tokens_per_s_end_to_end": B * n_vec / t_all * 1e3}
        print(key, json.dumps(out[key]), flush=True)


if __name__ == "__main__":
    main()
