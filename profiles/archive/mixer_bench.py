"""A/B in one process: the mixer forward with W through memory (psf_mlp_fwd_f32 + psf_chord_chain_fwd_f32) against the
mixer whose steps compute their own W tile (psf_mixer_fwd_f32, csrc/fwd_mlp_step.h). HIP events over back-to-back calls,
rounds interleaved, median and min per arm. Usage: python profiles/mixer_bench.py [--rounds 7] [--reps 20] [--limits 0,2,3]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsefactorization_amd as sfa  # noqa: E402
from sparsefactorization_amd import fused_mixer, fused_mlp  # noqa: E402
from sparsefactorization_amd.psfnet import MLPBlock  # noqa: E402

# name, B, N, E, h, C, L, M, residual
SHAPES = [
    ("cfg2_adding_B64", 64, 16384, 32, 32, 8, 15, 14, True),
    ("order_B40", 40, 16384, 32, 32, 8, 15, 14, True),
    ("genome_B16", 16, 16384, 32, 32, 32, 15, 14, False),
    ("pathfinder_B64", 64, 1024, 32, 128, 32, 12, 11, False),
    ("imdb_B32", 32, 4097, 32, 128, 32, 13, 12, True),
    ("cifar10_B32", 32, 1024, 16, 16, 16, 11, 10, False),
    ("adding_n2048_B64", 64, 2048, 32, 32, 8, 12, 11, True),
    ("cfg1_n128_B40", 40, 128, 32, 32, 8, 8, 7, True),       # BASELINE configs[0]: the single-launch LDS-resident mixer
    ("adding_n512_B64", 64, 512, 32, 32, 8, 10, 9, True),
    ("adding_n512_B512", 512, 512, 32, 32, 8, 10, 9, True),
]


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--limits", default="0")
    ap.add_argument("--only", default="")
    ap.add_argument("--alt-lib", default="", help="a second build of the library: its mixer is timed as arm `alt` in the same process")
    ap.add_argument("--ablate", default="", help="comma list of mixer_ablate masks to time as extra arms (results are wrong)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    limits = [int(v) for v in args.limits.split(",")]
    print(sfa.build_info() if hasattr(sfa, "build_info") else "", flush=True)
    for name, B, N, E, h, C, L, M, res in SHAPES:
        if args.only and args.only not in name:
            continue
        torch.manual_seed(0)
        g = MLPBlock([h, 'GELU'], E, C).to(dev)
        fs = [MLPBlock([h, 'GELU'], E, L).to(dev) for _ in range(M)]
        x = torch.randn(B, N, E, device=dev)
        with torch.no_grad():
            def unfused():
                outs = fused_mlp.fused_mlp_forward(x, [g, *fs])
                return sfa.chord_chain(outs[1:], outs[0], res)

            def producer():
                return fused_mlp.fused_mlp_forward(x, [g, *fs])

            outs = fused_mlp.fused_mlp_forward(x, [g, *fs])

            def chain():
                return sfa.chord_chain(outs[1:], outs[0], res)

            def fused():
                return fused_mixer.mixer_forward(x, g, fs, res)

            # the same with `data` itself computed in the kernels: x = table[token] + pos (the recipe of the token networks)
            vocab = 32
            table = torch.randn(vocab, E, device=dev)
            pos = torch.randn(N, E, device=dev)
            tok = torch.randint(0, vocab, (B, N), device=dev)
            rec = fused_mixer.Recipe.tokens(tok, table, pos)

            def embed_then_unfused():
                d = sfa.token_linear.embed_tokens(tok, _Emb(table), pos) if False else (table[tok] + pos)
                outs_ = fused_mlp.fused_mlp_forward(d, [g, *fs])
                return sfa.chord_chain(outs_[1:], outs_[0], res)

            def fused_recipe():
                return fused_mixer.mixer_forward_in(rec, g, fs, res)

            arms = {"unfused": unfused, "producer": producer, "chain": chain, "embed+unfused": embed_then_unfused,
                    "fused_from_tokens": fused_recipe}
            for lim in limits:
                arms[f"fused_wg{lim}"] = (lambda lim=lim: (sfa.set_tuning("mixer_wg_limit", lim), fused())[1])
            if args.alt_lib:
                import ctypes
                from sparsefactorization_amd import _lib
                main_lib = _lib.load()
                alt = ctypes.CDLL(os.path.abspath(args.alt_lib))
                for fname, (argtypes, restype) in _lib.SIGNATURES.items():
                    fn = getattr(alt, fname)
                    fn.argtypes, fn.restype = argtypes, restype

                def fused_alt():
                    _lib._lib = alt
                    try:
                        return fused_mixer.mixer_forward(x, g, fs, res)
                    finally:
                        _lib._lib = main_lib

                arms["alt"] = fused_alt
            for ab in [int(v) for v in args.ablate.split(",") if v]:
                arms[f"ablate{ab}"] = (lambda ab=ab: (sfa.set_tuning("mixer_ablate", ab), fused(), sfa.set_tuning("mixer_ablate", 0))[1])
            a, b = unfused(), fused()
            rel = float((a - b).abs().max() / a.abs().max())
            for fn in arms.values():
                for _ in range(3):
                    fn()
            t = {k: [] for k in arms}
            for _ in range(args.rounds):
                for k, fn in arms.items():
                    t[k].append(timed(fn, args.reps))
            sfa.set_tuning("mixer_wg_limit", 0)
        tok = B * N
        line = f"{name:18s} rel(fused vs unfused) {rel:.1e} |"
        for k in arms:
            med, mn = statistics.median(t[k]), min(t[k])
            line += f" {k} {med:8.1f} us (min {mn:8.1f}; {tok / med / 1e3:6.2f} Gtok/s) |"
        print(line, flush=True)
        del g, fs, x, outs


if __name__ == "__main__":
    main()
