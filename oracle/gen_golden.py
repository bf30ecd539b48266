#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REFERENCE's Python in this container.

Runs only where /root/reference exists (the build container); the fixtures it writes are committed and are
the only thing that travels to the GPU box. TEST INFRASTRUCTURE — never imported by the product.

What comes from the reference itself (imported from /root/reference, executed unmodified on CPU):
  * get_chord_indices_assym                      SyntheticExperiments/psf.py:7-32
  * PSFNet.__init__/forward (the hot loop, the reshape of W, the residual, the heads)
                                                 SyntheticExperiments/psf.py:62-191, LRA/psf.py:63-250
  * trained weights                              LRA/attention_maps/pathfinder_epoch27.pt
What does NOT: `torch_sparse.spmm`. torch-sparse==0.6.11 (requirements.txt:146) is an un-vendored third-party
dependency that is neither in /root/reference nor installed, so `from torch_sparse import spmm`
(psf.py:5) is satisfied with oracle.chord_oracle.torch_spmm_port — the published algorithm
(index_select -> mul -> scatter_add) restated. The fixtures therefore pin the index pattern, the loop
structure, W's memory layout and the residual semantics against real reference code, and the spmm arithmetic
against that restatement (cross-checked in tests against spmul/spmul_cuda.cu's formulas and a dense matmul).

    python oracle/gen_golden.py            # rewrites every fixture
"""
from __future__ import annotations

import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle.chord_oracle import torch_spmm_port  # noqa: E402


def import_reference(name: str, relpath: str):
    shim = types.ModuleType("torch_sparse")
    shim.spmm = torch_spmm_port
    sys.modules["torch_sparse"] = shim
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha256(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name: str, **arrays):
    path = os.path.join(OUT, name)
    np.savez(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  keys={sorted(arrays)[:6]}...")


# ---------------------------------------------------------------------------------------------------
# 1. index pattern
# ---------------------------------------------------------------------------------------------------
INDEX_CASES_FULL = [(1, 1), (1, 3), (2, 2), (3, 4), (8, 4), (16, 5), (100, 9), (128, 8), (257, 10)]
INDEX_CASES_HASH = [(1024, 12), (2000, 12), (2048, 12), (4097, 13), (16384, 15)]


def gen_indices(se):
    arrays = {}
    for n, l in INDEX_CASES_FULL:
        rows, cols = se.get_chord_indices_assym(n, l)
        arrays[f"rows_{n}_{l}"] = np.asarray(rows, dtype=np.int64)
        arrays[f"cols_{n}_{l}"] = np.asarray(cols, dtype=np.int64)
    hashes = []
    for n, l in INDEX_CASES_HASH:
        rows, cols = se.get_chord_indices_assym(n, l)
        rows = np.asarray(rows, dtype=np.int64)
        cols = np.asarray(cols, dtype=np.int64)
        hashes.append(f"{n},{l},{sha256(rows)},{sha256(cols)}")
        arrays[f"colshead_{n}_{l}"] = cols[: 4 * l]
        arrays[f"colstail_{n}_{l}"] = cols[-4 * l:]
    arrays["hashed"] = np.asarray(hashes)
    arrays["full_cases"] = np.asarray(INDEX_CASES_FULL, dtype=np.int64)
    save("chord_indices.npz", **arrays)


# ---------------------------------------------------------------------------------------------------
# 2. running the reference PSFNet and capturing the operands of its hot loop
# ---------------------------------------------------------------------------------------------------
def run_and_capture(net, x, loss_fn=None, vfinal_module=None):
    """Forward (and backward) through the REFERENCE module, capturing W_m = fs[m](data), V0 = g(data) and the
    tensor that leaves the hot loop."""
    cap = {"W": [], "V0": None, "Vfin": None}
    hooks = []

    def on_f(_m, _inp, out):
        out.retain_grad()
        cap["W"].append(out)

    def on_g(_m, _inp, out):
        out.retain_grad()
        cap["V0"] = out

    for f in net.fs:
        hooks.append(f.register_forward_hook(on_f))
    hooks.append(net.g.register_forward_hook(on_g))

    if vfinal_module is not None:  # LRA: dropout3 (p=0) sees V right after the loop
        def on_vf(_m, inp):
            inp[0].retain_grad()
            cap["Vfin"] = inp[0]
        hooks.append(vfinal_module.register_forward_pre_hook(on_vf))
    else:  # synthetic: `final` consumes V.view(B, -1)
        def on_final(_m, inp):
            inp[0].retain_grad()
            cap["Vfin"] = inp[0]
        hooks.append(net.final.register_forward_pre_hook(on_final))

    out = net(x)
    res = {"logits": out.detach().numpy().copy()}
    if loss_fn is not None:
        loss = loss_fn(out)
        loss.backward()
        res["loss"] = np.asarray(loss.item(), dtype=np.float64)
        res["dW"] = np.stack([w.grad.numpy() for w in cap["W"]])
        res["dV0"] = cap["V0"].grad.numpy().copy()
        res["gVfin"] = cap["Vfin"].grad.numpy().copy()
    for h in hooks:
        h.remove()
    B = x.shape[0]
    res["W"] = np.stack([w.detach().numpy() for w in cap["W"]])
    res["V0"] = cap["V0"].detach().numpy().copy()
    res["Vfin"] = cap["Vfin"].detach().numpy().reshape(B, net.n_vec, -1).copy()
    if "gVfin" in res:
        res["gVfin"] = res["gVfin"].reshape(B, net.n_vec, -1)
    return res


def state_arrays(net):
    return {"sd::" + k: v.detach().numpy().copy() for k, v in net.state_dict().items()}


def gen_synthetic(se):
    # cfg1: Adding, N=128, config of SyntheticExperiments/synthetic_training_config.py:4-18, n_W = log2(N)
    # (psf_training.py:34), seed as psf_training.py:16.
    torch.manual_seed(42)
    N, n_W = 128, 7
    net = se.PSFNet(vocab_size=1, add_init_linear_layer=True, embedding_size=32, n_vec=N, n_W=n_W,
                    Ws=[32, 'GELU'], V=[32, 'GELU'], n_channels_V=8, n_class=1, pooling_type="FLATTEN",
                    head=['linear'], use_cuda=False, use_residuals=True, use_pos_embedding=False, problem="adding")
    g = torch.Generator().manual_seed(7)
    B = 4
    vals = torch.rand(B, N, generator=g) * 2 - 1
    marks = torch.zeros(B, N)
    for b in range(B):
        idx = torch.randperm(N, generator=g)[:2]
        marks[b, idx] = 1.0
    x = torch.stack([vals, marks], dim=-1)  # [B, N, 2] as synth_data_generation.py:8-28
    y = 0.5 + (vals * marks).sum(1) / 4
    res = run_and_capture(net, x, lambda out: torch.nn.functional.mse_loss(out.squeeze(), y))
    save("psfnet_adding_n128.npz", x=x.numpy(), y=y.numpy(), **res, **state_arrays(net),
         chord_indicies=net.chord_indicies.numpy())

    # Temporal order flavour: embedding + positional embedding, 4 classes (synthetic_training_config.py:72-86)
    torch.manual_seed(43)
    N, n_W = 128, 7
    net = se.PSFNet(vocab_size=6, add_init_linear_layer=False, embedding_size=32, n_vec=N, n_W=n_W,
                    Ws=[32, 'GELU'], V=[32, 'GELU'], n_channels_V=8, n_class=4, pooling_type="FLATTEN",
                    head=['linear'], use_cuda=False, use_residuals=True, use_pos_embedding=True, problem="order")
    g = torch.Generator().manual_seed(8)
    B = 3
    x = torch.randint(0, 4, (B, N, 1), generator=g)
    for b in range(B):
        i, j = sorted(torch.randperm(N, generator=g)[:2].tolist())
        x[b, i, 0], x[b, j, 0] = 4, 5
    y = torch.randint(0, 4, (B,), generator=g)
    res = run_and_capture(net, x, lambda out: torch.nn.functional.cross_entropy(out, y))
    save("psfnet_order_n128.npz", x=x.numpy(), y=y.numpy(), **res, **state_arrays(net))


def gen_lra(lra):
    # (a) trained Pathfinder checkpoint, config LRA/psf_training_config.py:60-88. N=1024, L=12: the last link
    #     has offset 2^10 = 1024 == 0 (mod N): a duplicate self link.
    cfg = dict(vocab_size=225, embedding_size=32, n_vec=1024, n_W=11, Ws=[128, 'GELU'], V=[128, 'GELU'],
               n_channels_V=32, n_class=2, pooling_type="FLATTEN", head=['linear'], use_cuda=False,
               use_residuals=False, dropout1_p=0, dropout2_p=0, dropout3_p=0, init_embedding_weights=False,
               use_pos_embedding=True, problem="pathfinder")
    net = lra.PSFNet(**cfg)
    sd = torch.load(os.path.join(REF, "LRA/attention_maps/pathfinder_epoch27.pt"), map_location="cpu",
                    weights_only=True)
    net.load_state_dict(sd, strict=True)
    net.eval()
    g = torch.Generator().manual_seed(11)
    x = torch.randint(0, 225, (1, 1024), generator=g)
    res = run_and_capture(net, x, None, vfinal_module=net.dropout3)
    # dense attention map W_M...W_1 as ChangedPSF.forward builds it (pathfinder_inference.py:57,75-81):
    # the same spmm applied to an unbatched eye(N); restated here without .cuda()
    W_final = torch.eye(1024, 1024)
    for m in range(net.n_W):
        W = torch.from_numpy(res["W"][m])
        W_final = torch_spmm_port(net.chord_indicies, W.reshape(W.size(0), W.size(1) * W.size(2)), 1024, 1024, W_final)
    wf = W_final.numpy()
    save("lra_pathfinder_ckpt.npz", x=x.numpy(), **res, **state_arrays(net),
         Wfinal_rows=wf[0, ::16, :].copy(), Wfinal_rowsum=wf.sum(-1), Wfinal_sha=np.asarray(sha256(wf)))

    # (b) non-power-of-two N with wrapping far links, CLS pooling, residual, non-linear head. State + input are
    #     stored; W is recomputed by the test on the CPU from the stored weights (keeps the fixture small).
    torch.manual_seed(44)
    cfg = dict(vocab_size=20, embedding_size=16, n_vec=2000, n_W=11, Ws=[16, 'GELU'], V=[16, 'GELU'],
               n_channels_V=16, n_class=10, pooling_type="CLS", head=['non-linear', 32], use_cuda=False,
               use_residuals=True, dropout1_p=0, dropout2_p=0, dropout3_p=0, init_embedding_weights=True,
               use_pos_embedding=True, problem="listops")
    net = lra.PSFNet(**cfg)
    net.eval()
    g = torch.Generator().manual_seed(12)
    x = torch.randint(0, 20, (2, 2000), generator=g)
    res = run_and_capture(net, x, None, vfinal_module=net.dropout3)
    save("lra_listops_n2000.npz", x=x.numpy(), logits=res["logits"], Vfin=res["Vfin"], V0=res["V0"],
         W_sha=np.asarray(sha256(res["W"])), **state_arrays(net))

    # (c) N = 4097 = 4096 + CLS (imdb geometry, psf_training_config.py:89-117), L = 13, C = 8
    torch.manual_seed(45)
    cfg = dict(vocab_size=97, embedding_size=8, n_vec=4097, n_W=12, Ws=[8, 'GELU'], V=[8, 'GELU'],
               n_channels_V=8, n_class=2, pooling_type="CLS", head=['linear'], use_cuda=False,
               use_residuals=True, dropout1_p=0, dropout2_p=0, dropout3_p=0, init_embedding_weights=True,
               use_pos_embedding=False, problem="imdb")
    net = lra.PSFNet(**cfg)
    net.eval()
    g = torch.Generator().manual_seed(13)
    x = torch.randint(0, 95, (2, 4097), generator=g)
    res = run_and_capture(net, x, None, vfinal_module=net.dropout3)
    save("lra_imdb_n4097.npz", x=x.numpy(), logits=res["logits"], Vfin=res["Vfin"], V0=res["V0"],
         W_sha=np.asarray(sha256(res["W"])), **state_arrays(net))

    # state_dict layouts of every shipped PSF checkpoint (names and shapes only)
    lines = []
    for ck in ("pathfinder_epoch27.pt", "imdb_epoch138.pt", "cifar10_epoch35.pt", "PSF_5.pt"):
        sd = torch.load(os.path.join(REF, "LRA/attention_maps", ck), map_location="cpu", weights_only=True)
        for k, v in sd.items():
            lines.append(f"{ck}|{k}|{'x'.join(str(s) for s in v.shape)}")
    save("checkpoint_layouts.npz", layouts=np.asarray(lines))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)  # deterministic CPU reductions
    se = import_reference("ref_se_psf", "SyntheticExperiments/psf.py")
    lra = import_reference("ref_lra_psf", "LRA/psf.py")
    gen_indices(se)
    gen_synthetic(se)
    gen_lra(lra)


if __name__ == "__main__":
    main()
