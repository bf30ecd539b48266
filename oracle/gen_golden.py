#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REFERENCE's Python in this container.

Runs only where /root/reference exists (the build container); the fixtures it writes are committed and are
the only thing that travels to the GPU box. TEST INFRASTRUCTURE — never imported by the product.

What comes from the reference itself (imported from /root/reference, executed unmodified on CPU):
  * get_chord_indices_assym                      SyntheticExperiments/psf.py:7-32
  * PSFNet.__init__/forward (the hot loop, the reshape of W, the residual, the heads)
                                                 SyntheticExperiments/psf.py:62-191, LRA/psf.py:63-250
  * trained weights                              LRA/attention_maps/{pathfinder_epoch27,imdb_epoch138,cifar10_epoch35}.pt
  * the genome model                             Genome_Clf/psf.py:63-240
  * the Adding / Temporal-Order data generators  SyntheticExperiments/synth_data_generation.py:8-70
  * TrainModel / TrainPSF (the training /        SyntheticExperiments/psf_utils.py:48-137, LRA/psf_utils.py:48-128 — called as
    evaluation loops)                            they are; their
                                                 `.cuda()` calls are made the identity for the run (no GPU here)
  * the stand-alone attention block              attention_block.py:70-178 (imported with an empty
                                                 `torch_geometric` module: line 9 only imports it)
What does NOT: `torch_sparse.spmm`. torch-sparse==0.6.11 (requirements.txt:146) is an un-vendored third-party
dependency that is neither in /root/reference nor installed, so `from torch_sparse import spmm`
(psf.py:5) is satisfied with oracle.chord_oracle.torch_spmm_port — the published algorithm
(index_select -> mul -> scatter_add) restated. The fixtures therefore pin the index pattern, the loop
structure, W's memory layout and the residual semantics against real reference code, and the spmm arithmetic
against that restatement (cross-checked in tests against spmul/spmul_cuda.cu's formulas and a dense matmul).

    python oracle/gen_golden.py            # rewrites every fixture
    python oracle/gen_golden.py cfg3       # only train_lra_listops_cfg3_n2048.npz
    python oracle/gen_golden.py genome_train   # only train_genome_n320.npz
    PSF_GOLDEN_OUT=/tmp/g python oracle/gen_golden.py && python oracle/compare_golden.py /tmp/g   # do the fixtures reproduce?
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
# PSF_GOLDEN_OUT=<dir>: write there instead (to check that the committed fixtures reproduce: tests/golden/README or
# `python oracle/compare_golden.py <dir>`)
OUT = os.environ.get("PSF_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")
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

    # (d) trained IMDb checkpoint (imdb_epoch138.pt), config LRA/psf_training_config.py:89-117: N = 4097 (4096 + CLS),
    #     L = 13, C = 32, CLS pooling, residual. Logits, V after the loop (every 4th row) and the dense attention map as
    #     ChangedPSF.forward builds it (imdb_inference.py:41,53-59: the same spmm on an unbatched eye(N); restated without
    #     .cuda()): rows [::256], row sums. pos_embedding (4097 x 32, unused: use_pos_embedding = False) is left out of
    #     the stored state to keep the fixture small.
    cfg = dict(vocab_size=97, embedding_size=32, n_vec=4097, n_W=12, Ws=[128, 'GELU'], V=[128, 'GELU'], n_channels_V=32,
               n_class=2, pooling_type="CLS", head=['linear'], use_cuda=False, use_residuals=True, dropout1_p=0.4,
               dropout2_p=0, dropout3_p=0, init_embedding_weights=True, use_pos_embedding=False, problem="imdb")
    net = lra.PSFNet(**cfg)
    net.load_state_dict(torch.load(os.path.join(REF, "LRA/attention_maps/imdb_epoch138.pt"), map_location="cpu",
                                   weights_only=True), strict=True)
    net.eval()
    g = torch.Generator().manual_seed(14)
    x = torch.randint(0, 95, (1, 4097), generator=g)
    x[0, 0] = 96  # CLS token id = vocab_size - 1 (imdb_training.py CLS block)
    res = run_and_capture(net, x, None, vfinal_module=net.dropout3)
    W_final = torch.eye(4097, 4097)
    for m in range(net.n_W):
        W = torch.from_numpy(res["W"][m])
        W_final = torch_spmm_port(net.chord_indicies, W.reshape(W.size(0), W.size(1) * W.size(2)), 4097, 4097, W_final)
    wf = W_final.numpy()
    st = {k: v for k, v in state_arrays(net).items() if k != "sd::pos_embedding.weight"}
    save("lra_imdb_ckpt.npz", x=x.numpy(), logits=res["logits"], V0_rows=res["V0"][:, ::4].copy(),
         Vfin_rows=res["Vfin"][:, ::4].copy(), W_sha=np.asarray(sha256(res["W"])), Wfinal_rows=wf[0, ::256, :].copy(),
         Wfinal_rowsum=wf.sum(-1), **st)
    del W_final, wf

    # (e) trained CIFAR-10 checkpoint (cifar10_epoch35.pt), config psf_training_config.py:31-58: N = 1024, L = 11, C = 16,
    #     no residual. Chain level only (V after the loop); the 16384 x 16 head is left out of the stored state.
    cfg = dict(vocab_size=256, embedding_size=16, n_vec=1024, n_W=10, Ws=[16, 'GELU'], V=[16, 'GELU'], n_channels_V=16,
               n_class=10, pooling_type="FLATTEN", head=['non-linear', 16], use_cuda=False, use_residuals=False,
               dropout1_p=0, dropout2_p=0.2, dropout3_p=0.8, init_embedding_weights=False, use_pos_embedding=True,
               problem="cifar10")
    net = lra.PSFNet(**cfg)
    net.load_state_dict(torch.load(os.path.join(REF, "LRA/attention_maps/cifar10_epoch35.pt"), map_location="cpu",
                                   weights_only=True), strict=True)
    net.eval()
    g = torch.Generator().manual_seed(15)
    x = torch.randint(0, 256, (2, 1024), generator=g)
    res = run_and_capture(net, x, None, vfinal_module=net.dropout3)
    st = {k: v for k, v in state_arrays(net).items() if not k.startswith("sd::final.")}
    save("lra_cifar10_ckpt.npz", x=x.numpy(), V0=res["V0"], Vfin=res["Vfin"], W_sha=np.asarray(sha256(res["W"])), **st)

    # state_dict layouts of every shipped PSF checkpoint (names and shapes only)
    lines = []
    for ck in ("pathfinder_epoch27.pt", "imdb_epoch138.pt", "cifar10_epoch35.pt", "PSF_5.pt"):
        sd = torch.load(os.path.join(REF, "LRA/attention_maps", ck), map_location="cpu", weights_only=True)
        for k, v in sd.items():
            lines.append(f"{ck}|{k}|{'x'.join(str(s) for s in v.shape)}")
    save("checkpoint_layouts.npz", layouts=np.asarray(lines))


# ---------------------------------------------------------------------------------------------------
# 3. the reference's TrainModel run as it is: loss after every optimiser step, evaluation numbers
# ---------------------------------------------------------------------------------------------------
class RecordingLoss(torch.nn.Module):
    """Wraps the loss module handed to TrainModel and keeps every value it returns, in call order."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner
        self.values = []

    def forward(self, pred, target):
        out = self.inner(pred, target)
        self.values.append(float(out.detach()))
        return out


TRAIN_BATCH, TRAIN_BATCHES, EVAL_BATCHES, TRAIN_EPOCHS = 40, 8, 2, 2


def gen_training(se, utils):
    """cfg1 (N = 128, n_W = 7, batch 40; synthetic_training_config.py:4-18,72-86), `seed_everything(42)` as
    psf_training.py:16, Adam(lr) and the loss choice of psf_training.py:50-58, the reference's own TrainModel over
    8 fixed training batches per epoch (shuffle off), 2 epochs, evaluation after each (test_freq = 1).
    The INPUT sequences and labels are drawn with this package's own seeded generators (``synth_data``, whose label rules and
    distributions are pinned to the reference's generators by ``synth_data_reference_samples.npz``) and stored in the fixture:
    what the fixture pins is the reference's model, loss, optimiser and loop on those inputs, not the reference's sampler."""
    import contextlib
    import io
    import re
    from torch.utils.data import DataLoader
    from sparsefactorization_amd import synth_data  # input data only: seeded Adding / Temporal-Order sequences

    N, n_W = 128, 7
    model_cfg = {
        "adding": dict(vocab_size=1, add_init_linear_layer=True, embedding_size=32, n_vec=N, n_W=n_W, Ws=[32, 'GELU'],
                       V=[32, 'GELU'], n_channels_V=8, n_class=1, pooling_type="FLATTEN", head=['linear'],
                       use_cuda=False, use_residuals=True, use_pos_embedding=False, problem="adding"),
        "order": dict(vocab_size=6, add_init_linear_layer=False, embedding_size=32, n_vec=N, n_W=n_W, Ws=[32, 'GELU'],
                      V=[32, 'GELU'], n_channels_V=8, n_class=4, pooling_type="FLATTEN", head=['linear'],
                      use_cuda=False, use_residuals=True, use_pos_embedding=True, problem="order"),
    }
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self  # TrainModel moves every batch with X.cuda() (psf_utils.py:65-66)
    try:
        for problem, cfg in model_cfg.items():
            utils.seed_everything(42)
            net = se.PSFNet(**cfg)
            sd0 = state_arrays(net)
            optimizer = torch.optim.Adam(net.parameters(), lr=0.001)
            loss = RecordingLoss(torch.nn.MSELoss() if problem == "adding" else torch.nn.CrossEntropyLoss())
            g = torch.Generator().manual_seed(2024)
            make = synth_data.adding if problem == "adding" else synth_data.temporal_order
            Xtr, Ytr = make(TRAIN_BATCH * TRAIN_BATCHES, N, generator=g)
            Xva, Yva = make(TRAIN_BATCH * EVAL_BATCHES, N, generator=g)
            Xte, Yte = make(TRAIN_BATCH * EVAL_BATCHES, N, generator=g)
            mk = lambda X, Y: DataLoader(utils.DatasetCreator(X, Y), batch_size=TRAIN_BATCH, shuffle=False,  # noqa: E731
                                         drop_last=True, num_workers=0)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                utils.TrainModel(net=net, trainloader=mk(Xtr, Ytr), valloader=mk(Xva, Yva), testloader=mk(Xte, Yte),
                                 n_epochs=TRAIN_EPOCHS, test_freq=1, optimizer=optimizer, loss=loss, problem=problem,
                                 saving_criteria=1e9)
            text = buf.getvalue()
            num = r"(?:tensor\()?([-+0-9.eE]+)"
            printed = {key: [float(x) for x in re.findall(pat + num, text)]
                       for key, pat in (("train_loss", r"Training loss:\s+"), ("val_loss", r"Val  loss: "),
                                        ("test_loss", r"Test loss: "), ("val_acc", r"Val  accuracy: "),
                                        ("test_acc", r"Test accuracy: "))}
            per_epoch = TRAIN_BATCHES + 2 * EVAL_BATCHES
            assert len(loss.values) == TRAIN_EPOCHS * per_epoch and all(len(v) == TRAIN_EPOCHS for v in printed.values())
            vals = np.asarray(loss.values, dtype=np.float64).reshape(TRAIN_EPOCHS, per_epoch)
            to_np = (lambda t: t.numpy()) if problem == "adding" else (lambda t: t.numpy().astype(np.uint8))
            save(f"train_{problem}_n128.npz", Xtr=to_np(Xtr), Ytr=to_np(Ytr), Xva=to_np(Xva), Yva=to_np(Yva),
                 Xte=to_np(Xte), Yte=to_np(Yte), step_loss=vals[:, :TRAIN_BATCHES].copy(),
                 val_batch_loss=vals[:, TRAIN_BATCHES:TRAIN_BATCHES + EVAL_BATCHES].copy(),
                 test_batch_loss=vals[:, TRAIN_BATCHES + EVAL_BATCHES:].copy(),
                 **{"printed_" + k: np.asarray(v) for k, v in printed.items()}, **sd0,
                 **{"final::" + k[4:]: v for k, v in state_arrays(net).items()})
            print(f"  {problem}: step losses epoch0 {vals[0, :3]} ... epoch1 {vals[1, TRAIN_BATCHES - 1]:.6f}; printed {printed}")
    finally:
        torch.Tensor.cuda = real_cuda


def gen_training_lra(lra, utils):
    """The LRA loop: the reference's TrainPSF (LRA/psf_utils.py:48-128) on its LRA PSFNet, ListOps-style — CLS pooling
    with the CLS token (id vocab_size - 1) prepended as listops_training.py:65-72 does, padding_idx embedding, no
    residual, dropouts 0 (psf_training_config.py:2-30) — at a small size: N = 1 + 127, n_W = 7, C = 16, 2 epochs x 6 fixed
    batches of 32, CrossEntropyLoss, Adam(1e-3), seed_everything(42)."""
    import contextlib
    import io
    import re
    from torch.utils.data import DataLoader
    cfg = dict(vocab_size=17, embedding_size=32, n_vec=128, n_W=7, Ws=[32, 'GELU'], V=[32, 'GELU'], n_channels_V=16,
               n_class=10, pooling_type="CLS", head=['linear'], use_cuda=False, use_residuals=False, dropout1_p=0,
               dropout2_p=0, dropout3_p=0, init_embedding_weights=False, use_pos_embedding=True, problem="listops")
    BATCH, NB, NE, EPOCHS = 32, 6, 2, 2
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        utils.seed_everything(42)
        net = lra.PSFNet(**cfg)
        sd0 = state_arrays(net)
        optimizer = torch.optim.Adam(net.parameters(), lr=0.001)
        loss = RecordingLoss(torch.nn.CrossEntropyLoss())
        g = torch.Generator().manual_seed(2025)

        def split(n):
            data = torch.randint(0, 15, (n, 127), generator=g)
            labels = torch.randint(0, 10, (n,), generator=g)
            cls_token = torch.tensor([[cfg['vocab_size'] - 1] * data.size(0)]).T  # listops_training.py:65-72
            return data, torch.cat([cls_token, data], -1), labels

        (raw_tr, Xtr, Ytr), (raw_va, Xva, Yva), (raw_te, Xte, Yte) = split(BATCH * NB), split(BATCH * NE), split(BATCH * NE)
        mk = lambda X, Y: DataLoader(utils.DatasetCreator(X, Y), batch_size=BATCH, shuffle=False, drop_last=True,  # noqa: E731
                                     num_workers=0)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            utils.TrainPSF(net=net, trainloader=mk(Xtr, Ytr), valloader=mk(Xva, Yva), testloader=mk(Xte, Yte),
                           n_epochs=EPOCHS, test_freq=1, optimizer=optimizer, loss=loss, problem="listops",
                           saving_criteria=1e9)
        text = buf.getvalue()
        num = r"([-+0-9.eE]+)"
        printed = {key: [float(x) for x in re.findall(pat + num, text)]
                   for key, pat in (("train_loss", r"Training loss:\s+"), ("val_loss", r"Val  loss: "),
                                    ("test_loss", r"Test loss: "), ("val_acc", r"Val  accuracy: "),
                                    ("test_acc", r"Test accuracy: "))}
        per_epoch = NB + 2 * NE
        assert len(loss.values) == EPOCHS * per_epoch and all(len(v) == EPOCHS for v in printed.values())
        vals = np.asarray(loss.values, dtype=np.float64).reshape(EPOCHS, per_epoch)
        u8 = lambda t: t.numpy().astype(np.uint8)  # noqa: E731
        save("train_lra_listops_n128.npz", raw_tr=u8(raw_tr), Ytr=u8(Ytr), raw_va=u8(raw_va), Yva=u8(Yva), raw_te=u8(raw_te),
             Yte=u8(Yte), step_loss=vals[:, :NB].copy(), val_batch_loss=vals[:, NB:NB + NE].copy(),
             test_batch_loss=vals[:, NB + NE:].copy(), **{"printed_" + k: np.asarray(v) for k, v in printed.items()},
             **sd0, **{"final::" + k[4:]: v for k, v in state_arrays(net).items()})
        print(f"  listops-style: step losses epoch0 {vals[0, :3]} ... printed {printed}")
    finally:
        torch.Tensor.cuda = real_cuda


CFG3_MODEL = dict(vocab_size=17, embedding_size=64, n_vec=2048, n_W=11, Ws=[128, 'GELU'], V=[128, 'GELU'], n_channels_V=64,
                  n_class=10, pooling_type="CLS", head=['linear'], use_cuda=False, use_residuals=False, dropout1_p=0,
                  dropout2_p=0, dropout3_p=0, init_embedding_weights=False, use_pos_embedding=True, problem="listops")


def gen_training_lra_cfg3(lra, utils):
    """BASELINE.json configs[2] at its own size — "LRA ListOps N=2048, dim=64, full PSF model training loop": the
    reference's TrainPSF (LRA/psf_utils.py:48-128) on its LRA PSFNet with the ListOps configuration
    (LRA/psf_training_config.py:2-30: 11 factors, hidden 128, CLS pooling, padding_idx embedding, positional embedding, no
    residual) at embedding_size = n_channels_V = 64 and N = 1 + 2047 (the CLS token prepended as listops_training.py:65-72
    does). 1 epoch x 4 fixed batches of 4, evaluation on 1 + 1 batches, CrossEntropyLoss, Adam(1e-3), seed_everything(42).
    The initial state is pinned by hash (seed_everything(42) + construction reproduces it), the end state by its small tensors."""
    import contextlib
    import io
    import re
    from torch.utils.data import DataLoader
    cfg = CFG3_MODEL
    BATCH, NB, NE = 4, 4, 1
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        utils.seed_everything(42)
        net = lra.PSFNet(**cfg)
        sd0 = state_arrays(net)
        optimizer = torch.optim.Adam(net.parameters(), lr=0.001)
        loss = RecordingLoss(torch.nn.CrossEntropyLoss())
        g = torch.Generator().manual_seed(2026)

        def split(n):
            data = torch.randint(0, 15, (n, cfg["n_vec"] - 1), generator=g)
            labels = torch.randint(0, 10, (n,), generator=g)
            cls_token = torch.tensor([[cfg['vocab_size'] - 1] * data.size(0)]).T  # listops_training.py:65-72
            return data, torch.cat([cls_token, data], -1), labels

        (raw_tr, Xtr, Ytr), (raw_va, Xva, Yva), (raw_te, Xte, Yte) = split(BATCH * NB), split(BATCH * NE), split(BATCH * NE)
        mk = lambda X, Y: DataLoader(utils.DatasetCreator(X, Y), batch_size=BATCH, shuffle=False, drop_last=True,  # noqa: E731
                                     num_workers=0)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            utils.TrainPSF(net=net, trainloader=mk(Xtr, Ytr), valloader=mk(Xva, Yva), testloader=mk(Xte, Yte),
                           n_epochs=1, test_freq=1, optimizer=optimizer, loss=loss, problem="listops", saving_criteria=1e9)
        text = buf.getvalue()
        num = r"([-+0-9.eE]+)"
        printed = {key: [float(x) for x in re.findall(pat + num, text)]
                   for key, pat in (("train_loss", r"Training loss:\s+"), ("val_loss", r"Val  loss: "),
                                    ("test_loss", r"Test loss: "), ("val_acc", r"Val  accuracy: "),
                                    ("test_acc", r"Test accuracy: "))}
        assert len(loss.values) == NB + 2 * NE and all(len(v) == 1 for v in printed.values())
        vals = np.asarray(loss.values, dtype=np.float64)
        u8 = lambda t: t.numpy().astype(np.uint8)  # noqa: E731
        end = state_arrays(net)
        small = {"final::" + k[4:]: v for k, v in end.items() if v.size <= 2048}  # biases, second layers of the link MLPs, head
        save("train_lra_listops_cfg3_n2048.npz", raw_tr=u8(raw_tr), Ytr=u8(Ytr), raw_va=u8(raw_va), Yva=u8(Yva), raw_te=u8(raw_te),
             Yte=u8(Yte), step_loss=vals[:NB].copy(), val_batch_loss=vals[NB:NB + NE].copy(), test_batch_loss=vals[NB + NE:].copy(),
             **{"printed_" + k: np.asarray(v) for k, v in printed.items()},
             sd0_names=np.asarray([k[4:] for k in sorted(sd0)]), sd0_sha256=np.asarray([sha256(sd0[k]) for k in sorted(sd0)]),
             end_names=np.asarray([k[4:] for k in sorted(end)]), end_absmax=np.asarray([float(np.abs(end[k]).max()) for k in sorted(end)]),
             end_sum=np.asarray([float(end[k].astype(np.float64).sum()) for k in sorted(end)]), **small)
        print(f"  cfg3 (N = 2048, E = C = 64): step losses {vals[:NB]}; printed {printed}")
    finally:
        torch.Tensor.cuda = real_cuda


# ---------------------------------------------------------------------------------------------------
# 4. attention_block.py
# ---------------------------------------------------------------------------------------------------
def gen_attention_block():
    """The stand-alone block (attention_block.py:70-178): n_W = ceil(log2(max_seq_len)), C = E, output [B, N, E].
    The module constructs and prints a 1024-token network when imported (181-192); that one is not used."""
    import contextlib
    import io
    sys.modules.setdefault("torch_geometric", types.ModuleType("torch_geometric"))  # imported, never used (line 9)
    with contextlib.redirect_stdout(io.StringIO()):
        ab = import_reference("ref_attention_block", "attention_block.py")
    for tag, seed, cfg, B in (
            ("n300_e32_res", 46, dict(vocab_size=50, embedding_size=32, max_seq_len=300, use_cuda=False,
                                      use_residuals=True, dropout1_p=0, dropout2_p=0, dropout3_p=0), 2),
            ("n64_e16", 47, dict(vocab_size=11, embedding_size=16, max_seq_len=64, use_cuda=False,
                                 use_residuals=False, dropout1_p=0, dropout2_p=0, dropout3_p=0), 3)):
        torch.manual_seed(seed)
        net = ab.PSFNet(**cfg)
        net.n_vec = cfg["max_seq_len"]  # run_and_capture reshapes with net.n_vec
        g = torch.Generator().manual_seed(seed + 100)
        x = torch.randint(0, cfg["vocab_size"], (B, cfg["max_seq_len"]), generator=g)
        gout = torch.randn(B, cfg["max_seq_len"], cfg["embedding_size"], generator=g)
        res = run_and_capture(net, x, lambda out: (out * gout).sum(), vfinal_module=net.dropout3)
        save(f"attention_block_{tag}.npz", x=x.numpy(), gout=gout.numpy(), out=res.pop("logits"), **res,
             **state_arrays(net), chord_indicies=net.chord_indicies.numpy(),
             n_W=np.asarray(net.n_W), n_links=np.asarray(net.n_links))


# ---------------------------------------------------------------------------------------------------
# 5. Genome_Clf/psf.py (the LRA model without `problem`)
# ---------------------------------------------------------------------------------------------------
def gen_genome(gen):
    torch.manual_seed(48)
    cfg = dict(vocab_size=6, embedding_size=16, n_vec=320, n_W=9, Ws=[16, 'GELU'], V=[16, 'GELU'], n_channels_V=16,
               n_class=2, pooling_type="FLATTEN", head=['linear'], use_cuda=False, use_residuals=True, dropout1_p=0,
               dropout2_p=0, dropout3_p=0, init_embedding_weights=True, use_pos_embedding=True)
    net = gen.PSFNet(**cfg)
    net.eval()
    g = torch.Generator().manual_seed(16)
    x = torch.randint(0, 6, (2, 320), generator=g)
    res = run_and_capture(net, x, None, vfinal_module=net.dropout3)
    save("genome_n320.npz", x=x.numpy(), logits=res["logits"], V0=res["V0"], Vfin=res["Vfin"], W=res["W"],
         **state_arrays(net))


def gen_training_genome(gen, utils):
    """The genome classification loop: the reference's TrainPSF of Genome_Clf/psf_utils.py:48-151 — the LRA loop plus
    clip_grad_norm_(max_norm=1.0) between backward and step (:73) and a ROC-AUC of the hard predictions in both evaluation
    loops (:109-126) — on its own PSFNet (Genome_Clf/psf.py:63-240) configured as genome_training_config.py:2-22 (vocab 6,
    E = C = 32, FLATTEN pooling, linear head, no residual, no positional embedding; dropouts 0 for a device-independent
    trajectory) at a small size: N = 320, n_W = 9, 2 epochs x 6 fixed batches of 8, evaluation on 2 + 2 batches,
    CrossEntropyLoss(reduction="sum") — so that the gradient norms exceed 1 and the clip acts —, Adam(1e-4 as :115),
    seed_everything(42). Inputs: seeded torch.randint tokens / labels."""
    import contextlib
    import io
    import re
    from torch.utils.data import DataLoader
    cfg = dict(vocab_size=6, embedding_size=32, n_vec=320, n_W=9, Ws=[32, 'GELU'], V=[32, 'GELU'], n_channels_V=32,
               n_class=2, pooling_type="FLATTEN", head=['linear'], use_cuda=False, use_residuals=False, dropout1_p=0,
               dropout2_p=0, dropout3_p=0, init_embedding_weights=False, use_pos_embedding=False)
    BATCH, NB, NE, EPOCHS = 8, 6, 2, 2
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        utils.seed_everything(42)
        net = gen.PSFNet(**cfg)
        sd0 = state_arrays(net)
        optimizer = torch.optim.Adam(net.parameters(), lr=0.0001)
        # reduction="sum": with the mean the gradient norms of this small network stay below 1 and the clip would be a no-op
        loss = RecordingLoss(torch.nn.CrossEntropyLoss(reduction="sum"))
        g = torch.Generator().manual_seed(2027)
        # gradient norms before clipping, as clip_grad_norm_ returns them (the loop discards the value)
        norms = []
        real_clip = torch.nn.utils.clip_grad_norm_

        def recording_clip(parameters, max_norm, *a, **k):
            total = real_clip(parameters, max_norm, *a, **k)
            norms.append(float(total))
            return total

        def split(n):
            return torch.randint(0, 5, (n, cfg["n_vec"]), generator=g), torch.randint(0, 2, (n,), generator=g)

        (Xtr, Ytr), (Xva, Yva), (Xte, Yte) = split(BATCH * NB), split(BATCH * NE), split(BATCH * NE)
        mk = lambda X, Y: DataLoader(utils.DatasetCreator(X, Y), batch_size=BATCH, shuffle=False, drop_last=True,  # noqa: E731
                                     num_workers=0)
        buf = io.StringIO()
        torch.nn.utils.clip_grad_norm_ = recording_clip
        try:
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                utils.TrainPSF(net=net, trainloader=mk(Xtr, Ytr), valloader=mk(Xva, Yva), testloader=mk(Xte, Yte),
                               n_epochs=EPOCHS, test_freq=1, optimizer=optimizer, loss=loss, saving_criteria=1e9)
        finally:
            torch.nn.utils.clip_grad_norm_ = real_clip
        text = buf.getvalue()
        num = r"([-+0-9.eE]+)"
        printed = {key: [float(x) for x in re.findall(pat + num, text)]
                   for key, pat in (("train_loss", r"Training loss:\s+"), ("val_loss", r"Val  loss: "),
                                    ("test_loss", r"Test loss: "), ("val_acc", r"Val  accuracy: "),
                                    ("test_acc", r"Test accuracy: "), ("val_rocauc", r"Val  ROCAUC: "),
                                    ("test_rocauc", r"Test ROCAUC: "))}
        per_epoch = NB + 2 * NE
        assert len(loss.values) == EPOCHS * per_epoch and all(len(v) == EPOCHS for v in printed.values())
        assert len(norms) == EPOCHS * NB and max(norms) > 1.0, norms  # the clip is exercised
        vals = np.asarray(loss.values, dtype=np.float64).reshape(EPOCHS, per_epoch)
        u8 = lambda t: t.numpy().astype(np.uint8)  # noqa: E731
        save("train_genome_n320.npz", Xtr=u8(Xtr), Ytr=u8(Ytr), Xva=u8(Xva), Yva=u8(Yva), Xte=u8(Xte), Yte=u8(Yte),
             step_loss=vals[:, :NB].copy(), val_batch_loss=vals[:, NB:NB + NE].copy(), test_batch_loss=vals[:, NB + NE:].copy(),
             grad_norm=np.asarray(norms, dtype=np.float64).reshape(EPOCHS, NB),
             **{"printed_" + k: np.asarray(v) for k, v in printed.items()},
             **sd0, **{"final::" + k[4:]: v for k, v in state_arrays(net).items()})
        print(f"  genome: step losses epoch0 {vals[0, :3]} ... grad norms {norms[:3]} ... printed {printed}")
    finally:
        torch.Tensor.cuda = real_cuda


# ---------------------------------------------------------------------------------------------------
# 6. the reference's data generators
# ---------------------------------------------------------------------------------------------------
def gen_synth_data(gen):
    """adding() and temporal_order() of SyntheticExperiments/synth_data_generation.py:8-70, run as they are (seeded):
    samples of the distributions the on-device generators must match, and the reference's own labels for them."""
    import contextlib
    import io
    import random
    random.seed(42)
    torch.manual_seed(42)
    with contextlib.redirect_stderr(io.StringIO()):  # tqdm
        xa, ya = gen.adding(512, 64)
        xo, yo = gen.temporal_order(2048, 32)
    save("synth_data_reference_samples.npz", adding_data=xa.numpy(), adding_labels=ya.numpy(),
         order_data=xo.numpy().astype(np.uint8), order_labels=yo.numpy().astype(np.uint8))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)  # deterministic CPU reductions
    only = sys.argv[1] if len(sys.argv) > 1 else None  # "cfg3" / "genome_train": only that fixture
    se = import_reference("ref_se_psf", "SyntheticExperiments/psf.py")
    lra = import_reference("ref_lra_psf", "LRA/psf.py")
    if only is None:
        gen_indices(se)
        gen_synthetic(se)
        gen_lra(lra)
        gen_training(se, import_reference("ref_se_psf_utils", "SyntheticExperiments/psf_utils.py"))
    lra_utils = import_reference("ref_lra_psf_utils", "LRA/psf_utils.py")
    if only in (None, "cfg3"):
        gen_training_lra_cfg3(lra, lra_utils)
    if only in (None, "genome_train"):
        gen_training_genome(import_reference("ref_genome_psf", "Genome_Clf/psf.py"),
                            import_reference("ref_genome_psf_utils", "Genome_Clf/psf_utils.py"))
    if only is not None:
        return
    gen_training_lra(lra, lra_utils)
    gen_attention_block()
    gen_synth_data(import_reference("ref_synth_data_generation", "SyntheticExperiments/synth_data_generation.py"))
    gen_genome(import_reference("ref_genome_psf", "Genome_Clf/psf.py"))


if __name__ == "__main__":
    main()
