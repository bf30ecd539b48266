"""Pins of the callers of the hot path to outputs of the REFERENCE's own code, run in the build container by
oracle/gen_golden.py (fixtures in tests/golden):

  train_{adding,order}_n128.npz      the reference's TrainModel (SyntheticExperiments/psf_utils.py:48-137) on its
                                     PSFNet at cfg1, seed_everything(42), Adam(1e-3), 2 epochs x 8 fixed batches of 40:
                                     loss after every optimiser step, evaluation losses / accuracies it printed
  attention_block_*.npz              attention_block.py:70-178 PSFNet: operands of its hot loop, output, gradients

CPU tests: the oracle and the host-side construction logic against those fixtures. GPU tests (`-m gpu`): this
package's train harness / attention block, through the HIP kernels, against the same numbers.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_state_dict, load_golden, rel_inf
from oracle import chord_oracle as oc

CFG1 = {
    "adding": dict(vocab_size=1, add_init_linear_layer=True, embedding_size=32, n_vec=128, n_W=7, Ws=[32, 'GELU'],
                   V=[32, 'GELU'], n_channels_V=8, n_class=1, pooling_type="FLATTEN", head=['linear'],
                   use_residuals=True, use_pos_embedding=False, problem="adding"),
    "order": dict(vocab_size=6, add_init_linear_layer=False, embedding_size=32, n_vec=128, n_W=7, Ws=[32, 'GELU'],
                  V=[32, 'GELU'], n_channels_V=8, n_class=4, pooling_type="FLATTEN", head=['linear'],
                  use_residuals=True, use_pos_embedding=True, problem="order"),
}
BLOCKS = {
    "attention_block_n300_e32_res.npz": (46, dict(vocab_size=50, embedding_size=32, max_seq_len=300, use_residuals=True,
                                                  dropout1_p=0, dropout2_p=0, dropout3_p=0)),
    "attention_block_n64_e16.npz": (47, dict(vocab_size=11, embedding_size=16, max_seq_len=64, use_residuals=False,
                                             dropout1_p=0, dropout2_p=0, dropout3_p=0)),
}


def _train_tensors(g, problem, device="cpu"):
    cast = (lambda a: torch.from_numpy(a.copy())) if problem == "adding" else (lambda a: torch.from_numpy(a.astype(np.int64)))
    return {k: cast(g[k]).to(device) for k in ("Xtr", "Ytr", "Xva", "Yva", "Xte", "Yte")}


# ------------------------------------------------------------------------------------------------------------
# CPU
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("problem", ["adding", "order"])
def test_training_fixture_is_the_reference_loop(problem):
    """What TrainModel printed per epoch is the mean of the per-step values the loss module returned
    (psf_utils.py:73-79: running_loss / len(trainloader)), and evaluation means likewise (92-121)."""
    g = load_golden(f"train_{problem}_n128.npz")
    assert g["step_loss"].shape == (2, 8) and g["val_batch_loss"].shape == (2, 2)
    assert np.allclose(g["step_loss"].mean(1), g["printed_train_loss"], rtol=1e-12)
    assert np.allclose(g["val_batch_loss"].mean(1), g["printed_val_loss"], rtol=1e-12)
    assert np.allclose(g["test_batch_loss"].mean(1), g["printed_test_loss"], rtol=1e-12)
    t = _train_tensors(g, problem)
    assert t["Xtr"].shape[0] == 320 and t["Xva"].shape[0] == 80 and t["Xte"].shape[0] == 80


@pytest.mark.parametrize("problem", ["adding", "order"])
def test_seed_42_draws_the_reference_initial_weights(problem):
    """Same construction order as SyntheticExperiments/psf.py:98-145 => seed_everything(42) gives the reference's
    initial state_dict bit for bit (what makes a training trajectory comparable at all)."""
    from sparsefactorization_amd.synthetic_psf import PSFNet
    from sparsefactorization_amd.train import seed_everything
    g = load_golden(f"train_{problem}_n128.npz")
    seed_everything(42)
    net = PSFNet(**CFG1[problem], use_cuda=False)
    want = golden_state_dict(g)
    have = net.state_dict()
    assert list(have.keys()) == list(want.keys())
    for k in want:
        assert torch.equal(have[k], want[k]), k


@pytest.mark.parametrize("fixture", sorted(BLOCKS))
def test_attention_block_construction_matches_reference(fixture):
    from sparsefactorization_amd.attention_block import PSFNet
    seed, cfg = BLOCKS[fixture]
    g = load_golden(fixture)
    torch.manual_seed(seed)
    net = PSFNet(**cfg, use_cuda=False)
    assert net.n_W == int(g["n_W"]) and net.n_links == int(g["n_links"])
    assert np.array_equal(net.chord_indicies.numpy(), g["chord_indicies"])
    want = golden_state_dict(g)
    have = net.state_dict()
    assert list(have.keys()) == list(want.keys())
    for k in want:
        assert torch.equal(have[k], want[k]), k


@pytest.mark.parametrize("fixture", sorted(BLOCKS))
def test_oracle_matches_attention_block_loop(fixture):
    """Oracle chain and gradients vs the tensors captured around the reference block's hot loop (158-174)."""
    _, cfg = BLOCKS[fixture]
    g = load_golden(fixture)
    W, V0 = g["W"], g["V0"]
    M, B, N, L = W.shape
    res = cfg["use_residuals"]
    rows, cols = oc.chord_indices(N, L)
    steps = oc.chain(np.stack([rows, cols]), W, V0, res)
    assert rel_inf(steps[-1], g["Vfin"]) <= 1e-6
    assert rel_inf(steps[-1], g["out"]) <= 1e-6  # dropout3 with p = 0 is the identity
    off = oc.spmul_offsets(L) % N
    grad = g["gVfin"].copy()
    res_acc = np.zeros_like(V0)
    dW = np.zeros_like(W)
    for m in range(M - 1, -1, -1):
        x_in = V0 if m == 0 else steps[m - 1]
        if res:
            res_acc += grad
        dW[m], grad = oc.spmul_bwd(grad, W[m], x_in, off)
    assert rel_inf(dW, g["dW"]) <= 1e-5
    assert rel_inf(grad + res_acc, g["dV0"]) <= 1e-5


def test_bench_refuses_more_ranks_than_gpus_without_touching_a_gpu():
    """`python bench.py --gpus N` starts N ranks itself; with fewer GPUs than ranks the parent says so and returns
    non-zero before any rank (or any GPU call) is started. Here: no GPU at all, or a 1-GPU box."""
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(n, 2)), "--steps", "1"],
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0
    assert "GPU(s)" in proc.stderr and "one rank per GPU" in proc.stderr
    assert '{"metric"' not in proc.stdout


# ------------------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------------------
class _Recording(torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.inner, self.values = inner, []

    def forward(self, pred, target):
        out = self.inner(pred, target)
        self.values.append(out.detach().clone())
        return out


@pytest.mark.gpu
@pytest.mark.parametrize("problem", ["adding", "order"])
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_train_harness_follows_the_reference_trajectory(gpu, problem, mode):
    """train.TrainModel (eager, and replaying train.GraphedStep) on the fixture's batches reproduces the loss after
    each of the reference loop's 16 optimiser steps to 1e-4 relative, and the evaluation numbers it printed."""
    from sparsefactorization_amd.synthetic_psf import PSFNet
    from sparsefactorization_amd.train import DeviceBatches, GraphedStep, TrainModel, make_adam, seed_everything
    g = load_golden(f"train_{problem}_n128.npz")
    seed_everything(42)
    net = PSFNet(**CFG1[problem], use_cuda=True).to(gpu)
    for k, v in golden_state_dict(g).items():
        assert torch.equal(net.state_dict()[k].cpu(), v), k
    t = _train_tensors(g, problem, gpu)
    mk = lambda X, Y: DeviceBatches(X, Y, 40, shuffle=False, drop_last=True)  # noqa: E731
    loss = _Recording(torch.nn.MSELoss() if problem == "adding" else torch.nn.CrossEntropyLoss())
    optimizer = make_adam(net.parameters(), 0.001, capturable=mode == "graph")
    graphed = None
    if mode == "graph":
        graphed = GraphedStep(net, optimizer, loss.inner, t["Xtr"][:40], t["Ytr"][:40])
        steps = []
        real_call = graphed.__call__

        class _Tap:  # record the loss each replay returns (the graph holds loss.inner, not the recorder)
            def __call__(self, X, Y):
                out = real_call(X, Y)
                steps.append(out.detach().clone())
                return out
        tap = _Tap()
    logs = []
    hist = TrainModel(net=net, trainloader=mk(t["Xtr"], t["Ytr"]), valloader=mk(t["Xva"], t["Yva"]),
                      testloader=mk(t["Xte"], t["Yte"]), n_epochs=2, test_freq=1, optimizer=optimizer, loss=loss,
                      problem=problem, saving_criteria=1e9, log=logs.append, graphed=tap if graphed is not None else None)
    vals = np.asarray([float(v) for v in loss.values])
    if mode == "graph":
        got_steps = np.asarray([float(v) for v in steps]).reshape(2, 8)
        got_eval = vals.reshape(2, 4)
    else:
        got = vals.reshape(2, 12)
        got_steps, got_eval = got[:, :8], got[:, 8:]
    assert np.max(np.abs(got_steps - g["step_loss"]) / np.abs(g["step_loss"])) <= 1e-4, (got_steps, g["step_loss"])
    assert np.allclose(got_eval[:, :2], g["val_batch_loss"], rtol=2e-4)
    assert np.allclose(got_eval[:, 2:], g["test_batch_loss"], rtol=2e-4)
    for e in range(2):
        assert abs(hist[e]["train"]["loss"] - g["printed_train_loss"][e]) <= 1e-4 * g["printed_train_loss"][e]
        assert abs(hist[e]["val"]["loss"] - g["printed_val_loss"][e]) <= 2e-4 * g["printed_val_loss"][e]
        # accuracy is a count over 80 samples: allow one borderline sample (1.25 %)
        assert abs(hist[e]["val"]["accuracy"] - g["printed_val_acc"][e]) <= 1.25 + 1e-6
        assert abs(hist[e]["test"]["accuracy"] - g["printed_test_acc"][e]) <= 1.25 + 1e-6
    assert any("Training loss" in s for s in logs)
    # parameters after 16 Adam steps (each moves an element by at most lr = 1e-3)
    final = {k[7:]: g[k] for k in g.files if k.startswith("final::")}
    worst = max(float(np.max(np.abs(net.state_dict()[k].cpu().numpy() - v))) for k, v in final.items())
    assert worst <= 5e-4, worst


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", sorted(BLOCKS))
def test_attention_block_matches_reference(gpu, fixture):
    """attention_block.PSFNet on the GPU: module output vs the reference's (<= 1e-4, the MLPs round differently),
    the chain on the reference's captured operands vs its result (<= 1e-5; bit-equal to the oracle), gradients of
    the chain vs autograd through the reference (<= 1e-5)."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.attention_block import PSFNet
    _, cfg = BLOCKS[fixture]
    g = load_golden(fixture)
    net = PSFNet(**cfg, use_cuda=True)
    net.load_state_dict(golden_state_dict(g), strict=True)
    net = net.to(gpu).eval()
    with torch.no_grad():
        out = net(torch.from_numpy(g["x"]).to(gpu))
    assert out.shape == g["out"].shape and rel_inf(out.cpu().numpy(), g["out"]) <= 1e-4

    Ws = [torch.from_numpy(w.copy()).to(gpu).requires_grad_(True) for w in g["W"]]
    V0 = torch.from_numpy(g["V0"].copy()).to(gpu).requires_grad_(True)
    Vf = sfa.chord_chain(Ws, V0, cfg["use_residuals"])
    assert rel_inf(Vf.detach().cpu().numpy(), g["Vfin"]) <= 1e-5
    M, B, N, L = g["W"].shape
    rows, cols = oc.chord_indices(N, L)
    want = oc.chain(np.stack([rows, cols]), g["W"], g["V0"], cfg["use_residuals"])[-1]
    assert np.array_equal(Vf.detach().cpu().numpy(), want)
    Vf.backward(torch.from_numpy(g["gVfin"].copy()).to(gpu))
    dW = np.stack([w.grad.cpu().numpy() for w in Ws])
    assert rel_inf(dW, g["dW"]) <= 1e-5
    assert rel_inf(V0.grad.cpu().numpy(), g["dV0"]) <= 1e-5

    # and through the module: parameter gradients flow, the same loss as the fixture's
    net.train()
    out = net(torch.from_numpy(g["x"]).to(gpu))
    loss = (out * torch.from_numpy(g["gout"]).to(gpu)).sum()
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.gpu
def test_graphed_step_refuses_an_uncapturable_embedding(gpu):
    """A 600-token vocabulary is beyond the capturable table-gradient kernel (vocab <= 512): its gradient would run
    on aten::embedding_dense_backward, whose replay from a HIP graph faulted the GPU in round 1
    (profiles/r01_graph_step_lab.log). GraphedStep must raise BEFORE capturing anything; eager training still works."""
    from sparsefactorization_amd.lra_psf import PSFNet
    from sparsefactorization_amd.token_linear import embedding_wgrad
    from sparsefactorization_amd.train import GraphedStep, make_adam
    torch.manual_seed(0)
    net = PSFNet(vocab_size=600, embedding_size=16, n_vec=128, n_W=7, Ws=[16, 'GELU'], V=[16, 'GELU'], n_channels_V=8,
                 n_class=2, pooling_type="FLATTEN", head=['linear'], use_cuda=True, use_residuals=False, dropout1_p=0,
                 dropout2_p=0, dropout3_p=0, init_embedding_weights=False, use_pos_embedding=True,
                 problem="pathfinder").to(gpu)
    X = torch.randint(0, 600, (8, 128), device=gpu)
    Y = torch.randint(0, 2, (8,), device=gpu)
    loss = torch.nn.CrossEntropyLoss()
    opt = make_adam(net.parameters(), 1e-3, capturable=True)
    before = [p.detach().clone() for p in net.parameters()]
    with pytest.raises(RuntimeError, match="vocab <= 512"):
        GraphedStep(net, opt, loss, X, Y)
    assert not torch.cuda.is_current_stream_capturing()
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))  # nothing ran
    # the eager path is untouched by the refusal
    loss(net(X), Y).backward()
    assert net.embedding.weight.grad is not None and torch.isfinite(net.embedding.weight.grad).all()
    # and the operator itself refuses under capture instead of calling the aten fallback
    idx = torch.randint(0, 600, (4096,), device=gpu)
    dout = torch.randn(4096, 16, device=gpu)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="cannot be captured"):
        with torch.cuda.graph(graph):
            embedding_wgrad(idx, dout, 600, None)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_training_driver_under_a_one_rank_rccl_group(gpu, monkeypatch, capsys):
    """psf_training.main inside a real RCCL process group (one rank — all a 1-GPU box has) with the flat gradient
    all-reduce forced on: the collective path runs on the real PSFNet, replicas are broadcast, and parameters that
    never receive a gradient (Adding: `embedding`, `pos_embedding`, SyntheticExperiments/psf.py:98-107) stay at
    grad None through the reducer, in eager mode and with forward+backward replayed from a HIP graph."""
    import socket
    import torch.distributed as dist
    from sparsefactorization_amd import psf_training
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    for k, v in dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)).items():
        monkeypatch.setenv(k, v)
    try:
        for extra in ([], ["--graph"]):
            r = psf_training.main(["--problem", "adding", "--n-vec", "256", "--train-seqs", "240", "--eval-seqs", "80",
                                   "--json", "--max-steps", "4", "--force-allreduce", *extra])
            assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
            net, reducer = r["net"], r["reducer"]
            assert reducer is not None and reducer.calls >= 4
            assert net.embedding.weight.grad is None and net.pos_embedding.weight.grad is None
            assert net.final.weight.grad is not None and torch.isfinite(net.final.weight.grad).all()
            assert np.isfinite(r["stats"]["loss"])
            assert '"metric": "PSF train tokens/sec"' in capsys.readouterr().out
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_on_one_gpu(gpu):
    """The N-rank path of bench.py end to end — the parent starts torch.distributed.run as a child, two ranks, barriers,
    per-rank gathers, the training leg with the flat gradient all-reduce and forward+backward replayed from a HIP graph
    — on a 1-GPU box: PSF_BENCH_REHEARSAL=1 lets the ranks share the GPU and talk over gloo (RCCL refuses two ranks on
    one device). The numbers mean nothing; that the line is produced and well-formed is the test."""
    import json
    env = dict(os.environ, PSF_BENCH_REHEARSAL="1")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                           "--train-steps", "2", "--train-graph"], capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 128
    assert len(line["roofline"]["frac_per_gpu"]) == 2 and "rehearsal" in line and "cpu_baseline" not in line
    tr = line["train"]
    assert "error" not in tr, tr
    assert tr["global_batch"] == 80 and tr["allreduce_us"] > 0 and tr["allreduce_bytes"] > 4e6 and tr["hip_graph"] == "fwd+bwd"
    assert np.isfinite(tr["loss"]) and tr["value"] > 0


# ------------------------------------------------------------------------------------------------------------
# shipped checkpoints (IMDb, CIFAR-10) and the genome model
# ------------------------------------------------------------------------------------------------------------
IMDB_CKPT = dict(vocab_size=97, embedding_size=32, n_vec=4097, n_W=12, Ws=[128, 'GELU'], V=[128, 'GELU'], n_channels_V=32,
                 n_class=2, pooling_type="CLS", head=['linear'], use_residuals=True, dropout1_p=0.4, dropout2_p=0,
                 dropout3_p=0, init_embedding_weights=True, use_pos_embedding=False, problem="imdb")
CIFAR_CKPT = dict(vocab_size=256, embedding_size=16, n_vec=1024, n_W=10, Ws=[16, 'GELU'], V=[16, 'GELU'], n_channels_V=16,
                  n_class=10, pooling_type="FLATTEN", head=['non-linear', 16], use_residuals=False, dropout1_p=0,
                  dropout2_p=0.2, dropout3_p=0.8, init_embedding_weights=False, use_pos_embedding=True, problem="cifar10")
GENOME = dict(vocab_size=6, embedding_size=16, n_vec=320, n_W=9, Ws=[16, 'GELU'], V=[16, 'GELU'], n_channels_V=16, n_class=2,
              pooling_type="FLATTEN", head=['linear'], use_residuals=True, dropout1_p=0, dropout2_p=0, dropout3_p=0,
              init_embedding_weights=True, use_pos_embedding=True)


def _cpu_links(twin, g, use_pos):
    """W_m and V0 recomputed on the CPU from the stored weights exactly as the generator's reference run did."""
    torch.set_num_threads(1)
    with torch.no_grad():
        data = twin.embedding(torch.from_numpy(g["x"]))
        if use_pos:
            data = data + twin.pos_embedding.weight.unsqueeze(0)
        return twin.g(data), twin.link_weights(data)


def test_oracle_matches_genome_reference_loop():
    g = load_golden("genome_n320.npz")
    M, B, N, L = g["W"].shape
    rows, cols = oc.chord_indices(N, L)
    assert rel_inf(oc.chain(np.stack([rows, cols]), g["W"], g["V0"], True)[-1], g["Vfin"]) <= 1e-6


@pytest.mark.gpu
def test_genome_model_matches_reference(gpu):
    """Genome_Clf/psf.py:63-240 (the LRA model without `problem`): logits of the reference's run, chain bit-equal to the
    oracle on its captured operands."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.genome_psf import PSFNet
    g = load_golden("genome_n320.npz")
    net = PSFNet(**GENOME, use_cuda=True)
    net.load_state_dict(golden_state_dict(g), strict=True)
    net = net.to(gpu).eval()
    with torch.no_grad():
        logits = net(torch.from_numpy(g["x"]).to(gpu))
        Vf = sfa.chord_chain([torch.from_numpy(w.copy()).to(gpu) for w in g["W"]], torch.from_numpy(g["V0"].copy()).to(gpu), True)
    assert rel_inf(logits.cpu().numpy(), g["logits"]) <= 1e-4
    assert rel_inf(Vf.cpu().numpy(), g["Vfin"]) <= 1e-5
    M, B, N, L = g["W"].shape
    rows, cols = oc.chord_indices(N, L)
    assert np.array_equal(Vf.cpu().numpy(), oc.chain(np.stack([rows, cols]), g["W"], g["V0"], True)[-1])


@pytest.mark.gpu
def test_imdb_checkpoint_logits_chain_and_attention_map(gpu):
    """imdb_epoch138.pt (N = 4097 = 4096 + CLS, L = 13, C = 32, residual, CLS pooling): the chain on CPU-recomputed links
    vs the reference's V after its loop, the whole model's logits, and the dense 4097 x 4097 attention map of
    imdb_inference.py:41,53-59 (ChangedPSF) against rows and row sums of the reference op sequence's map."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.lra_psf import ChangedPSF, PSFNet
    g = load_golden("lra_imdb_ckpt.npz")
    sd = golden_state_dict(g)
    twin = PSFNet(**IMDB_CKPT, use_cuda=False)
    missing = twin.load_state_dict(sd, strict=False)
    assert missing.missing_keys == ["pos_embedding.weight"] and not missing.unexpected_keys  # unused, left out of the fixture
    twin.eval()
    V0, Ws = _cpu_links(twin, g, use_pos=False)
    assert rel_inf(V0.numpy()[:, ::4], g["V0_rows"]) <= 1e-6
    with torch.no_grad():
        Wg = [w.to(gpu) for w in Ws]
        Vf = sfa.chord_chain(Wg, V0.to(gpu), True).cpu().numpy()
        att = sfa.chord_chain(Wg, torch.eye(4097, device=gpu), False)
    assert rel_inf(Vf[:, ::4], g["Vfin_rows"]) <= 1e-5
    assert att.shape == (1, 4097, 4097)
    att = att.cpu().numpy()
    assert rel_inf(att[0, ::256, :], g["Wfinal_rows"]) <= 1e-5
    assert rel_inf(att.sum(-1), g["Wfinal_rowsum"]) <= 1e-5
    net = ChangedPSF(**IMDB_CKPT, use_cuda=True)
    net.load_state_dict(sd, strict=False)
    net = net.to(gpu).eval()
    with torch.no_grad():
        logits, att2 = net(torch.from_numpy(g["x"]).to(gpu))
    assert rel_inf(logits.cpu().numpy(), g["logits"]) <= 1e-4
    assert rel_inf(att2.cpu().numpy()[0, ::256, :], g["Wfinal_rows"]) <= 1e-4  # links from the GPU MLPs here


@pytest.mark.gpu
def test_cifar10_checkpoint_chain(gpu):
    """cifar10_epoch35.pt (N = 1024, L = 11, C = 16, no residual): chain on CPU-recomputed links vs the reference's V after
    its loop (the 16384 x 16 head is not part of the fixture), and the same through the module's `features`."""
    import sparsefactorization_amd as sfa
    from sparsefactorization_amd.lra_psf import PSFNet
    g = load_golden("lra_cifar10_ckpt.npz")
    sd = golden_state_dict(g)
    twin = PSFNet(**CIFAR_CKPT, use_cuda=False)
    missing = twin.load_state_dict(sd, strict=False)
    assert all(k.startswith("final.") for k in missing.missing_keys) and not missing.unexpected_keys
    twin.eval()
    V0, Ws = _cpu_links(twin, g, use_pos=True)
    assert rel_inf(V0.numpy(), g["V0"]) <= 1e-6
    with torch.no_grad():
        Vf = sfa.chord_chain([w.to(gpu) for w in Ws], V0.to(gpu), False).cpu().numpy()
    assert rel_inf(Vf, g["Vfin"]) <= 1e-5
    net = PSFNet(**CIFAR_CKPT, use_cuda=True)
    net.load_state_dict(sd, strict=False)
    net = net.to(gpu).eval()
    with torch.no_grad():
        V = net.features(torch.from_numpy(g["x"]).to(gpu))
    assert rel_inf(V.cpu().numpy(), g["Vfin"]) <= 1e-4


LRA_SMALL = dict(vocab_size=17, embedding_size=32, n_vec=128, n_W=7, Ws=[32, 'GELU'], V=[32, 'GELU'], n_channels_V=16,
                 n_class=10, pooling_type="CLS", head=['linear'], use_residuals=False, dropout1_p=0, dropout2_p=0,
                 dropout3_p=0, init_embedding_weights=False, use_pos_embedding=True, problem="listops")


def test_lra_seed_42_draws_the_reference_initial_weights():
    from sparsefactorization_amd.lra_psf import PSFNet
    from sparsefactorization_amd.train import seed_everything
    g = load_golden("train_lra_listops_n128.npz")
    seed_everything(42)
    net = PSFNet(**LRA_SMALL, use_cuda=False)
    want = golden_state_dict(g)
    assert list(net.state_dict().keys()) == list(want.keys())
    for k, v in net.state_dict().items():
        assert torch.equal(v, want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_lra_train_harness_follows_the_reference_trajectory(gpu, mode):
    """The LRA loop: train.TrainPSF + lra_training.add_cls_token (CLS id = vocab_size - 1 prepended,
    listops_training.py:65-72) on the fixture's raw token batches reproduces the 12 step losses of the reference's own
    TrainPSF (LRA/psf_utils.py:48-128) to 1e-4 relative and its evaluation losses; eager and HIP-graph replay."""
    from sparsefactorization_amd.lra_psf import PSFNet
    from sparsefactorization_amd.lra_training import add_cls_token
    from sparsefactorization_amd.train import DeviceBatches, GraphedStep, TrainPSF, make_adam, seed_everything
    g = load_golden("train_lra_listops_n128.npz")
    seed_everything(42)
    net = PSFNet(**LRA_SMALL, use_cuda=True).to(gpu)
    i64 = lambda k: torch.from_numpy(g[k].astype(np.int64)).to(gpu)  # noqa: E731
    data = {s: (add_cls_token(i64("raw_" + s), 17), i64("Y" + s)) for s in ("tr", "va", "te")}
    assert data["tr"][0].shape == (192, 128) and bool((data["tr"][0][:, 0] == 16).all())
    mk = lambda s: DeviceBatches(*data[s], 32, shuffle=False, drop_last=True)  # noqa: E731
    loss = _Recording(torch.nn.CrossEntropyLoss())
    optimizer = make_adam(net.parameters(), 0.001, capturable=mode == "graph")
    tap, steps = None, []
    if mode == "graph":
        graphed = GraphedStep(net, optimizer, loss.inner, data["tr"][0][:32], data["tr"][1][:32])

        def tap(X, Y):
            out = graphed(X, Y)
            steps.append(out.detach().clone())
            return out
    hist = TrainPSF(net=net, trainloader=mk("tr"), valloader=mk("va"), testloader=mk("te"), n_epochs=2, test_freq=1,
                    optimizer=optimizer, loss=loss, problem="listops", saving_criteria=1e9, log=lambda s: None, graphed=tap)
    vals = np.asarray([float(v) for v in loss.values])
    if mode == "graph":
        got_steps, got_eval = np.asarray([float(v) for v in steps]).reshape(2, 6), vals.reshape(2, 4)
    else:
        got = vals.reshape(2, 10)
        got_steps, got_eval = got[:, :6], got[:, 6:]
    assert np.max(np.abs(got_steps - g["step_loss"]) / np.abs(g["step_loss"])) <= 1e-4, (got_steps, g["step_loss"])
    assert np.allclose(got_eval[:, :2], g["val_batch_loss"], rtol=2e-4)
    assert np.allclose(got_eval[:, 2:], g["test_batch_loss"], rtol=2e-4)
    for e in range(2):
        assert abs(hist[e]["train"]["loss"] - g["printed_train_loss"][e]) <= 1e-4 * g["printed_train_loss"][e]
        assert abs(hist[e]["test"]["accuracy"] - g["printed_test_acc"][e]) <= 100.0 / 64 + 1e-6  # one borderline sample


CFG3 = dict(vocab_size=17, embedding_size=64, n_vec=2048, n_W=11, Ws=[128, 'GELU'], V=[128, 'GELU'], n_channels_V=64,
            n_class=10, pooling_type="CLS", head=['linear'], use_residuals=False, dropout1_p=0, dropout2_p=0, dropout3_p=0,
            init_embedding_weights=False, use_pos_embedding=True, problem="listops")


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_cfg3_seed_42_draws_the_reference_initial_weights():
    """BASELINE.json configs[2] at its own size (ListOps, N = 2048, dim = 64): the fixture holds the SHA-256 of every tensor
    of the reference network's initial state_dict; the same seed and constructor here reproduce them bit for bit."""
    from sparsefactorization_amd.lra_psf import PSFNet
    from sparsefactorization_amd.train import seed_everything
    g = load_golden("train_lra_listops_cfg3_n2048.npz")
    seed_everything(42)
    sd = PSFNet(**CFG3, use_cuda=False).state_dict()
    assert sorted(sd.keys()) == [str(n) for n in g["sd0_names"]]
    for name, want in zip(g["sd0_names"], g["sd0_sha256"]):
        assert _sha(sd[str(name)].numpy()) == str(want), name


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_cfg3_listops_training_follows_the_reference_at_full_size(gpu, mode):
    """cfg3 at BASELINE size — N = 1 + 2047 (CLS prepended, listops_training.py:65-72), E = C = 64, 11 factors, B = 4: the
    four step losses, the evaluation losses and the end state of the reference's own TrainPSF run (LRA/psf_utils.py:48-128,
    executed by oracle/gen_golden.py) are reproduced to 1e-4 by train.TrainPSF on the HIP path — embedding kernel, the
    WIDE producer-MLP kernels (E = 64, 64 channels), the LDS-resident chain and its backward kernels, fused Adam —
    eagerly and replaying one captured HIP graph."""
    from sparsefactorization_amd import fused_mlp
    from sparsefactorization_amd.lra_psf import PSFNet
    from sparsefactorization_amd.lra_training import add_cls_token
    from sparsefactorization_amd.train import DeviceBatches, GraphedStep, TrainPSF, make_adam, seed_everything
    g = load_golden("train_lra_listops_cfg3_n2048.npz")
    seed_everything(42)
    net = PSFNet(**CFG3, use_cuda=True).to(gpu)
    i64 = lambda k: torch.from_numpy(g[k].astype(np.int64)).to(gpu)  # noqa: E731
    data = {s: (add_cls_token(i64("raw_" + s), 17), i64("Y" + s)) for s in ("tr", "va", "te")}
    assert data["tr"][0].shape == (16, 2048) and bool((data["tr"][0][:, 0] == 16).all())
    mk = lambda s: DeviceBatches(*data[s], 4, shuffle=False, drop_last=True)  # noqa: E731
    loss = _Recording(torch.nn.CrossEntropyLoss())
    optimizer = make_adam(net.parameters(), 0.001, capturable=mode == "graph")
    wide_calls = []
    orig = fused_mlp.wide_apply
    fused_mlp.wide_apply = lambda x, b: (wide_calls.append(len(b)), orig(x, b))[1]
    try:
        tap, steps = None, []
        if mode == "graph":
            graphed = GraphedStep(net, optimizer, loss.inner, data["tr"][0][:4], data["tr"][1][:4])

            def tap(X, Y):
                out = graphed(X, Y)
                steps.append(out.detach().clone())
                return out
        hist = TrainPSF(net=net, trainloader=mk("tr"), valloader=mk("va"), testloader=mk("te"), n_epochs=1, test_freq=1,
                        optimizer=optimizer, loss=loss, problem="listops", saving_criteria=1e9, log=lambda s: None, graphed=tap)
    finally:
        fused_mlp.wide_apply = orig
    assert wide_calls and all(n == 12 for n in wide_calls)  # g + 11 link MLPs through psf_mlp_wide_*
    vals = np.asarray([float(v) for v in loss.values])
    got_steps = np.asarray([float(v) for v in steps]) if mode == "graph" else vals[:4]
    got_eval = vals[-2:]
    assert np.all(np.isfinite(got_steps))
    assert np.max(np.abs(got_steps - g["step_loss"]) / np.abs(g["step_loss"])) <= 1e-4, (got_steps, g["step_loss"])
    assert abs(got_eval[0] - g["val_batch_loss"][0]) <= 2e-4 * g["val_batch_loss"][0]
    assert abs(got_eval[1] - g["test_batch_loss"][0]) <= 2e-4 * g["test_batch_loss"][0]
    assert abs(hist[0]["train"]["loss"] - g["printed_train_loss"][0]) <= 1e-4 * g["printed_train_loss"][0]
    assert hist[0]["val"]["accuracy"] == g["printed_val_acc"][0] and hist[0]["test"]["accuracy"] == g["printed_test_acc"][0]
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    for name, absmax, total in zip(g["end_names"], g["end_absmax"], g["end_sum"]):
        v = sd[str(name)]
        assert abs(float(np.abs(v).max()) - absmax) <= 1e-3 * absmax + 1e-7, name
    for key in g.files:
        if key.startswith("final::"):
            assert rel_inf(sd[key[7:]], g[key]) <= 2e-3, key  # four Adam steps of lr 1e-3 on top of 1e-4-accurate gradients


# ----------------------------------------------------------------------------------------------------------------------
# Genome classification loop (Genome_Clf/psf_utils.py:48-151): gradient-norm clip + ROC-AUC
# ----------------------------------------------------------------------------------------------------------------------
GENOME_SMALL = dict(vocab_size=6, embedding_size=32, n_vec=320, n_W=9, Ws=[32, 'GELU'], V=[32, 'GELU'], n_channels_V=32,
                    n_class=2, pooling_type="FLATTEN", head=['linear'], use_residuals=False, dropout1_p=0, dropout2_p=0,
                    dropout3_p=0, init_embedding_weights=False, use_pos_embedding=False)


def test_binary_roc_auc_is_the_rank_statistic():
    """train.binary_roc_auc = sklearn.metrics.roc_auc_score on binary targets (the call of Genome_Clf/psf_utils.py:112,126):
    hand-checked cases, ties, hard predictions, and sklearn itself where it is installed."""
    from sparsefactorization_amd.train import binary_roc_auc
    assert binary_roc_auc([0, 0, 1, 1], [0.1, 0.4, 0.35, 0.8]) == pytest.approx(0.75)
    assert binary_roc_auc([0, 1], [0, 1]) == 1.0 and binary_roc_auc([0, 1], [1, 0]) == 0.0
    assert binary_roc_auc([0, 1, 0, 1], [1, 1, 1, 1]) == 0.5  # all tied
    y, p = [1, 1, 1, 0, 0, 0, 0, 1], [1, 0, 1, 0, 0, 1, 0, 1]  # hard predictions: (TPR + TNR) / 2 = (3/4 + 3/4) / 2
    assert binary_roc_auc(y, p) == pytest.approx(0.75)
    with pytest.raises(ValueError):
        binary_roc_auc([1, 1, 1], [0.2, 0.3, 0.4])
    try:
        from sklearn.metrics import roc_auc_score
    except Exception:
        return
    rng = np.random.default_rng(7)
    for n in (3, 16, 257):
        for _ in range(10):
            yy = rng.integers(0, 2, n)
            if yy.min() == yy.max():
                continue
            for ss in (rng.integers(0, 2, n), rng.normal(size=n), rng.integers(0, 4, n)):
                assert binary_roc_auc(yy, ss) == pytest.approx(roc_auc_score(yy, ss), abs=1e-12)


def test_genome_seed_42_draws_the_reference_initial_weights():
    from sparsefactorization_amd.genome_psf import PSFNet
    from sparsefactorization_amd.train import seed_everything
    g = load_golden("train_genome_n320.npz")
    seed_everything(42)
    net = PSFNet(**GENOME_SMALL, use_cuda=False)
    want = golden_state_dict(g)
    assert list(net.state_dict().keys()) == list(want.keys())
    for k, v in net.state_dict().items():
        assert torch.equal(v, want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_genome_train_harness_follows_the_reference_trajectory(gpu, mode, monkeypatch):
    """train.TrainGenomePSF on the fixture's batches reproduces what the reference's own TrainPSF
    (Genome_Clf/psf_utils.py:48-151) did on its own PSFNet: the 12 step losses to 1e-4 relative, the gradient norms its
    clip_grad_norm_(max_norm=1.0) saw (all above 1: the clip acts), the evaluation losses, accuracies and ROC-AUCs it
    printed, and the final parameters; eagerly and with the step (clip included) replayed from a HIP graph."""
    from sparsefactorization_amd.genome_psf import PSFNet
    from sparsefactorization_amd.train import DeviceBatches, GraphedStep, TrainGenomePSF, make_adam, seed_everything
    g = load_golden("train_genome_n320.npz")
    seed_everything(42)
    net = PSFNet(**GENOME_SMALL, use_cuda=True).to(gpu)
    i64 = lambda k: torch.from_numpy(g[k].astype(np.int64)).to(gpu)  # noqa: E731
    data = {s: (i64("X" + s), i64("Y" + s)) for s in ("tr", "va", "te")}
    mk = lambda s: DeviceBatches(*data[s], 8, shuffle=False, drop_last=True)  # noqa: E731
    loss = _Recording(torch.nn.CrossEntropyLoss(reduction="sum"))
    optimizer = make_adam(net.parameters(), 0.0001, capturable=mode == "graph")
    norms = []
    real_clip = torch.nn.utils.clip_grad_norm_

    def recording_clip(parameters, max_norm, *a, **k):
        assert max_norm == 1.0
        total = real_clip(parameters, max_norm, *a, **k)
        norms.append(total.detach().clone())
        return total

    tap, steps = None, []
    if mode == "graph":
        graphed = GraphedStep(net, optimizer, loss.inner, data["tr"][0][:8], data["tr"][1][:8], grad_clip_norm=1.0)
        assert graphed.grad_clip_norm == 1.0

        def tap(X, Y):
            out = graphed(X, Y)
            steps.append(out.detach().clone())
            return out
    else:
        monkeypatch.setattr(torch.nn.utils, "clip_grad_norm_", recording_clip)
    lines = []
    hist = TrainGenomePSF(net=net, trainloader=mk("tr"), valloader=mk("va"), testloader=mk("te"), n_epochs=2, test_freq=1,
                          optimizer=optimizer, loss=loss, saving_criteria=1e9, log=lines.append, graphed=tap)
    vals = np.asarray([float(v) for v in loss.values])
    if mode == "graph":
        got_steps, got_eval = np.asarray([float(v) for v in steps]).reshape(2, 6), vals.reshape(2, 4)
    else:
        got = vals.reshape(2, 10)
        got_steps, got_eval = got[:, :6], got[:, 6:]
        got_norms = np.asarray([float(v) for v in norms]).reshape(2, 6)
        assert g["grad_norm"].min() > 1.0
        assert np.max(np.abs(got_norms - g["grad_norm"]) / g["grad_norm"]) <= 1e-3, (got_norms, g["grad_norm"])
    assert np.max(np.abs(got_steps - g["step_loss"]) / np.abs(g["step_loss"])) <= 1e-4, (got_steps, g["step_loss"])
    assert np.allclose(got_eval[:, :2], g["val_batch_loss"], rtol=2e-4)
    assert np.allclose(got_eval[:, 2:], g["test_batch_loss"], rtol=2e-4)
    for e in range(2):
        assert abs(hist[e]["train"]["loss"] - g["printed_train_loss"][e]) <= 1e-4 * g["printed_train_loss"][e]
        assert hist[e]["val"]["accuracy"] == pytest.approx(g["printed_val_acc"][e], abs=1e-9)
        assert hist[e]["test"]["accuracy"] == pytest.approx(g["printed_test_acc"][e], abs=1e-9)
        assert 100.0 * hist[e]["val"]["rocauc"] == pytest.approx(g["printed_val_rocauc"][e], abs=1e-9)
        assert 100.0 * hist[e]["test"]["rocauc"] == pytest.approx(g["printed_test_rocauc"][e], abs=1e-9)
    assert sum(ln.startswith("Val  ROCAUC: ") for ln in lines) == 2 and sum(ln.startswith("Test ROCAUC: ") for ln in lines) == 2
    final = {k[len("final::"):]: v for k, v in g.items() if k.startswith("final::")}
    for k, v in net.state_dict().items():
        want = final[k]
        assert rel_inf(v.detach().cpu().numpy(), want) <= 2e-4, k
