"""Host logic of the callers of the hot path: synthetic data generators and the
train / eval harness. CPU for the logic, one GPU test that trains the real model for a few steps.
"""
import os

import pytest
import torch
from torch.utils.data import DataLoader

from sparsefactorization_amd import synth_data
from sparsefactorization_amd.train import DatasetCreator, TrainModel, count_params, evaluate, seed_everything


def _device(request, name):
    """'cpu', or the `gpu` fixture (skips without a GPU): the generators are written once for both and the
    distribution checks run on the device the benchmark and the drivers actually generate on."""
    return torch.device("cpu") if name == "cpu" else request.getfixturevalue("gpu")


DEVICES = ["cpu", pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("where", DEVICES)
def test_adding_distribution(request, where):
    """synth_data_generation.py:8-28: x in (-1,1), exactly two distinct markers, label 0.5 + (x1+x2)/4."""
    dev = _device(request, where)
    g = torch.Generator(device=dev).manual_seed(0)
    data, labels = synth_data.adding(5000, 64, device=dev, generator=g)
    assert data.device.type == dev.type
    data, labels = data.cpu(), labels.cpu()
    assert data.shape == (5000, 64, 2) and data.dtype == torch.float32 and labels.shape == (5000,)
    x, y = data[..., 0], data[..., 1]
    assert float(x.min()) >= -1 and float(x.max()) <= 1 and abs(float(x.mean())) < 0.01
    assert torch.all((y == 0) | (y == 1)) and torch.all(y.sum(1) == 2)
    assert torch.allclose(labels, 0.5 + (x * y).sum(1) / 4, atol=1e-6)
    # marker positions are uniform over unordered pairs: first marker index mean = (N-2)/3 + ...
    first = y.argmax(1).float().mean()
    assert abs(float(first) - (64 - 2) / 3) < 1.0


@pytest.mark.parametrize("where", DEVICES)
def test_temporal_order_distribution(request, where):
    """synth_data_generation.py:30-70: tokens 0..3, two ordered special positions from {4,5}, 4 classes."""
    dev = _device(request, where)
    g = torch.Generator(device=dev).manual_seed(1)
    data, labels = synth_data.temporal_order(4000, 32, device=dev, generator=g)
    assert data.device.type == dev.type
    data, labels = data.cpu(), labels.cpu()
    assert data.shape == (4000, 32, 1) and data.dtype == torch.int64
    x = data[..., 0]
    special = x >= 4
    assert torch.all(special.sum(1) == 2) and int(x.max()) <= 5 and int(x.min()) >= 0
    idx = special.float().argsort(dim=1, descending=True, stable=True)[:, :2].sort(dim=1).values
    v1, v2 = x.gather(1, idx[:, :1]).squeeze(1), x.gather(1, idx[:, 1:]).squeeze(1)
    assert torch.equal(labels, 2 * (v1 == 5).long() + (v2 == 5).long())
    counts = torch.bincount(labels, minlength=4).float() / 4000
    assert torch.all((counts - 0.25).abs() < 0.04)
    assert torch.all((torch.bincount(x[~special], minlength=4).float() / (~special).sum() - 0.25).abs() < 0.01)


def test_generators_are_seeded():
    a = synth_data.adding(10, 16, generator=torch.Generator().manual_seed(5))
    b = synth_data.adding(10, 16, generator=torch.Generator().manual_seed(5))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(8 * 2, 1)

    def forward(self, x):
        return self.lin(x.reshape(x.size(0), -1))


def test_train_model_loop_and_checkpoint_naming(tmp_path):
    """TrainModel semantics (psf_utils.py:48-137) on a toy regressor: loss falls, 0.04-tolerance accuracy,
    checkpoint named {problem}_epoch{e}_acc{a}.pt once the threshold is beaten."""
    seed_everything(3)
    g = torch.Generator().manual_seed(3)
    X, _ = synth_data.adding(512, 8, generator=g)
    Y = X.reshape(512, -1) @ torch.linspace(-0.2, 0.2, 16)  # learnable target
    mk = lambda: DataLoader(DatasetCreator(X, Y), batch_size=32, shuffle=True, drop_last=True)  # noqa: E731
    net = _Toy()
    assert count_params(net) == 17
    opt = torch.optim.Adam(net.parameters(), lr=0.05)
    logs = []
    hist = TrainModel(net, mk(), mk(), mk(), n_epochs=8, test_freq=2, optimizer=opt, loss=torch.nn.MSELoss(),
                      problem="adding", saving_criteria=50.0, save_dir=str(tmp_path), log=logs.append)
    assert hist[-1]["train"]["loss"] < 0.1 * hist[0]["train"]["loss"]
    assert [h["epoch"] for h in hist if "test" in h] == [0, 2, 4, 6]
    saved = [f for f in os.listdir(tmp_path) if f.startswith("adding_epoch") and f.endswith(".pt")]
    assert saved, logs
    sd = torch.load(tmp_path / saved[0])
    assert set(sd) == {"lin.weight", "lin.bias"}
    ev = evaluate(net, mk(), torch.nn.MSELoss(), "adding")
    assert ev["accuracy"] > 50.0


def test_order_accuracy_is_argmax():
    class Const(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            out = torch.zeros(x.size(0), 4)
            out[:, 2] = 1.0
            return out + self.p
    Y = torch.tensor([2, 2, 0, 1])
    loader = DataLoader(DatasetCreator(torch.zeros(4, 3, 1), Y), batch_size=2)
    ev = evaluate(Const(), loader, torch.nn.CrossEntropyLoss(), "order")
    assert ev["accuracy"] == 50.0


def test_lra_driver_host_logic():
    """CLS prepend (listops_training.py:65-72), synthetic token ranges, config -> model on the CPU."""
    from sparsefactorization_amd import lra_training
    X, Y = lra_training.synthetic_split("listops", 6, "cpu", 0)
    assert X.shape == (6, 1999) and X.dtype == torch.int64 and int(X.max()) < 15 and Y.shape == (6,)
    Xc = lra_training.add_cls_token(X, 17)
    assert Xc.shape == (6, 2000) and torch.all(Xc[:, 0] == 16) and torch.equal(Xc[:, 1:], X)
    Xp, _ = lra_training.synthetic_split("pathfinder", 2, "cpu", 0)
    assert Xp.shape == (2, 1024) and int(Xp.max()) < 225
    net = lra_training.build_model("cifar10", use_cuda=False)
    assert net.n_W == 10 and net.final[0].in_features == 1024 * 16 and net.dropout3.p == 0.8
    assert lra_training.build_model("imdb", use_cuda=False, n_vec=129, n_W=7).embedding.padding_idx == 95


@pytest.mark.gpu
def test_lra_training_driver_runs_on_gpu(gpu, capsys):
    from sparsefactorization_amd import lra_training
    import json
    import math
    lra_training.main(["--task", "listops", "--train-seqs", "160", "--eval-seqs", "32", "--json", "--max-steps", "3"])
    out = capsys.readouterr().out
    rec = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    # the reference configuration at full size (N = 2000, E = 512, 128 channels, batch 32): numbers, not substrings
    assert rec["task"] == "listops" and rec["n_vec"] == 2000 and rec["batch_per_gpu"] == 32 and rec["steps"] == 3
    assert math.isfinite(rec["loss"]) and 0.5 < rec["loss"] < 20.0  # ten classes, random labels: around ln 10 = 2.3
    assert rec["ms_per_step"] > 0 and math.isfinite(rec["value"]) and rec["value"] > 0
    lra_training.main(["--task", "pathfinder", "--train-seqs", "128", "--eval-seqs", "64", "--epochs", "1"])
    out = capsys.readouterr().out
    assert "Training loss" in out and "Test accuracy" in out


@pytest.mark.gpu
def test_training_driver_runs_on_gpu(gpu, capsys):
    from sparsefactorization_amd import psf_training
    psf_training.main(["--problem", "order", "--n-vec", "512", "--train-seqs", "400", "--eval-seqs", "80",
                       "--json", "--max-steps", "6"])
    out = capsys.readouterr().out
    assert '"metric": "PSF train tokens/sec"' in out
    psf_training.main(["--problem", "adding", "--n-vec", "256", "--train-seqs", "200", "--eval-seqs", "80",
                       "--epochs", "1"])
    out = capsys.readouterr().out
    assert "Training loss" in out and "Test accuracy" in out


def test_device_batches_has_dataloader_semantics():
    """DeviceBatches vs DataLoader(DatasetCreator, batch_size, shuffle, drop_last): same number of batches, same
    shapes, every sample at most once per epoch, sequential order when not shuffled, a new order every epoch."""
    from sparsefactorization_amd.train import DeviceBatches
    X, Y = torch.arange(23, dtype=torch.float32).reshape(23, 1), torch.arange(23)
    for drop_last in (True, False):
        ref = DataLoader(DatasetCreator(X, Y), batch_size=5, shuffle=False, drop_last=drop_last)
        mine = DeviceBatches(X, Y, 5, shuffle=False, drop_last=drop_last)
        assert len(mine) == len(ref)
        for (xa, ya), (xb, yb) in zip(mine, ref):
            assert torch.equal(xa, xb) and torch.equal(ya, yb)
    sh = DeviceBatches(X, Y, 5, shuffle=True, drop_last=True)
    e1 = torch.cat([y for _, y in sh])
    e2 = torch.cat([y for _, y in sh])
    assert e1.numel() == 20 and e1.unique().numel() == 20 and not torch.equal(e1, e2)
    for x, y in sh:
        assert torch.equal(x.squeeze(1).long(), y)  # samples and labels stay paired


def test_make_adam_is_plain_adam_on_cpu():
    from sparsefactorization_amd.train import make_adam
    lin = torch.nn.Linear(3, 2)
    opt = make_adam(lin.parameters(), 0.01)
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 0.01 and not opt.defaults.get("fused")


def _golden_samples():
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synth_data_reference_samples.npz")
    return np.load(path)


def test_label_rules_reproduce_the_reference_generators_labels():
    """Samples drawn by the REFERENCE's adding() / temporal_order() (synth_data_generation.py:8-70, run by
    oracle/gen_golden.py): the label rules this package's generators use — 0.5 + (x1 + x2) / 4 and
    2 [first == Y] + [second == Y] — give the reference's labels on the reference's data, exactly."""
    g = _golden_samples()
    xa, ya = torch.from_numpy(g["adding_data"]), torch.from_numpy(g["adding_labels"])
    x, y = xa[..., 0], xa[..., 1]
    assert torch.all((y == 0) | (y == 1)) and torch.all(y.sum(1) == 2)
    pos = y.argsort(dim=1, descending=True, stable=True)[:, :2].sort(dim=1).values
    x1, x2 = x.gather(1, pos[:, :1]).squeeze(1), x.gather(1, pos[:, 1:]).squeeze(1)
    assert torch.equal(0.5 + (x1 + x2) / 4, ya)
    xo, yo = torch.from_numpy(g["order_data"].astype("int64"))[..., 0], torch.from_numpy(g["order_labels"].astype("int64"))
    special = xo >= 4
    assert torch.all(special.sum(1) == 2)
    idx = special.float().argsort(dim=1, descending=True, stable=True)[:, :2].sort(dim=1).values
    v1, v2 = xo.gather(1, idx[:, :1]).squeeze(1), xo.gather(1, idx[:, 1:]).squeeze(1)
    assert torch.equal(2 * (v1 == 5).long() + (v2 == 5).long(), yo)


@pytest.mark.parametrize("where", DEVICES)
def test_generators_match_the_reference_samples_in_distribution(request, where):
    """Two-sample comparison of this package's generators (CPU and GPU) with the reference's own output: value range
    and mean, marker-gap and first-marker statistics, token and class frequencies."""
    dev = _device(request, where)
    g = _golden_samples()
    gen = torch.Generator(device=dev).manual_seed(11)
    # Adding, N = 64
    ref = torch.from_numpy(g["adding_data"])
    mine, labels = synth_data.adding(4096, 64, device=dev, generator=gen)
    mine, labels = mine.cpu(), labels.cpu()
    assert float(mine[..., 0].min()) >= -1 and float(mine[..., 0].max()) <= 1
    assert abs(float(mine[..., 0].mean()) - float(ref[..., 0].mean())) < 0.02
    assert abs(float(mine[..., 0].std()) - float(ref[..., 0].std())) < 0.02

    def marker_stats(data):
        pos = data[..., 1].argsort(dim=1, descending=True, stable=True)[:, :2].sort(dim=1).values.float()
        return pos[:, 0].mean(), pos[:, 1].mean(), (pos[:, 1] - pos[:, 0]).mean()
    for a, b in zip(marker_stats(mine), marker_stats(ref)):  # uniform unordered pair on 64: means 20.3 / 42.7 / 21.7
        assert abs(float(a) - float(b)) < 2.5  # the reference sample has 512 sequences: sigma of its means ~0.65
    assert abs(float(labels.mean()) - float(torch.from_numpy(g["adding_labels"]).mean())) < 0.03
    # Temporal order, N = 32
    ref_o = torch.from_numpy(g["order_data"].astype("int64"))[..., 0]
    ref_y = torch.from_numpy(g["order_labels"].astype("int64"))
    mine_o, mine_y = synth_data.temporal_order(8192, 32, device=dev, generator=gen)
    mine_o, mine_y = mine_o.cpu()[..., 0], mine_y.cpu()
    f_ref = torch.bincount(ref_o.flatten(), minlength=6).float() / ref_o.numel()
    f_mine = torch.bincount(mine_o.flatten(), minlength=6).float() / mine_o.numel()
    assert torch.all((f_ref - f_mine).abs() < 0.01)
    c_ref = torch.bincount(ref_y, minlength=4).float() / ref_y.numel()
    c_mine = torch.bincount(mine_y, minlength=4).float() / mine_y.numel()
    assert torch.all((c_ref - c_mine).abs() < 0.04)
    first = lambda x: (x >= 4).float().argmax(1).float().mean()  # noqa: E731
    assert abs(float(first(ref_o)) - float(first(mine_o))) < 1.0


def test_ranks_draw_different_training_data_from_the_same_model_seed():
    """Data parallel: every rank builds the SAME model (seed_everything(42), then rank 0's weights are broadcast) and
    draws ITS OWN data (seed 1000 + rank in psf_training.train_benchmark / 100 + rank in the drivers)."""
    from sparsefactorization_amd.psf_training import make_split
    x0, y0 = make_split("order", 8, 64, "cpu", 1000 + 0)
    x1, y1 = make_split("order", 8, 64, "cpu", 1000 + 1)
    x0b, _ = make_split("order", 8, 64, "cpu", 1000 + 0)
    assert torch.equal(x0, x0b) and not torch.equal(x0, x1)
    from sparsefactorization_amd import lra_training
    a, _ = lra_training.synthetic_split("listops", 4, "cpu", 100 + 0)
    b, _ = lra_training.synthetic_split("listops", 4, "cpu", 100 + 1)
    assert not torch.equal(a, b)
