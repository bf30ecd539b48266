#!/usr/bin/env python3
"""PSFNet forward end to end (Adding, N = 16384, B = 64, W through memory and W in the step) and the Order training step with
two builds of the library in one process, arms interleaved:   python profiles/e2e_two_libs_ab.py other/libpsf_chord.so"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsefactorization_amd import _lib, fused_mixer, psf_training  # noqa: E402
from sparsefactorization_amd.train import make_adam  # noqa: E402

dev = torch.device("cuda:0")
new = _lib.load()
old = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for fname, (argtypes, restype) in _lib.SIGNATURES.items():
    fn = getattr(old, fname, None)
    if fn is not None:
        fn.argtypes, fn.restype = argtypes, restype
torch.manual_seed(42)
net = psf_training.build_model("adding", 16384).to(dev).eval()
X, _ = psf_training.make_split("adding", 64, 16384, dev, 1000)
tnet = psf_training.build_model("order", 16384).to(dev)
opt = make_adam(tnet.parameters(), 1e-3)
TX, TY = psf_training.make_split("order", 40, 16384, dev, 1000)
loss = torch.nn.CrossEntropyLoss()


def fwd(route):
    fused_mixer.route = route
    with torch.no_grad():
        return net(X)


def step():
    opt.zero_grad(set_to_none=True)
    out = loss(tnet(TX).squeeze(), TY)
    out.backward()
    opt.step()


def timed(fn, n):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {}
for rd in range(5):
    for name, lib in (("this", new), ("other", old)):
        _lib._lib = lib
        res.setdefault((name, "fwd W through memory"), []).append(timed(lambda: fwd("never"), 20))
        res.setdefault((name, "fwd W in the step"), []).append(timed(lambda: fwd("always"), 20))
        res.setdefault((name, "order training step"), []).append(timed(step, 20))
_lib._lib = new
fused_mixer.route = "auto"
for what in ("fwd W through memory", "fwd W in the step", "order training step"):
    print(f"{what:24s} this {statistics.median(res[('this', what)]):.4f} ms   other {statistics.median(res[('other', what)]):.4f} ms", flush=True)
