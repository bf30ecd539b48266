"""Train / eval harness around PSFNet — the caller of the hot path.

Mirrors ``SyntheticExperiments/psf_utils.py`` (``seed_everything`` 10-20, ``DatasetCreator`` 23-43,
``count_params`` 44-46, ``TrainModel`` 48-137) and ``LRA/psf_utils.py`` (``TrainPSF`` 48-128: the same loop,
always arg-max accuracy): Adam steps on ``loss(net(X).squeeze(), Y)``, evaluation every ``test_freq`` epochs
over validation and test loaders, the Adding tolerance ``|pred - Y| < 0.04``, a state_dict checkpoint named
``{problem}_epoch{e}_acc{a}.pt`` whenever test accuracy beats ``saving_criteria``.

Differences, all on purpose: no ``.cuda()`` hard-wiring (tensors go to the model's device, so the loop runs
unchanged on one process per GPU); an optional gradient reducer (``dp.FlatGradAllReduce``) is called between
``backward()`` and ``step()`` for batch-sharded data parallelism; metrics are accumulated on the device and
read back once per epoch instead of one ``.item()`` sync per batch (psf_utils.py:73).
"""
from __future__ import annotations

import os
import random
import time
from typing import Callable, Dict, Iterable, Optional

import numpy as np
import torch
from torch.utils.data import Dataset


def seed_everything(seed: int = 1234) -> None:
    """Seed python, numpy and torch (CPU + GPU) — psf_utils.py:10-20."""
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class DatasetCreator(Dataset):
    """(sample, target) pairs from two indexable tensors — psf_utils.py:23-43."""

    def __init__(self, data, labels):
        self.data = data
        self.labels = labels

    def __getitem__(self, index):
        return self.data[index], self.labels[index]

    def __len__(self):
        return len(self.labels)


class DeviceBatches:
    """``DataLoader(DatasetCreator(data, labels), batch_size, shuffle, drop_last=True, num_workers=0)`` for tensors
    that already live on the GPU (psf_training.py:84-86 loads the whole split up front as well): a batch is ONE
    index_select of a shuffled index instead of ``batch_size`` per-sample reads plus a ``torch.stack`` — the
    default collate costs 0.3-0.5 ms of host time per step, which is the step time of the small LRA models.
    Same batches-per-epoch, same "every sample at most once per epoch, order reshuffled each epoch" contract."""

    def __init__(self, data: torch.Tensor, labels: torch.Tensor, batch_size: int, shuffle: bool = False,
                 drop_last: bool = True, generator: Optional[torch.Generator] = None):
        self.data, self.labels = data, labels
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.generator = generator
        self.dataset = DatasetCreator(data, labels)  # DataLoader-like attribute

    def __len__(self):
        n = len(self.labels)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n, bs = len(self.labels), self.batch_size
        if self.shuffle:
            perm = torch.randperm(n, generator=self.generator).to(self.data.device)  # host RNG, as DataLoader's sampler
        for i in range(len(self)):
            lo, hi = i * bs, min((i + 1) * bs, n)
            if self.shuffle:
                idx = perm[lo:hi]
                yield self.data.index_select(0, idx), self.labels.index_select(0, idx)
            else:
                yield self.data[lo:hi], self.labels[lo:hi]


class ChunkedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(params, lr, betas, eps)`` (no weight decay / amsgrad / maximize) on ``psf_adam_step_f32``
    (csrc/adam.hip): one launch per <= 40 parameters with 4096-element chunks per workgroup. PyTorch's fused multi-tensor
    Adam gives a workgroup 65 536 elements, which runs the two 524 288-element parameters of a PSFNet at N = 16384
    (``pos_embedding``, ``final``) on 16 workgroups: 2 x 43 us per step against ~2 x 5 here.

    State per parameter under torch's Adam's names (``step``, ``exp_avg``, ``exp_avg_sq``); ``state_dict`` /
    ``load_state_dict`` / ``torch.save`` work as for any optimizer (the host-side launch plan — ctypes pointer tables —
    lives outside ``param_groups`` and is rebuilt whenever a parameter, a gradient set or a state tensor changes address).

    ``capturable``: ``step`` is ONE device scalar shared by all parameters of a group (advanced by a one-element kernel,
    read by the update kernel), so a captured step can be replayed. That is torch's Adam as long as every parameter that
    ever gets a gradient gets one on EVERY step (PSFNet: parameters the forward does not use never get one and are simply
    skipped); a parameter that joins or leaves the set of updated parameters later would take another parameter's bias
    correction, so that raises instead. A non-capturable instance raises when stepped under stream capture (the host step
    count would be baked into the graph), like torch's Adam."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, capturable: bool = False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, capturable=capturable))
        self._plans: Dict[int, tuple] = {}       # group index -> (key, pointer tables..., states)
        self._step_devs: Dict[int, torch.Tensor] = {}  # group index -> the shared device step counter (capturable)
        self._cap_live: Dict[int, tuple] = {}    # group index -> ids of the parameters the shared counter counts for
        self._live_states: Dict[int, tuple] = {}  # group index -> (ids of the parameters with a gradient, their state dicts)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans.clear()
        self._step_devs.clear()
        self._cap_live.clear()
        self._live_states.clear()
        for gi, group in enumerate(self.param_groups):  # re-link the loaded per-parameter counters to ONE device scalar
            if not group["capturable"]:
                for p in group["params"]:
                    st = self.state.get(p)
                    if st and torch.is_tensor(st.get("step")):
                        st["step"] = float(st["step"])
                continue
            steps = [self.state[p]["step"] for p in group["params"] if self.state.get(p)]
            if not steps:
                continue
            vals = {float(t) for t in steps}
            if len(vals) != 1:
                raise RuntimeError("ChunkedAdam(capturable=True) shares one step counter per group; the loaded state has "
                                   f"different counts {sorted(vals)}")
            p0 = next(p for p in group["params"] if self.state.get(p))
            dev_step = torch.full((), vals.pop(), dtype=torch.float32, device=p0.device)
            self._step_devs[gi] = dev_step
            for p in group["params"]:
                if self.state.get(p):
                    self.state[p]["step"] = dev_step

    def zero_grad(self, set_to_none: bool = True):
        """As torch's (gradients dropped, not zeroed, by default) without its per-call bookkeeping: 75 -> 15 us of host time
        per step on a 50-tensor model, where the host is the bound (profiles/lra_host_profile.py)."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes
        from . import _lib
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            live, grads = [], []
            for p in group["params"]:  # (one look at .grad per parameter: it is a property that goes through the dispatcher)
                g = p.grad
                if g is not None:
                    live.append(p)
                    grads.append(g)
            if not live:
                continue
            dev = live[0].device
            cap = group["capturable"]
            if not cap and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ChunkedAdam: stepping under stream capture needs capturable=True (the host step count "
                                   "would be baked into the captured kernel arguments)")
            if cap:
                if gi not in self._step_devs:
                    self._step_devs[gi] = torch.zeros((), dtype=torch.float32, device=dev)
                ids = tuple(map(id, live))
                if self._cap_live.setdefault(gi, ids) != ids:
                    raise RuntimeError("ChunkedAdam(capturable=True): the set of parameters with a gradient changed between "
                                       "steps; the shared step counter would give the newcomers another parameter's bias "
                                       "correction. Use capturable=False (per-parameter counts) for such a model.")
            # (self.state is keyed by tensors: a look-up hashes through Python — the state dicts of an unchanged set of live
            # parameters are remembered per group)
            live_ids = tuple(map(id, live))
            seen = self._live_states.get(gi)
            # (every entry re-validated by identity — 8 us for 60 parameters: a user who clears, deletes or replaces ONE
            # parameter's state between steps, `opt.state[p].clear()` / `del opt.state[p]`, gets fresh moments for it, and
            # no update ever lands in a dict that self.state no longer holds)
            if (seen is not None and seen[0] == live_ids
                    and all(st and self.state.get(p) is st for p, st in zip(live, seen[1]))):
                states = seen[1]
            else:
                for p in live:
                    st = self.state[p]
                    if not st:
                        st["step"] = self._step_devs[gi] if cap else 0.0
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                states = [self.state[p] for p in live]
                self._live_states[gi] = (live_ids, states)
            # host-side plan (pointer tables of the parameters and their moments), rebuilt only when a tensor of it
            # moved: the small LRA models are launch-bound, every microsecond here counts
            key = tuple([(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) for p, st in zip(live, states)])
            plan = self._plans.get(gi)
            if plan is None or plan[0] != key:
                n = len(live)
                tab = lambda ts: (ctypes.c_void_p * n)(*[x.data_ptr() for x in ts])  # noqa: E731
                plan = (key, tab(live), tab([st["exp_avg"] for st in states]), tab([st["exp_avg_sq"] for st in states]),
                        (ctypes.c_int64 * n)(*[p.numel() for p in live]), states)
                self._plans[gi] = plan
            _, ptab, mtab, vtab, sizes, _ = plan
            n = len(live)
            grads = [g if g.is_contiguous() else g.contiguous() for g in grads]  # (copies stay alive until the launch is queued)
            gtab = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
            b1, b2 = group["betas"]
            stream = _lib.stream_ptr(dev)
            with torch.cuda.device(dev):
                if cap:
                    self._step_devs[gi].add_(1.0)
                    rc = lib.psf_adam_step_f32(ptab, gtab, mtab, vtab, sizes, n, group["lr"], b1, b2, group["eps"], 1.0,
                                               self._step_devs[gi].data_ptr(), stream)
                else:
                    t0 = states[0]["step"] + 1.0
                    uniform = True
                    for st in states:
                        st["step"] += 1.0
                        uniform = uniform and st["step"] == t0
                    if uniform:
                        rc = lib.psf_adam_step_f32(ptab, gtab, mtab, vtab, sizes, n, group["lr"], b1, b2, group["eps"], t0,
                                                   None, stream)
                    else:  # parameters that skipped steps (grad None) have their own count: one call per tensor
                        rc = 0
                        for i, st in enumerate(states):
                            one = lambda tb: (ctypes.c_void_p * 1)(tb[i])  # noqa: E731
                            rc = rc or lib.psf_adam_step_f32(one(ptab), one(gtab), one(mtab), one(vtab),
                                                             (ctypes.c_int64 * 1)(sizes[i]), 1, group["lr"], b1, b2,
                                                             group["eps"], st["step"], None, stream)
            _lib.check(rc, "psf_adam_step_f32")
        return loss


def make_adam(params, lr: float, capturable: bool = False) -> torch.optim.Optimizer:
    """``optim.Adam(net.parameters(), lr=...)`` of psf_training.py:50 / listops_training.py:84. On the GPU (contiguous
    fp32 parameters) the same update on ``ChunkedAdam``; otherwise torch's own Adam. ``capturable``: the step counter
    lives on the device, so the step can be captured in a HIP graph (``GraphedStep``)."""
    params = list(params)
    if params and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params):
        return ChunkedAdam(params, lr=lr, capturable=capturable)
    return torch.optim.Adam(params, lr=lr)


def count_params(net: torch.nn.Module) -> int:
    return sum(p.numel() for p in net.parameters() if p.requires_grad)


def _device_of(net: torch.nn.Module) -> torch.device:
    return next(net.parameters()).device


def _count_correct(pred: torch.Tensor, Y: torch.Tensor, problem: str) -> torch.Tensor:
    if problem == 'adding':
        return (torch.abs(pred.squeeze() - Y) < 0.04).sum()  # psf_utils.py:103
    return pred.max(1)[1].eq(Y).sum()                         # psf_utils.py:105-106


def binary_roc_auc(targets, scores) -> float:
    """Area under the ROC curve of ``scores`` for binary ``targets`` (what sklearn.metrics.roc_auc_score returns for them):
    the Mann-Whitney statistic with average ranks for ties. Genome_Clf/psf_utils.py:112,126 call it on the HARD predictions,
    where it equals (true-positive rate + true-negative rate) / 2. Raises, like sklearn, when only one class is present."""
    import numpy as np
    y = np.asarray(targets).astype(np.int64).ravel()
    s = np.asarray(scores, dtype=np.float64).ravel()
    n_pos = int((y == 1).sum())
    n_neg = int(y.size - n_pos)
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(s, kind="mergesort")
    ranks = np.empty(s.size, dtype=np.float64)
    sorted_s = s[order]
    i = 0
    while i < s.size:  # average rank of each run of equal scores
        j = i
        while j + 1 < s.size and sorted_s[j + 1] == sorted_s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y == 1].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


@torch.no_grad()
def evaluate(net, loader: Iterable, loss, problem: str, roc_auc: bool = False) -> Dict[str, float]:
    """Mean loss and accuracy (%) over a loader, as the validation / test loops of TrainModel (92-121). ``roc_auc``: also
    the ROC-AUC of the hard predictions over the whole loader, as Genome_Clf/psf_utils.py:95-126 (key "rocauc", in [0, 1];
    the predictions and targets behind it as "predictions" / "targets")."""
    dev = _device_of(net)
    was_training = net.training
    net.eval()
    total = 0
    loss_sum = torch.zeros((), device=dev)
    correct = torch.zeros((), device=dev)
    batches = 0
    kept_pred, kept_y = [], []
    for X, Y in loader:
        X, Y = X.to(dev, non_blocking=True), Y.to(dev, non_blocking=True)
        pred = net(X)
        loss_sum += loss(pred.squeeze(), Y)
        correct += _count_correct(pred, Y, problem)
        if roc_auc:
            kept_pred.append(pred.max(1)[1])
            kept_y.append(Y)
        total += Y.size(0)
        batches += 1
    net.train(was_training)
    out = {"loss": float(loss_sum) / max(batches, 1), "accuracy": 100.0 * float(correct) / max(total, 1)}
    if roc_auc:
        out["predictions"] = torch.cat(kept_pred).cpu().numpy() if kept_pred else []
        out["targets"] = torch.cat(kept_y).cpu().numpy() if kept_y else []
        out["rocauc"] = binary_roc_auc(out["targets"], out["predictions"])
    return out


def check_capturable(net: torch.nn.Module) -> None:
    """Raise before a capture if a token embedding of ``net`` would send its weight gradient through
    ``aten::embedding_dense_backward``: that operator sizes a rocprim partition from a host read-back, which cannot be
    captured, and REPLAYING such a capture faulted the GPU (profiles/r01_graph_step_lab.log). The capturable kernel,
    ``psf_embed_tokens_bwd_f32``, holds the table in LDS: vocab <= EMB_MAX_VOCAB, width <= EMB_MAX_WIDTH, fp32."""
    from .token_linear import EMB_MAX_VOCAB, EMB_MAX_WIDTH, TokenEmbedding
    for name, m in net.named_modules():
        if isinstance(m, TokenEmbedding) and m.weight.requires_grad:
            if m.num_embeddings > EMB_MAX_VOCAB or m.embedding_dim > EMB_MAX_WIDTH or m.weight.dtype != torch.float32:
                raise RuntimeError(
                    f"GraphedStep: embedding '{name}' ({m.num_embeddings} x {m.embedding_dim}, {m.weight.dtype}) is "
                    f"outside the capturable table-gradient kernel's limits (vocab <= {EMB_MAX_VOCAB}, width <= "
                    f"{EMB_MAX_WIDTH}, float32); its gradient would run on aten::embedding_dense_backward, which cannot "
                    "be captured in a HIP graph. Train this model eagerly (no --graph).")


class GraphedStep:
    """The training step of psf_utils.py:62-71 (zero_grad, forward, loss, backward, optimizer.step) captured ONCE in a
    HIP graph and replayed per batch: for the small LRA models a step is ~150 kernel launches for ~1 ms of GPU work,
    i.e. launch-bound from Python; a replay costs one launch. Fixed batch shape (``drop_last=True`` loaders).
    Every kernel of this package takes its sizes from host arguments, never from device data (the token-embedding
    gradient included — nn.Embedding's own backward is not capturable, see ``check_capturable``), so the captured step
    is shape-static.

    Single process (``reducer`` None): the whole step including the optimizer is one graph; the optimizer must be built
    with ``make_adam(..., capturable=True)``. Data parallel (``reducer`` = ``dp.FlatGradAllReduce``): the graph holds
    zero_grad + forward + loss + backward; the gradient all-reduce (a collective must not sit in a private capture of
    one rank's stream) and the optimizer step run eagerly after each replay, on gradients that live at fixed addresses
    in the graph's memory pool."""

    def __init__(self, net, optimizer, loss, X: torch.Tensor, Y: torch.Tensor, warmup_steps: int = 3,
                 reducer: Optional[Callable[[], None]] = None, capture_error_mode: Optional[str] = None,
                 grad_clip_norm: Optional[float] = None):
        """``capture_error_mode`` (``torch.cuda.graph``): None = "thread_local" when a reducer is given — with a process
        group alive its watchdog thread queries events while this thread captures, and in "global" mode a call from ANY
        thread can invalidate the capture — else "global" (the strictest check, nothing else runs in a single process)."""
        if not X.is_cuda:
            raise RuntimeError("GraphedStep needs GPU tensors")
        check_capturable(net)
        if reducer is None and not optimizer.defaults.get("capturable", False):
            raise RuntimeError("GraphedStep captures optimizer.step(): build the optimizer with capturable=True "
                               "(make_adam(..., capturable=True)); a host-side step count would be frozen into the graph")
        self.net, self.optimizer, self.loss, self.reducer = net, optimizer, loss, reducer
        # clip_grad_norm_(max_norm) between backward (and the all-reduce) and the step — Genome_Clf/psf_utils.py:73. Its
        # norm, coefficient and scaling are device operations with no read-back: capturable.
        self.grad_clip_norm = grad_clip_norm
        self.X, self.Y = X.clone(), Y.clone()
        # a plain nn.Embedding that is actually looked up (not just a parameter holder like pos_embedding, whose
        # .weight is added directly) has the same uncapturable gradient: watch for calls during the warm-up
        called, hooks = [], []
        for name, m in net.named_modules():
            if type(m) is torch.nn.Embedding and m.weight.requires_grad:
                hooks.append(m.register_forward_hook(lambda _m, _i, _o, _n=name: called.append(_n)))
        # the warm-up steps are real optimisation steps; they are undone after the capture (parameters, buffers and
        # optimizer state restored IN PLACE — the graph holds their addresses), so a graphed run starts from the same
        # state, and follows the same trajectory, as an eager one (tests/test_reference_pins.py; with dropout too: the
        # generator states are put back, and a replay advances the philox offset by what an eager step draws)
        with torch.no_grad():
            tensors = list(net.parameters()) + list(net.buffers())
            snapshot = [t.detach().clone() for t in tensors]
        opt_before = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in optimizer.state[p].items()}
                      for p in optimizer.state}
        # warm-up and capture draw random numbers (dropout): the generator states are restored below too
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(X.device)
        # Whatever happens below — a refused capture included — the warm-up's optimisation steps and random draws are undone
        # (try / finally): a caller that falls back to eager stepping after a failed capture (psf_training.train_benchmark
        # with several ranks) then continues from the same state as the ranks whose capture worked.
        side = torch.cuda.Stream(device=X.device)
        try:
            side.wait_stream(torch.cuda.current_stream(X.device))
            with torch.cuda.stream(side):  # eager warm-up on a side stream (allocator, lazy initialisations, autotuning)
                for _ in range(warmup_steps):
                    self._step()
            torch.cuda.current_stream(X.device).wait_stream(side)
            for h in hooks:
                h.remove()
            hooks = []
            if called:
                raise RuntimeError(f"GraphedStep: nn.Embedding module(s) {sorted(set(called))} are looked up in forward; their "
                                   "gradient (aten::embedding_dense_backward) cannot be captured in a HIP graph. Use "
                                   "token_linear.TokenEmbedding (same parameters and state_dict) or train eagerly.")
            self.graph = torch.cuda.CUDAGraph()
            optimizer.zero_grad(set_to_none=True)
            if capture_error_mode is None:
                capture_error_mode = "thread_local" if reducer is not None else "global"
            self.capture_error_mode = capture_error_mode
            with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
                self.output = self._step(zero=False, eager_tail=False)
            self.warmup_steps = warmup_steps
        finally:
            for h in hooks:
                h.remove()
            torch.cuda.current_stream(X.device).wait_stream(side)  # (a warm-up step that raised left work on the side stream)
            with torch.no_grad():
                for t, s0 in zip(tensors, snapshot):
                    t.copy_(s0)
                for p, st in optimizer.state.items():
                    before = opt_before.get(id(p), {})
                    for k, v in list(st.items()):
                        if torch.is_tensor(v):
                            v.copy_(before[k]) if k in before else v.zero_()
                        elif k == "step":  # a host-side step count (ChunkedAdam, not capturable)
                            st[k] = before.get(k, 0.0)
            torch.set_rng_state(rng_cpu)
            torch.cuda.set_rng_state(rng_dev, X.device)

    def _step(self, zero: bool = True, eager_tail: bool = True):
        if zero:
            self.optimizer.zero_grad(set_to_none=True)
        out = self.loss(self.net(self.X).squeeze(), self.Y)
        out.backward()
        if self.reducer is None:
            self._clip()
            self.optimizer.step()
        elif eager_tail:
            self.reducer()
            self._clip()
            self.optimizer.step()
        return out.detach()

    def _clip(self):
        if self.grad_clip_norm is not None:
            torch.nn.utils.clip_grad_norm_(self.net.parameters(), max_norm=self.grad_clip_norm)

    def __call__(self, X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
        """One optimisation step on (X, Y); returns the loss (a static tensor, overwritten by the next call)."""
        self.X.copy_(X, non_blocking=True)
        self.Y.copy_(Y, non_blocking=True)
        self.graph.replay()
        if self.reducer is not None:
            self.reducer()
            self._clip()
            self.optimizer.step()
        return self.output


def train_epoch(net, loader: Iterable, optimizer, loss, reducer: Optional[Callable[[], None]] = None,
                max_steps: Optional[int] = None, graphed: Optional[GraphedStep] = None,
                grad_clip_norm: Optional[float] = None) -> Dict[str, float]:
    """One pass of the training loop (psf_utils.py:60-74). Returns mean loss, steps and seconds. With ``graphed``
    every step is a replay of that captured step (same net / optimizer / loss; a data-parallel ``GraphedStep`` calls
    its own reducer, ``reducer`` here is then ignored; likewise its own ``grad_clip_norm``)."""
    dev = _device_of(net)
    running = torch.zeros((), device=dev)
    steps = 0
    t0 = time.perf_counter()
    for X, Y in loader:
        X, Y = X.to(dev, non_blocking=True), Y.to(dev, non_blocking=True)
        if graphed is not None:
            running += graphed(X, Y)
            steps += 1
            if max_steps is not None and steps >= max_steps:
                break
            continue
        optimizer.zero_grad(set_to_none=True)
        pred = net(X)
        output = loss(pred.squeeze(), Y)
        output.backward()
        if reducer is not None:
            reducer()
        if grad_clip_norm is not None:  # Genome_Clf/psf_utils.py:73; after the all-reduce, so every rank clips the same gradient
            torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=grad_clip_norm)
        optimizer.step()
        running += output.detach()
        steps += 1
        if max_steps is not None and steps >= max_steps:
            break
    mean = float(running) / max(steps, 1)  # the only host sync of the epoch
    return {"loss": mean, "steps": steps, "seconds": time.perf_counter() - t0}


def TrainModel(net, trainloader, valloader, testloader, n_epochs, test_freq, optimizer, loss, problem,
               saving_criteria, reducer: Optional[Callable[[], None]] = None, save_dir: str = ".",
               log: Callable[[str], None] = print, is_main: bool = True, graphed: Optional[GraphedStep] = None):
    """Same arguments and behaviour as ``TrainModel`` (SyntheticExperiments/psf_utils.py:48-137); for LRA use
    ``problem`` = the task name (arg-max accuracy, as ``TrainPSF``). Returns the per-epoch history. ``graphed``: a
    ``GraphedStep`` over the same net / optimizer / loss replaces the eager step (evaluation stays eager)."""
    history = []
    for epoch in range(n_epochs):
        stats = train_epoch(net, trainloader, optimizer, loss, reducer, graphed=graphed)
        if is_main:
            log("Epoch {} - Training loss:  {} — Time:  {}sec".format(epoch, stats["loss"], stats["seconds"]))
        record = {"epoch": epoch, "train": stats}
        if epoch % test_freq == 0:
            val = evaluate(net, valloader, loss, problem)
            test = evaluate(net, testloader, loss, problem)
            record.update(val=val, test=test)
            if is_main:
                log("Val  loss: {}".format(val["loss"]))
                log("Test loss: {}".format(test["loss"]))
                log("Val  accuracy: {}".format(val["accuracy"]))
                log("Test accuracy: {}".format(test["accuracy"]))
                log('_' * 40)
                if test["accuracy"] > saving_criteria:
                    path = os.path.join(save_dir, '{}_epoch{}_acc{}.pt'.format(problem, epoch, test["accuracy"]))
                    torch.save(net.state_dict(), path)
                    record["checkpoint"] = path
        history.append(record)
    return history


TrainPSF = TrainModel  # LRA/psf_utils.py:48 name


def TrainGenomePSF(net, trainloader, valloader, testloader, n_epochs, test_freq, optimizer, loss, saving_criteria,
                   reducer: Optional[Callable[[], None]] = None, save_dir: str = ".", log: Callable[[str], None] = print,
                   is_main: bool = True, graphed: Optional[GraphedStep] = None, grad_clip_norm: float = 1.0):
    """``TrainPSF`` of Genome_Clf/psf_utils.py:48-151 — same arguments (it has no ``problem``): the LRA loop with
    ``clip_grad_norm_(net.parameters(), max_norm=1.0)`` between backward and step (:73) and the ROC-AUC of the hard
    predictions in both evaluation loops (:95-126), printed in per cent (:134-135). A ``graphed`` step must have been built
    with the same ``grad_clip_norm``. When the test accuracy exceeds ``saving_criteria`` the state_dict, the test
    predictions and the test targets are saved (:138-151; the reference's file-name pattern there has more fields than
    arguments and cannot format — the names here are genome_psf_epoch{E}_acc{A}.pt / _predictions.pt / _targets.pt)."""
    if isinstance(graphed, GraphedStep) and graphed.grad_clip_norm != grad_clip_norm:
        raise ValueError(f"the graphed step clips at {graphed.grad_clip_norm}, this loop at {grad_clip_norm}")
    history = []
    for epoch in range(n_epochs):
        if is_main:
            log(str(len(trainloader)))  # :64
        stats = train_epoch(net, trainloader, optimizer, loss, reducer, graphed=graphed, grad_clip_norm=grad_clip_norm)
        if is_main:
            log("Epoch {} - Training loss:  {} — Time:  {}sec".format(epoch, stats["loss"], stats["seconds"]))
        record = {"epoch": epoch, "train": stats}
        if epoch % test_freq == 0:
            val = evaluate(net, valloader, loss, "genome", roc_auc=True)
            test = evaluate(net, testloader, loss, "genome", roc_auc=True)
            record.update(val={k: v for k, v in val.items() if k not in ("predictions", "targets")},
                          test={k: v for k, v in test.items() if k not in ("predictions", "targets")})
            if is_main:
                log("Val  loss: {}".format(val["loss"]))
                log("Test loss: {}".format(test["loss"]))
                log("Val  accuracy: {}".format(val["accuracy"]))
                log("Test accuracy: {}".format(test["accuracy"]))
                log("Val  ROCAUC: {}".format(100. * val["rocauc"]))
                log("Test ROCAUC: {}".format(100. * test["rocauc"]))
                log('_' * 40)
                if test["accuracy"] > saving_criteria:
                    stem = os.path.join(save_dir, 'genome_psf_epoch{}'.format(epoch))
                    torch.save(net.state_dict(), '{}_acc{}.pt'.format(stem, test["accuracy"]))
                    torch.save(list(test["predictions"]), stem + '_predictions.pt')
                    torch.save(list(test["targets"]), stem + '_targets.pt')
                    record["checkpoint"] = '{}_acc{}.pt'.format(stem, test["accuracy"])
        history.append(record)
    return history
