"""Adding / Temporal-Order data, generated on the device in one shot.

Same distributions as the reference generator (SyntheticExperiments/synth_data_generation.py:8-70), which
builds every sequence in a Python loop with ``random.sample`` (200 000 iterations per split) and saves ``.pt``
files; here a split is a handful of vectorised tensor ops on the GPU, seeded by a ``torch.Generator``.
Parity is distributional, not bitwise (the reference draws from Python's ``random``).

* Adding (``adding``, lines 8-28): ``x ~ U(-1, 1)`` per position, a 0/1 marker channel with exactly two
  distinct marked positions, stacked to ``[n, N, 2]`` float32; label ``0.5 + (x[p1] + x[p2]) / 4``.
* Temporal order (``temporal_order``, lines 30-70): tokens uniform in {0,1,2,3}; two distinct positions,
  in increasing order, overwritten with independent draws from {4, 5} (X, Y); label in {0..3} =
  ``2*[first == Y] + [second == Y]``; data ``[n, N, 1]`` int64.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def _two_distinct_positions(sequences: int, n_data: int, device, generator) -> Tuple[torch.Tensor, torch.Tensor]:
    """Uniform unordered pair of distinct positions per sequence, returned sorted (pos_1 < pos_2)."""
    if n_data < 2:
        raise ValueError("need at least two positions per sequence")
    a = torch.randint(0, n_data, (sequences,), device=device, generator=generator)
    b = torch.randint(0, n_data - 1, (sequences,), device=device, generator=generator)
    b = b + (b >= a).to(b.dtype)  # uniform over the n_data-1 positions != a
    return torch.minimum(a, b), torch.maximum(a, b)


def adding(sequences: int, n_data: int, device="cpu", generator: Optional[torch.Generator] = None):
    """Returns (data [sequences, n_data, 2] float32, labels [sequences] float32)."""
    x = torch.rand(sequences, n_data, device=device, generator=generator) * 2 - 1
    p1, p2 = _two_distinct_positions(sequences, n_data, device, generator)
    y = torch.zeros(sequences, n_data, device=device)
    rows = torch.arange(sequences, device=device)
    y[rows, p1] = 1.0
    y[rows, p2] = 1.0
    labels = 0.5 + (x[rows, p1] + x[rows, p2]) / 4
    return torch.stack([x, y], dim=-1), labels


def temporal_order(sequences: int, n_data: int, device="cpu", generator: Optional[torch.Generator] = None):
    """Returns (data [sequences, n_data, 1] int64, labels [sequences] int64)."""
    x = torch.randint(0, 4, (sequences, n_data), device=device, generator=generator)
    p1, p2 = _two_distinct_positions(sequences, n_data, device, generator)
    v1 = torch.randint(4, 6, (sequences,), device=device, generator=generator)
    v2 = torch.randint(4, 6, (sequences,), device=device, generator=generator)
    rows = torch.arange(sequences, device=device)
    x[rows, p1] = v1
    x[rows, p2] = v2
    labels = 2 * (v1 == 5).long() + (v2 == 5).long()
    return x.unsqueeze(-1), labels
