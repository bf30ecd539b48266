"""A training script in the STYLE of the reference's (written for this repo, not a copy of any reference file): the
reference's import lines, its loop shape (one ``spmm`` + one residual add per factor, SyntheticExperiments/psf.py:172-188)
and its loader / optimizer calls (SyntheticExperiments/psf_training.py:50-58,80-86; psf_utils.py:65-73) — run with
``sparsefactorization_amd/shims`` on PYTHONPATH and nothing else changed. Prints one JSON line.

    PYTHONPATH=sparsefactorization_amd/shims:. python tests/dropin_reference_style.py
"""
import json
import sys

import torch
from torch import nn, optim
import torch_geometric
from torch_sparse import spmm
from torch.utils.data import Dataset


def chord_index_lists(n_vec, n_link):
    """Rows / columns of the chord pattern: node i links to itself and to i + 2^k (mod n_vec), k < n_link - 1."""
    rows, cols = [], []
    for i in range(n_vec):
        for k in range(n_link):
            rows.append(i)
            cols.append(i if k == 0 else (i + (1 << (k - 1))) % n_vec)
    return [rows, cols]


class Pairs(Dataset):
    def __init__(self, data, labels):
        self.data, self.labels = data, labels

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.data[idx], self.labels[idx]


class TinyChordNet(nn.Module):
    def __init__(self, n_vec, width, channels, n_W):
        super().__init__()
        self.n_vec, self.n_W, self.n_links = n_vec, n_W, n_W + 1
        self.lift = nn.Linear(2, width)
        mlp = lambda out: nn.Sequential(nn.Linear(width, width), nn.GELU(), nn.Linear(width, out))  # noqa: E731
        self.fs = nn.ModuleList([mlp(self.n_links) for _ in range(n_W)])
        self.g = mlp(channels)
        self.final = nn.Linear(n_vec * channels, 1)
        self.chord_indicies = torch.tensor(chord_index_lists(n_vec, self.n_links)).cuda()

    def forward(self, data):
        data = self.lift(data)
        V = self.g(data)
        res_conn = V
        for m in range(self.n_W):
            W = self.fs[m](data)
            V = spmm(self.chord_indicies, W.reshape(W.size(0), W.size(1) * W.size(2)), self.n_vec, self.n_vec, V)
            V = V + res_conn
        return self.final(V.view(V.size(0), -1))


def main():
    torch.manual_seed(7)
    n_vec, n_W, B = 256, 8, 16
    x = torch.rand(64, n_vec, 2) * 2 - 1
    y = x[..., 0].mean(dim=1)
    loader = torch_geometric.data.DataLoader(Pairs(x, y), batch_size=B, shuffle=True, drop_last=True, num_workers=0)
    net = TinyChordNet(n_vec, 32, 8, n_W).cuda()
    optimizer = optim.Adam(net.parameters(), lr=1e-3)
    loss = nn.MSELoss()
    losses = []
    for _ in range(3):
        for X, Y in loader:
            X, Y = X.cuda(), Y.cuda()
            optimizer.zero_grad()
            out = loss(net(X).squeeze(), Y)
            out.backward()
            optimizer.step()
            losses.append(float(out.item()))
    # the hot loop against the package's explicit chain call on the same operands
    from sparsefactorization_amd import chord_chain
    with torch.no_grad():
        X = x[:B].cuda()
        data = net.lift(X)
        V0 = net.g(data)
        Ws = [f(data) for f in net.fs]
        V = V0
        for W in Ws:
            V = spmm(net.chord_indicies, W.reshape(W.size(0), W.size(1) * W.size(2)), n_vec, n_vec, V)
            V = V + V0
        ref = chord_chain(Ws, V0, True)
        err = float((V - ref).abs().max() / ref.abs().max())
    print(json.dumps({"losses": losses, "chain_rel_err": err, "spmm_module": spmm.__module__,
                      "loader_class": type(loader).__module__ + "." + type(loader).__name__,
                      "torch_sparse_file": sys.modules["torch_sparse"].__file__}))


if __name__ == "__main__":
    main()
