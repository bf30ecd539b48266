"""Zero-edit drop-in: with ``sparsefactorization_amd/shims`` on PYTHONPATH the reference's import lines
(``from torch_sparse import spmm`` — SyntheticExperiments/psf.py:5; ``import torch_geometric`` +
``torch_geometric.data.DataLoader(...)`` — SyntheticExperiments/psf_training.py:8,80-114) resolve to this package."""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = os.path.join(ROOT, "sparsefactorization_amd", "shims")


def _env():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([SHIMS, ROOT] + [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p])
    return env


def test_shim_imports_resolve_and_loader_behaves_like_the_torch_loader():
    code = (
        "import torch, torch_geometric\n"
        "from torch_sparse import spmm\n"
        "from torch.utils.data import Dataset\n"
        "class P(Dataset):\n"
        "    def __len__(self): return 10\n"
        "    def __getitem__(self, i): return torch.full((3,), float(i)), i\n"
        "dl = torch_geometric.data.DataLoader(P(), batch_size=4, shuffle=False, drop_last=True, num_workers=1)\n"
        "b = list(dl)\n"
        "assert len(b) == 2 and b[0][0].shape == (4, 3) and b[1][1].tolist() == [4, 5, 6, 7]\n"
        "assert isinstance(dl, torch.utils.data.DataLoader)\n"
        "import torch_sparse\n"
        "print(spmm.__module__, torch_sparse.__file__, torch_geometric.data.DataLoader.__module__)\n")
    out = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    mod, path, loader_mod = out.stdout.split()
    assert mod == "sparsefactorization_amd.lazy" and path.startswith(SHIMS) and loader_mod == "torch_geometric.data"


def test_install_appends_the_shim_directory_last():
    from sparsefactorization_amd import shims
    before = list(sys.path)
    try:
        d = shims.install()
        assert sys.path[-1] == d == SHIMS and shims.install() == d and sys.path.count(d) == 1
    finally:
        sys.path[:] = before


@pytest.mark.gpu
def test_reference_style_script_trains_through_the_shims(gpu):
    """tests/dropin_reference_style.py: the reference's import lines and loop shape, unedited, on the HIP path."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_reference_style.py")], env=_env(),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["spmm_module"] == "sparsefactorization_amd.lazy" and rec["torch_sparse_file"].startswith(SHIMS)
    assert rec["loader_class"] == "torch_geometric.data.DataLoader"
    assert len(rec["losses"]) == 12 and all(math.isfinite(v) for v in rec["losses"])
    assert rec["losses"][-1] < rec["losses"][0]
    assert rec["chain_rel_err"] <= 1e-6  # the lazy loop IS the chain call: same kernels, same order
