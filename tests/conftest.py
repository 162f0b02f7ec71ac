import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load a fixture written by tests/golden/make_golden.py as {key: torch tensor}."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def sub(d, prefix):
    """Entries of ``d`` under ``prefix`` with the prefix stripped."""
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def checksum(tensors):
    s1 = sum(float(t.detach().double().sum()) for t in tensors)
    s2 = sum(float(t.detach().double().abs().sum()) for t in tensors)
    return np.array([s1, s2], dtype=np.float64)


def seeded_case(name):
    """A seed-regenerable reference fixture (tests/golden/make_golden.py section 8): re-create the weights with the build's
    own module (bit-identical seeded initialisation) and the inputs with the build's seeded generators on the CPU, verify
    both against the checksums the REFERENCE run recorded, and return (fixture, state_dict, x, y, mask, cfg)."""
    import unet_convlstm_amd as U
    g = load_golden_np(name)
    base_ch, skip, B, T, HW, seed, use_mask, layers = (int(v) for v in g["cfg"])
    torch.manual_seed(seed)
    m = U.TemporalUNetDualView(1, 1, base_ch=base_ch, lstm_layers=layers, use_skip_lstm=bool(skip), use_attention=False)
    d = U.SyntheticSequences(B, T, HW, HW, seed=seed + 1, kind=str(g["kind"]), device="cpu")
    np.testing.assert_allclose(checksum(m.parameters()), g["param_checksum"], rtol=1e-12, err_msg=f"{name}: seeded init differs")
    np.testing.assert_allclose(checksum([d.x, d.y, d.mask]), g["input_checksum"], rtol=1e-12, err_msg=f"{name}: seeded inputs differ")
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cfg = dict(base_ch=base_ch, skip=bool(skip), B=B, T=T, HW=HW, use_mask=bool(use_mask), lstm_layers=layers)
    gt = {k: (torch.from_numpy(v) if v.dtype.kind in "fiu" else v) for k, v in g.items()}
    return gt, sd, d.x, d.y, d.mask, cfg


def load_golden_np(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: np.asarray(z[k]) for k in z.files}
