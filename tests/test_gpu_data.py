"""GPU tests of the data path around the step: device-side dataset transform, fused
metric sums, and the train_one_epoch / evaluate loops of main.py:77-205 driving the HIP model."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import unet_convlstm_amd as U
from oracle import unet_oracle as O

DEV = "cuda"


def _dataset(tmp_path, g):
    path = tmp_path / "ds.npz"
    np.savez(path, X=g["X"].numpy(), Y=g["Y"].numpy())
    return U.NPZSequenceDataset(str(path))


def test_device_transform_matches_reference_fixture_and_host_dataset(tmp_path):
    g = load_golden("dataset")
    ds = _dataset(tmp_path, g)
    x, y, m = U.device_transform(ds, g["X"].to(DEV), g["Y"].to(DEV))
    # item 1 as produced by the reference's own NPZSequenceDataset.__getitem__
    torch.testing.assert_close(x[1].cpu(), g["x1"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(y[1].cpu(), g["y1"], rtol=1e-5, atol=2e-6)
    assert torch.equal(m[1].cpu(), g["mask1"])
    for i in range(len(ds)):
        xh, yh, mh = ds[i]
        torch.testing.assert_close(x[i].cpu(), xh, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(y[i].cpu(), yh, rtol=1e-5, atol=2e-6)
        assert torch.equal(m[i].cpu(), mh)


def test_metric_sums_match_main_py_formulas(tmp_path):
    g = load_golden("dataset")
    ds = _dataset(tmp_path, g)
    torch.manual_seed(3)
    y = torch.rand(3, 4, 1, 8, 8) * 2 - 1
    yp = y + 0.1 * torch.randn_like(y)
    mask = (torch.rand_like(y) > 0.4).float()
    from unet_convlstm_amd.engine import _Metrics
    for use_mask in (True, False):
        met = _Metrics(DEV)
        met.add(ds, y.to(DEV), yp.to(DEV), mask.to(DEV), use_mask)
        met.add(ds, y.to(DEV), yp.to(DEV), mask.to(DEV), use_mask)          # accumulates over batches
        mae, rmse, me = met.result()
        d = (ds.denormalize(yp.numpy()) - ds.denormalize(y.numpy())).astype(np.float64)   # main.py:115-119
        if use_mask:
            d = d[mask.numpy().astype(bool)]                                              # main.py:123-126
        assert abs(mae - np.abs(d).mean()) <= 1e-5 * max(1.0, np.abs(d).mean())
        assert abs(rmse - math.sqrt((d ** 2).mean())) <= 1e-5 * max(1.0, math.sqrt((d ** 2).mean()))
        assert abs(me - d.mean()) <= 1e-5


def test_train_one_epoch_and_evaluate_loops(tmp_path):
    """main.py:77-205 with the HIP model: an epoch over a tiny .npz, then evaluate(); returns the reference's 4-tuple."""
    rng = np.random.default_rng(0)
    N, T, H, W = 8, 3, 32, 32
    X = (rng.random((N, T, 2, H, W)) * 30).astype(np.float32)
    X[X < 6] = 0.0
    Yv = np.tanh(X[:, :, :1] / 15.0 - 1.0).astype(np.float32) * 4.0
    path = tmp_path / "train.npz"
    np.savez(path, X=X, Y=Yv)
    ds = U.NPZSequenceDataset(str(path))
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)
    torch.manual_seed(0)
    model = U.TemporalUNetDualView(1, 1, base_ch=8, use_skip_lstm=True).to(DEV)
    opt = U.FusedAdamW(model.parameters(), lr=2e-3, weight_decay=1e-4, max_grad_norm=1.0)
    hist = [U.train_one_epoch(model, loader, opt, torch.device(DEV), ds, use_mask=True) for _ in range(6)]
    for out in hist:
        assert len(out) == 4 and all(math.isfinite(v) for v in out)
    assert hist[-1][0] < hist[0][0], [h[0] for h in hist]
    ev = U.evaluate(model, loader, torch.device(DEV), ds, use_mask=True)
    assert len(ev) == 4 and all(math.isfinite(v) for v in ev)
    # evaluate()'s loss equals compute_loss of the oracle run on the same weights (eval mode), within bf16 drift
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    tot, n = 0.0, 0
    with torch.no_grad():
        for x, y, m in loader:
            outs, _ = O.model_forward(sd, x, None, training=False)
            tot += float(O.compute_loss(torch.stack(outs, 1), y, m, True)) * x.shape[0]
            n += x.shape[0]
    assert abs(ev[0] - tot / n) <= 2e-2 * abs(tot / n), (ev[0], tot / n)
