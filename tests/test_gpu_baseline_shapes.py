"""GPU parity at the shapes BASELINE.json names, against results the REFERENCE itself produced there.

The fixtures (tests/golden/ref_*.npz, generator tests/golden/make_golden.py section 8) hold the reference's f32 outputs,
loss and per-tensor gradient norms, plus the reference's OWN drift when it runs under ``torch.autocast(bfloat16)`` in train
mode.  Weights and inputs are re-created from the recorded seeds and verified against recorded checksums.

Stated tolerances (bf16 kernels vs the fp32 reference):
  eval-mode forward      per-timestep rel-L2 <= 1e-2
  train-mode forward     per-timestep rel-L2 <= 1.25 x the reference's own bf16-autocast drift on the same case
                         (2.2e-2 at base_ch 64 / B 32, 2.6e-2 at base_ch 8 / B 16: batch-statistics BatchNorm at random
                         init re-amplifies every bf16 rounding; the reference does it too)
  loss <= 5e-3 relative, gradient norm <= 5 %
  whole gradient vs the f32 oracle (pinned to the reference at these very shapes by tests/test_oracle_golden.py):
                         rel-L2 <= 1.25 x the reference's autocast gradient drift, cosine >= 1 - 1.25 (1 - its cosine)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import seeded_case, load_golden_np, checksum, rel_l2

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import unet_convlstm_amd as U
    from unet_convlstm_amd import ops
from oracle import unet_oracle as O

DEV = "cuda"
torch.set_num_threads(16)


def cosine(a, b):
    return float(F.cosine_similarity(a.double().flatten(), b.double().flatten(), dim=0))


def per_t(got, ref):
    return [rel_l2(got[:, t], ref[:, t]) for t in range(ref.shape[1])]


def build(sd, cfg):
    m = U.TemporalUNetDualView(1, 1, base_ch=cfg["base_ch"], lstm_layers=cfg["lstm_layers"], use_skip_lstm=cfg["skip"]).to(DEV)
    m.load_state_dict(sd)
    return m


def oracle_grads(sd, x, y, mask, use_mask):
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if O.is_trainable(k)}
    outs, _ = O.model_forward({**sd, **leaves}, x, None, True, {})
    loss = O.compute_loss(torch.stack(outs, 1), y, mask, use_mask)
    names = list(leaves)
    return dict(zip(names, torch.autograd.grad(loss, [leaves[k] for k in names])))


def check_case(name, expect_kernels=(), check_grads=True):
    g, sd, x, y, mask, cfg = seeded_case(name)
    model = build(sd, cfg).eval()
    xd, yd, md = x.to(DEV), y.to(DEV), mask.to(DEV)
    with torch.no_grad():
        outs, _ = model(xd)
    e_eval = per_t(torch.stack(outs, 1).cpu(), g["out_eval"])
    print(f"[parity] {name}: eval forward vs reference per-timestep rel-L2 {[round(e, 6) for e in e_eval]} (tol 1e-2)")
    assert max(e_eval) <= 1e-2
    if "out_train" not in g:
        return g, model, xd
    model.train()
    ops.KERNEL_LOG = []
    try:
        outs, _ = model(xd)
        y_pred = torch.stack(outs, 1)
        loss = U.compute_loss(y_pred, yd, md, cfg["use_mask"])
        loss.backward()
        log = set(ops.KERNEL_LOG)
    finally:
        ops.KERNEL_LOG = None
    for k in expect_kernels:
        assert k in log, f"{name}: kernel (epilogue, shape) {k} was not exercised; saw {sorted(log)}"
    e_train = per_t(y_pred.detach().cpu(), g["out_train"])
    ref_drift = [float(v) for v in g["ac_out_rel_l2_per_t"]]
    print(f"[parity] {name}: train forward vs reference per-timestep rel-L2 {[round(e, 5) for e in e_train]}; the reference's own "
          f"bf16-autocast drift {[round(e, 5) for e in ref_drift]}")
    for t, (e, r) in enumerate(zip(e_train, ref_drift)):
        assert e <= 1.25 * r, f"{name} t={t}: train forward drift {e:.4f} > 1.25 x reference autocast drift {r:.4f}"
    dl = abs(float(loss) - float(g["loss"])) / abs(float(g["loss"]))
    print(f"[parity] {name}: loss {float(loss):.6f} vs reference {float(g['loss']):.6f} (rel {dl:.2e}; reference autocast "
          f"{abs(float(g['ac_loss']) - float(g['loss'])) / abs(float(g['loss'])):.2e})")
    assert dl <= 5e-3
    fg = torch.cat([p.grad.flatten().cpu() for _, p in model.named_parameters()])
    gn = float(fg.double().norm())
    print(f"[parity] {name}: gradient norm {gn:.5f} vs reference {float(g['grad_norm']):.5f} (reference autocast {float(g['ac_grad_norm']):.5f})")
    assert abs(gn - float(g["grad_norm"])) <= 5e-2 * float(g["grad_norm"])
    if check_grads:
        rg = oracle_grads(sd, x, y, mask, cfg["use_mask"])
        fr = torch.cat([rg[k].flatten() for k, _ in model.named_parameters()])
        r, c = rel_l2(fg, fr), cosine(fg, fr)
        r_ref, c_ref = float(g["ac_grad_rel_l2"]), float(g["ac_grad_cosine"])
        print(f"[parity] {name}: whole gradient vs f32 oracle rel-L2 {r:.4f} cosine {c:.5f}; the reference's own autocast "
              f"gradient drift rel-L2 {r_ref:.4f} cosine {c_ref:.5f}")
        assert r <= 1.25 * r_ref and (1 - c) <= 1.25 * (1 - c_ref)
    return g, model, xd


def test_config1_benchmark_kernel_plan_vs_reference():
    """BASELINE configs[1] at the bench's own batch: base_ch 64 + skip LSTMs, B=32, 64x64 (T=2 keeps the oracle's
    gradient under 30 s).  At B=32 the library picks the fused ConvLSTM cell for skip2, split-K partial tiles for skip3 and
    the bottleneck LSTM, the patch loop for the 3x3 family and the ring kernel for the 64->64 layers: the plan the driver
    times -- asserted through ops.KERNEL_LOG."""
    check_case("ref_cfg1_b32", expect_kernels=[(1, 2), (2, 2), (0, 2), (0, 3), (0, 1)])


def test_config1_train_forward_also_matches_bf16_storage_oracle():
    """Level A at the benchmark plan: same rounding points as the kernels, so the comparison is tighter than the drift."""
    g, sd, x, y, mask, cfg = seeded_case("ref_cfg1_b32")
    model = build(sd, cfg).train()
    with torch.no_grad():
        outs, _ = model(x.to(DEV))
        with O.bf16_storage():
            emu, _ = O.model_forward(sd, x, None, True, {})
    e = per_t(torch.stack(outs, 1).cpu(), torch.stack(emu, 1))
    print(f"[parity] ref_cfg1_b32: train forward vs bf16-storage oracle per-timestep rel-L2 {[round(v, 5) for v in e]} (tol 1.5e-2)")
    assert max(e) <= 1.5e-2


def test_autocast_anchor_case_b16():
    check_case("ref_autocast_b16")


def test_moving_mnist_shaped_blobs_vs_reference():
    check_case("ref_blobs64")


def test_config2_cloud_128_vs_reference():
    check_case("ref_cloud128", expect_kernels=[(0, 2)])


def test_config3_256_vs_reference():
    check_case("ref_256")


@pytest.mark.parametrize("use_graph", [True, False])
def test_config4_512_rollout_vs_reference(use_graph):
    """BASELINE configs[4]: 512x512, inference only.  Full-sequence eval forward vs the reference fixture, then the stateful
    frame-by-frame rollout (captured HIP graph) vs both."""
    g, model, xd = check_case("ref_512")
    with torch.no_grad():
        full, _ = model(xd)
    full = torch.stack(full, 1).cpu()
    sp = U.StreamingPredictor(model, use_graph=use_graph, warmup=1)
    got = sp.rollout(xd).cpu()
    e_full, e_ref = per_t(got, full), per_t(got, g["out_eval"])
    print(f"[parity] ref_512 rollout (graph={use_graph}): vs full-sequence forward {[round(e, 6) for e in e_full]}, vs reference "
          f"{[round(e, 6) for e in e_ref]}")
    assert max(e_full) <= 2e-3 and max(e_ref) <= 1e-2
    if use_graph:
        assert all(g is not None for g in sp._graphs)


def test_config0_convlstm_plus_head_vs_reference():
    """BASELINE configs[0] plumbing on the HIP path: ConvLSTM(2,16,1) + OutConv(16,1) over 20 Moving-MNIST-shaped frames,
    list-in / list-out as a reference-style script drives it; outputs, final state, loss and every gradient against the
    reference's own run."""
    from test_oracle_golden import cfg0_model
    g = {k: torch.from_numpy(v) for k, v in load_golden_np("ref_cfg0_convlstm_head").items()}
    lstm, head, d = cfg0_model()
    np.testing.assert_allclose(checksum(list(lstm.parameters()) + list(head.parameters())), g["param_checksum"].numpy(), rtol=1e-12)
    lstm, head = lstm.to(DEV), head.to(DEV)
    x = d.x.to(DEV)
    outs, st = lstm([x[:, t] for t in range(x.shape[1])])
    y_pred = torch.stack([head(o) for o in outs], dim=1)
    loss = U.compute_loss(y_pred, d.y.to(DEV), d.mask.to(DEV), True)
    loss.backward()
    e = per_t(y_pred.detach().cpu(), g["out"])
    print(f"[parity] cfg0: outputs per-timestep rel-L2 max {max(e):.5f}, loss {float(loss):.6f} vs {float(g['loss']):.6f}")
    assert max(e) <= 1e-2
    assert rel_l2(st[0][0].cpu(), g["h_final"]) <= 1e-2 and rel_l2(st[0][1].cpu(), g["c_final"]) <= 1e-2
    assert abs(float(loss) - float(g["loss"])) <= 5e-3 * abs(float(g["loss"]))
    worst = {}
    for mod, pre in ((lstm, "lstm."), (head, "head.")):
        for k, p in mod.named_parameters():
            worst[pre + k] = rel_l2(p.grad.cpu(), g["g/" + pre + k])
    print("[parity] cfg0: gradient rel-L2 " + ", ".join(f"{k}={v:.4f}" for k, v in worst.items()))
    assert max(worst.values()) <= 3e-2


@pytest.mark.parametrize("ch,hw,seed", [(64, 16, 800), (64, 8, 801), (128, 8, 802), (256, 4, 803), (512, 2, 804)])
def test_resnet18_convlstms_vs_reference(ch, hw, seed):
    """ConvLSTM(ch, ch, num_layers=2) at the five ResNet18 widths, driven exactly as
    train/resnet18.py:101-104,:120-128 does (per-timestep views of a [B*T]-flattened feature tensor, list in, list out,
    stack + view back), against the reference's own outputs and gradients."""
    from test_oracle_golden import resnet_lstm_case
    g = {k: torch.from_numpy(v) for k, v in load_golden_np("ref_resnet_lstms").items()}
    lstm, feat = resnet_lstm_case(ch, hw, seed)
    tag = f"{ch}_{hw}"
    np.testing.assert_allclose(checksum(lstm.parameters()), g[f"{tag}/param_checksum"].numpy(), rtol=1e-12)
    B, T = 2, 3
    lstm = lstm.to(DEV)
    feat = feat.to(DEV).requires_grad_(True)
    feat_seq = feat.view(B, T, ch, hw, hw)
    lstm_in = [feat_seq[:, t] for t in range(T)]
    lstm_out_list, _ = lstm(lstm_in)
    out = torch.stack(lstm_out_list, dim=1).view(B * T, ch, hw, hw)
    (out * out).sum().backward()
    e_out, e_g = rel_l2(out.detach().cpu(), g[f"{tag}/out"]), rel_l2(feat.grad.cpu(), g[f"{tag}/gfeat"])
    norms = torch.tensor([float(p.grad.double().norm()) for p in lstm.parameters()], dtype=torch.float64)
    e_n = float(((norms - g[f"{tag}/grad_norms"]).abs() / g[f"{tag}/grad_norms"]).max())
    e_b = rel_l2(lstm.layers[0].conv.bias.grad.cpu(), g[f"{tag}/gbias0"])
    e_w = rel_l2(lstm.layers[1].conv.weight.grad[:8, :8].cpu(), g[f"{tag}/gw1_slice"])
    print(f"[parity] resnet18 ConvLSTM ch={ch} {hw}x{hw}: out {e_out:.5f}, d/dfeatures {e_g:.5f}, gradient norms {e_n:.5f}, "
          f"layer-0 bias gradient {e_b:.5f}, layer-1 weight-gradient slice {e_w:.5f}")
    assert e_out <= 1e-2 and e_g <= 3e-2 and e_n <= 3e-2 and e_b <= 3e-2 and e_w <= 3e-2


def test_config1_full_size_properties():
    """BASELINE configs[1] at its FULL size (base_ch 64 + skip LSTMs, B=32, T=20, 64x64 -- too large for the CPU oracle in a
    test), through size-independent properties: (a) a training step is reproducible (same loss, gradients equal to f32
    rounding: no float atomics anywhere on the path); (b) eval-mode batch independence: sample b of the B=32 forward equals the
    B=1 forward of that sample (different kernel plans: fused cell / split-K at B=32, per-tap shapes at B=1); (c) the stateful
    frame-by-frame rollout over all 20 frames equals the full-sequence forward."""
    torch.manual_seed(91)
    model = U.TemporalUNetDualView(1, 1, base_ch=64, use_skip_lstm=True).to(DEV)
    data = U.SyntheticSequences(32, 20, 64, 64, seed=7, kind="blobs")
    opt = U.FusedAdamW(model.parameters(), lr=0.0, weight_decay=0.0, max_grad_norm=None)
    model.train()
    runs = []
    for _ in range(2):
        loss, _ = U.train_step(model, opt, data.x, data.y, data.mask, True, clip_norm=None)
        runs.append((float(loss), opt.flat.flat_g.detach().clone()))
    d = rel_l2(runs[1][1].cpu(), runs[0][1].cpu())
    print(f"[parity] full-size step twice: loss {runs[0][0]:.6f} / {runs[1][0]:.6f}, gradient rel-L2 {d:.2e}")
    assert runs[0][0] == runs[1][0] and d <= 1e-6 and bool(torch.isfinite(runs[0][1]).all()) and float(runs[0][1].abs().max()) > 0
    model.eval()
    with torch.no_grad():
        full, _ = model(data.x)
        full = torch.stack(full, 1)
        for b in (0, 17, 31):
            one, _ = model(data.x[b:b + 1].contiguous())
            e = max(per_t(torch.stack(one, 1).cpu(), full[b:b + 1].cpu()))
            print(f"[parity] full-size eval: sample {b} alone vs inside the B=32 batch, worst per-timestep rel-L2 {e:.2e}")
            assert e <= 5e-3        # measured 2.6e-3: f32 summation order differs between the plans, h is re-rounded to bf16 20 times
        sp = U.StreamingPredictor(model, use_graph=True, warmup=1)
        roll = sp.rollout(data.x[:4].contiguous())
        e = max(per_t(roll.cpu(), full[:4].cpu()))
        print(f"[parity] full-size eval: 20-frame stateful rollout (HIP graph) vs full-sequence forward, worst rel-L2 {e:.2e}")
        assert e <= 5e-3            # measured 2.4e-3 (batch 4 frame by frame vs batch 32 in one pass: different kernel plans, as above)
