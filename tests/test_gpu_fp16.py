"""The fp16-MFMA twin kernels (BASELINE.json configs[3]: "256x256, 4-level UNet-ConvLSTM, fp16 MFMA").

Same sources compiled for IEEE binary16 (entry points ``*_f16``); selected by ``ops.compute_dtype(torch.float16)``.
Checker: the CPU oracle with ``fp16_storage()`` (rounds to binary16 exactly where the kernels store) and the reference's own
results -- including the reference's OWN drift under ``torch.autocast(float16)`` recorded in the fixtures.

Stated tolerances: eval forward per-timestep rel-L2 <= 2e-3 (binary16 has 11 significand bits: 8x finer than bfloat16's
1e-2); train forward <= 1.25 x the reference's fp16-autocast drift; loss <= 1e-3; gradient vs the f32 oracle within
1.25 x the reference's fp16-autocast gradient drift (static / dynamic loss scaling on both sides)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import seeded_case, rel_l2, eval_parity

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import unet_convlstm_amd as U
    from unet_convlstm_amd import ops
    L = U._lib
from oracle import unet_oracle as O

DEV = "cuda"
torch.set_num_threads(16)


def per_t(got, ref):
    return [rel_l2(got[:, t], ref[:, t]) for t in range(ref.shape[1])]


def cosine(a, b):
    return float(F.cosine_similarity(a.double().flatten(), b.double().flatten(), dim=0))


def test_double_conv_block_fp16_is_bit_exact_against_the_fp16_storage_oracle():
    """(conv3x3 + BN + ReLU) x 2 in train mode on the fp16 kernels: forward bit-identical to the oracle that rounds to
    binary16 where the kernels store, gradients to 1e-2, running statistics updated like the bf16 path."""
    torch.manual_seed(71)
    dc = U.DoubleConv(24, 40).to(DEV).train()
    x = torch.randn(6, 24, 20, 20)
    sd = {"dc." + k: v.detach().cpu().clone() for k, v in dc.state_dict().items()}
    with ops.compute_dtype(torch.float16):
        xd = x.to(DEV).requires_grad_(True)
        yd = dc(xd)
    assert ops.get_compute_dtype() == torch.bfloat16
    (yd * yd).sum().backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if O.is_trainable(k)}
    xr = x.clone().requires_grad_(True)
    buf = {}
    with O.fp16_storage():
        yr = O.double_conv(xr, {**sd, **leaves}, "dc", True, buf)
    (yr * yr).sum().backward()
    e = rel_l2(yd.detach().cpu(), yr.detach())
    print(f"[parity] fp16 DoubleConv forward vs fp16-storage oracle rel-L2 {e:.2e}; dx {rel_l2(xd.grad.cpu(), xr.grad):.4f}")
    assert e <= 1e-3
    assert rel_l2(xd.grad.cpu(), xr.grad) <= 1e-2
    for k, p in dc.named_parameters():
        r = leaves["dc." + k].grad
        if k in ("net.0.bias", "net.3.bias"):          # conv bias in front of BatchNorm: analytically zero, exact zeros here
            assert float(p.grad.abs().max()) == 0.0 and float(r.abs().max()) < 2e-3
            continue
        assert rel_l2(p.grad.cpu(), r) <= 1e-2, k
    torch.testing.assert_close(dc.net[1].running_mean.cpu(), buf["dc.net.1.running_mean"], rtol=2e-3, atol=2e-4)


def test_convlstm_sequence_fp16_matches_oracle():
    torch.manual_seed(72)
    lstm = U.ConvLSTM(24, 40, num_layers=2).to(DEV)
    xs = [torch.randn(3, 24, 12, 12) for _ in range(4)]
    sd = {"l." + k: v.detach().cpu().clone() for k, v in lstm.state_dict().items()}
    with ops.compute_dtype(torch.float16):
        xd = [t.to(DEV).requires_grad_(True) for t in xs]
        outs, st = lstm(xd)
    sum((o * o).sum() for o in outs).backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = [t.clone().requires_grad_(True) for t in xs]
    with O.fp16_storage():
        ro, rst = O.convlstm(xr, leaves, "l", 2)
    sum((o * o).sum() for o in ro).backward()
    e = rel_l2(torch.stack([o.detach().cpu() for o in outs]), torch.stack([o.detach() for o in ro]))
    eg = rel_l2(torch.stack([t.grad.cpu() for t in xd]), torch.stack([t.grad for t in xr]))
    print(f"[parity] fp16 ConvLSTM x2 layers: outputs {e:.2e}, dx {eg:.2e}")
    assert e <= 2e-3 and eg <= 1e-2
    assert rel_l2(st[1][1].detach().cpu(), rst[1][1].detach()) <= 2e-3
    for k, p in lstm.named_parameters():
        assert rel_l2(p.grad.cpu(), leaves["l." + k].grad) <= 1e-2, k


@pytest.mark.parametrize("name", ["ref_256", "ref_autocast_b16"])
def test_model_fp16_vs_reference_and_its_fp16_autocast_drift(name):
    """configs[3] (256x256) and the well-sized anchor case on the fp16 kernels, against the reference's f32 results and the
    reference's own fp16-autocast drift; gradients through dynamic loss scaling (FusedAdamW(loss_scale=...))."""
    g, sd, x, y, mask, cfg = seeded_case(name)
    with ops.compute_dtype(torch.float16):
        model = U.TemporalUNetDualView(1, 1, base_ch=cfg["base_ch"], use_skip_lstm=cfg["skip"]).to(DEV)
        model.load_state_dict(sd)
        model.eval()

        def forward():
            with torch.no_grad():
                outs, _ = model(x.to(DEV))
            return torch.stack(outs, 1).cpu()

        # raw <= 2e-3; mean-removed (relative to the signal) against the reference's own fp16-autocast eval drift
        eval_parity(name + " fp16", model, g, sd, forward, pre="ac16_", raw_tol=2e-3)
        model.train()
        opt = U.FusedAdamW(model.parameters(), lr=0.0, weight_decay=0.0, max_grad_norm=None, loss_scale=2.0 ** 14)
        ops.KERNEL_LOG = []
        loss, y_pred = U.train_step(model, opt, x.to(DEV), y.to(DEV), mask.to(DEV), cfg["use_mask"], clip_norm=None)
        ops.KERNEL_LOG = None
    assert y_pred.dtype == torch.float32
    e_train = per_t(y_pred.cpu(), g["out_train"])
    ref_drift = [float(v) for v in g["ac16_out_rel_l2_per_t"]]
    print(f"[parity] {name} fp16: train forward per-timestep rel-L2 {[round(e, 5) for e in e_train]}; reference fp16-autocast drift "
          f"{[round(e, 5) for e in ref_drift]}")
    for e, r in zip(e_train, ref_drift):
        assert e <= 1.25 * r + 2e-4
    assert abs(float(loss) - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
    scale = 2.0 ** 14
    fg = opt.flat.flat_g.detach().cpu() / scale
    assert bool(torch.isfinite(fg).all())
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if O.is_trainable(k)}
    ro, _ = O.model_forward({**sd, **leaves}, x, None, True, {})
    rl = O.compute_loss(torch.stack(ro, 1), y, mask, cfg["use_mask"])
    names = [k for k, _ in model.named_parameters()]
    fr = torch.cat([t.flatten() for t in torch.autograd.grad(rl, [leaves[k] for k in names])])
    r, c = rel_l2(fg, fr), cosine(fg, fr)
    print(f"[parity] {name} fp16: whole gradient vs f32 oracle rel-L2 {r:.4f} cosine {c:.5f}; reference fp16-autocast drift rel-L2 "
          f"{float(g['ac16_grad_rel_l2']):.4f} cosine {float(g['ac16_grad_cosine']):.5f}; grad norm {float(opt.grad_norm()):.5f} vs "
          f"{float(g['grad_norm']):.5f}")
    assert r <= 1.25 * float(g["ac16_grad_rel_l2"]) + 2e-3 and (1 - c) <= 1.25 * (1 - float(g["ac16_grad_cosine"])) + 1e-5
    assert abs(float(opt.grad_norm()) - float(g["grad_norm"])) <= 2e-2 * float(g["grad_norm"])


def test_loss_scaled_adamw_equals_plain_adamw_and_skips_overflowed_steps():
    torch.manual_seed(73)
    n = 10007
    p0, g0 = torch.randn(n, device=DEV), torch.randn(n, device=DEV) * 1e-3

    def run(scaled, g, steps=3):
        p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        sq = torch.zeros(1, dtype=torch.float64, device=DEV)
        state = torch.tensor([1024.0, 0.0, 0.0], device=DEV)
        for s in range(1, steps + 1):
            sq.zero_()
            gg = g * state[0] if scaled else g
            L.check(L.lib.uclstm_sumsq(gg.data_ptr(), n, sq.data_ptr(), None), "sumsq")
            if scaled:
                L.check(L.lib.uclstm_adamw_step_scaled(p.data_ptr(), m.data_ptr(), v.data_ptr(), gg.data_ptr(), n, sq.data_ptr(), 0.01, 1e-2, 0.9,
                                                       0.999, 1e-8, 1e-4, state.data_ptr(), None), "adamw_scaled")
                L.check(L.lib.uclstm_loss_scale_update(state.data_ptr(), sq.data_ptr(), 2.0, 0.5, 2, None), "scale_update")
            else:
                L.check(L.lib.uclstm_adamw_step(p.data_ptr(), m.data_ptr(), v.data_ptr(), gg.data_ptr(), n, sq.data_ptr(), 0.01, 1e-2, 0.9, 0.999,
                                                1e-8, 1e-4, s, None), "adamw")
        return p, state
    pa, _ = run(False, g0)
    pb, st = run(True, g0)
    torch.testing.assert_close(pb, pa, rtol=2e-5, atol=2e-6)
    assert st.tolist() == [2048.0, 1.0, 3.0]              # grew once after two good steps in a row, three successful steps
    bad = g0.clone()
    bad[5] = float("inf")
    pc, st = run(True, bad, steps=2)
    assert torch.equal(pc, p0) and st.tolist() == [256.0, 0.0, 0.0]      # both steps skipped, scale halved twice


def test_fp16_training_steps_with_dynamic_loss_scaling_learn():
    """A too-large initial scale overflows binary16 in backward: those steps are skipped on the device (no host sync), the
    scale backs off, then training proceeds and the loss falls."""
    torch.manual_seed(74)
    with ops.compute_dtype(torch.float16):
        model = U.TemporalUNetDualView(1, 1, base_ch=8, use_skip_lstm=True).to(DEV).train()
        opt = U.FusedAdamW(model.parameters(), lr=2e-3, weight_decay=1e-4, max_grad_norm=1.0, loss_scale=2.0 ** 24, scale_interval=4)
        data = U.SyntheticSequences(4, 4, 64, 64, seed=3, kind="blobs")
        p_before = opt.flat.flat_p.clone()
        losses, scales = [], []
        for _ in range(14):
            loss, _ = U.train_step(model, opt, data.x, data.y, data.mask, True)
            losses.append(float(loss))
            scales.append(float(opt.scale_state[0]))
    print("[parity] fp16 dynamic loss scaling: scales", scales, "losses", [round(v, 4) for v in losses])
    assert scales[0] < 2.0 ** 24                      # the first step overflowed and was skipped
    assert float(opt.scale_state[2]) >= 6             # successful steps were taken afterwards
    assert all(v == v for v in losses) and losses[-1] < losses[0]
    assert not torch.equal(opt.flat.flat_p, p_before) and bool(torch.isfinite(opt.flat.flat_p).all())


@pytest.mark.parametrize("N,C0,C1,Co,H,W", [(2, 64, 0, 64, 8, 128), (2, 64, 64, 64, 4, 64), (2, 64, 0, 256, 16, 16), (2, 64, 0, 128, 32, 32)])
def test_weight_gradient_kernels_fp16(N, C0, C1, Co, H, W):
    """The fp16 twins of the weight-gradient kernels (ring-staged 64-channel kernel, 8-phase 256x256, 128x128) against
    F.conv2d's weight gradient on the same binary16-rounded operands."""
    import ctypes
    torch.manual_seed(81)
    h = lambda t: t.half().float()
    x0, dy = h(torch.randn(N, C0, H, W)), h(torch.randn(N, Co, H, W))
    with ops.compute_dtype(torch.float16):
        srcs = [ops.SrcView(ops.ToNHWC.apply(x0.to(DEV)))]
        xin, cv = x0, [C0]
        if C1:
            x1 = h(torch.randn(N, C1, H, W))
            srcs.append(ops.SrcView(ops.ToNHWC.apply(x1.to(DEV))))
            xin, cv = torch.cat((x0, x1), 1), [C0, C1]
        dyn = ops.ToNHWC.apply(dy.to(DEV))
    assert dyn.dtype == torch.float16
    pd = ops.conv_pack_desc(Co, C0 + C1, cv, cv)
    dwp = ops.igemm_wgrad(srcs, [(dyn, 0, Co, 0, 1, 0, 0)], pd.N, pd.Ktot, (H, W), N, ktap=3, pad=1)
    got = ops.unpack_wgrad(pd, dwp, torch.zeros(Co, C0 + C1, 3, 3, device=DEV)).cpu()
    wr = torch.zeros(Co, C0 + C1, 3, 3, requires_grad=True)
    (F.conv2d(xin, wr, None, padding=1) * dy).sum().backward()
    e = rel_l2(got, wr.grad)
    print(f"[parity] fp16 weight gradient N={N} C={C0}+{C1} -> {Co} {H}x{W}: rel-L2 {e:.2e}")
    assert e <= 2e-6
