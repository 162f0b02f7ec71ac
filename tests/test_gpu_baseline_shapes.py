"""GPU parity at the shapes BASELINE.json names, against results the REFERENCE itself produced there.

The fixtures (tests/golden/ref_*.npz, generator tests/golden/make_golden.py section 8) hold the reference's f32 outputs,
loss and per-tensor gradient norms, plus the reference's OWN drift when it runs under ``torch.autocast(bfloat16)`` in train
mode.  Weights and inputs are re-created from the recorded seeds and verified against recorded checksums.

Stated tolerances (bf16 kernels vs the fp32 reference):
  eval-mode forward      per-timestep rel-L2 <= 1e-2 on the raw output (weak: at random init the output is a DC offset 60-230x
                         the spatial signal) AND, on MEAN-REMOVED frames (error relative to the signal),
                         <= min(1.25 x the reference's own bf16-autocast eval drift on the same case, EVAL_MR_CAP);
                         twice: at random init and with running statistics populated by three reference train-mode forwards
                         (the fixture holds those buffers).  The cap matters where the reference's autocast anchor is
                         useless: its bf16 OUTPUT rounds a 0.3 offset to 1e-3, i.e. away the whole 2e-3 signal (drift 0.31),
                         while this path writes the output layer in f32.
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

from conftest import seeded_case, load_golden_np, checksum, rel_l2, mr_rel_l2_per_t, per_tensor_grad_report, eval_parity, EVAL_MR_CAP

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
    def forward():
        with torch.no_grad():
            outs, _ = model(xd)
        return torch.stack(outs, 1).cpu()

    eval_parity(name, model, g, sd, forward)
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
        # every parameter tensor on its own (the whole-vector figure is dominated by the three ConvLSTM weights and cannot
        # see a wrong outc / BatchNorm gamma, beta / bias gradient): HIP vs f32 oracle against the reference's OWN per-tensor
        # autocast drift, and the norm against the reference's per-tensor norm
        names = [k for k, _ in model.named_parameters()]
        assert names == [str(n) for n in g["grad_names"]]
        bad, rows, med = per_tensor_grad_report(names, {k: p.grad.cpu() for k, p in model.named_parameters()}, rg,
                                                [float(v) for v in g["grad_norms"]], [float(v) for v in g["ac_grad_rel_l2_per_tensor"]])
        print(f"[parity] {name}: per-tensor gradients: median rel-L2 / reference-autocast drift {med:.3f} (bound 1.25); worst five of {len(rows)} (rel-L2 / bound): " +
              "; ".join(f"{k} {rel:.4f}/{bound:.4f}" for _, k, rel, bound, _, z in rows[:5] if not z))
        assert not bad, f"{name}: " + " | ".join(bad[:8])
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


def test_secondary_model_S_at_the_benchmark_batch_vs_reference():
    """SURVEY 8(d) cfg 2 names P *and* S: the reference's class defaults (train/unet.py:132: base_ch 32, use_skip_lstm False) at
    B=32, 64x64 -- eval and train forward, loss, whole and per-tensor gradients against the reference's own f32 run and its own
    autocast drift.  One ConvLSTM only (hidden 512 on 4x4 pixels, M = 512 per timestep: split-K partial tiles)."""
    check_case("ref_cfgS_b32")


def test_autocast_anchor_case_b16():
    check_case("ref_autocast_b16")


def test_moving_mnist_shaped_blobs_vs_reference():
    check_case("ref_blobs64")


def test_config2_cloud_128_vs_reference():
    check_case("ref_cloud128", expect_kernels=[(0, 2)])


def test_config3_256_vs_reference():
    check_case("ref_256")


@pytest.mark.parametrize("case,use_graph", [("ref_512", True), ("ref_512", False), ("ref_512_s950", True)])
def test_config4_512_rollout_vs_reference(case, use_graph):
    """BASELINE configs[4]: 512x512, inference only.  Full-sequence eval forward vs the reference fixture, then the stateful
    frame-by-frame rollout (captured HIP graph) vs both."""
    g, model, xd = check_case(case)
    with torch.no_grad():
        full, _ = model(xd)
    full = torch.stack(full, 1).cpu()
    sp = U.StreamingPredictor(model, use_graph=use_graph, warmup=1)
    got = sp.rollout(xd).cpu()
    e_full, e_ref = per_t(got, full), per_t(got, g["out_eval"])
    m_full, m_ref = mr_rel_l2_per_t(got, full), mr_rel_l2_per_t(got, g["out_eval"])
    anchor = [float(v) for v in g["ac_eval_mr_rel_l2_per_t"]]
    print(f"[parity] {case} rollout (graph={use_graph}): vs full-sequence forward {[round(e, 6) for e in e_full]} (mean-removed "
          f"{[round(e, 5) for e in m_full]}), vs reference {[round(e, 6) for e in e_ref]} (mean-removed {[round(e, 5) for e in m_ref]}, "
          f"reference autocast {[round(e, 5) for e in anchor]})")
    # seed 950's output is all signal (|out| 0.004): its raw figure IS the signal-relative one and sits at the reference's own 1.1e-2
    assert max(e_full) <= 2e-3 and max(e_ref) <= (1.25 * 1.13e-2 if case.endswith("s950") else 1e-2)
    assert max(m_full) <= 2e-2 and all(e <= min(1.25 * r, EVAL_MR_CAP) for e, r in zip(m_ref, anchor))
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


# ---------------------------------------------------------------------------------------------
# configs[2], [3], [4] at their FULL size (base_ch 64 + skip LSTMs): too large for the CPU oracle, so size-independent
# properties, with the kernel plan asserted through ops.LAUNCH_LOG -- the strip-mode patch tiles on the C >= 128 levels of
# images wider than 64 pixels, the 256 x 256 / 8-phase weight gradient, the ring kernels, and the launch splitting at the
# 2-GiB descriptor range run HERE and nowhere else in the suite at real sizes.
# ---------------------------------------------------------------------------------------------
def _full_size_properties(size, T, B, dtype, indep=(0,), roll_b=2, train_tol=5e-6, seed=31):
    """(a) a training step run twice from the same state gives the same loss and, to f32 rounding, the same gradient (no float
    atomics on the GEMM / BatchNorm paths; the bias column sums end in one f32 atomic per block: 1e-6 on those tensors.  This
    check found a write-after-read race in the two ring kernels at strip boundaries in round 3 -- one tensor off by
    1e-4 .. 6e-4 in a third of the processes at 256 x 256, profiles/round3_notes.md); (b) eval-mode batch independence: sample b inside the batch == that sample alone (different kernel plans);
    (c) the stateful frame-by-frame rollout (HIP graph) == the full-sequence forward.  Returns the launch log of (a)."""
    torch.manual_seed(seed)
    with ops.compute_dtype(dtype):
        model = U.TemporalUNetDualView(1, 1, base_ch=64, use_skip_lstm=True).to(DEV)
        data = U.SyntheticSequences(B, T, size, size, seed=seed + 1, kind="uniform")
        opt = U.FusedAdamW(model.parameters(), lr=0.0, weight_decay=0.0, max_grad_norm=None,
                           loss_scale=2.0 ** 14 if dtype == torch.float16 else None)
        model.train()
        runs, log = [], None
        for i in range(2):
            ops.LAUNCH_LOG = [] if i == 0 else None
            loss, _ = U.train_step(model, opt, data.x, data.y, data.mask, True, clip_norm=None)
            if i == 0:
                log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
            runs.append((float(loss), opt.flat.flat_g.detach().clone()))
        d = rel_l2(runs[1][1].cpu(), runs[0][1].cpu())
        worst = sorted(((rel_l2(runs[1][1][o:o + p.numel()].cpu(), runs[0][1][o:o + p.numel()].cpu()), k)
                        for (k, p), o in zip(model.named_parameters(), opt.flat.offsets)), reverse=True)[:4]
        print(f"[parity] {size}x{size} T={T} B={B} {dtype}: step twice: loss {runs[0][0]:.6f} / {runs[1][0]:.6f}, gradient rel-L2 {d:.2e}; "
              f"least reproducible tensors {[(k, float(f'{e:.1e}')) for e, k in worst]}")
        assert runs[0][0] == runs[1][0] and d <= train_tol and bool(torch.isfinite(runs[0][1]).all()) and float(runs[0][1].abs().max()) > 0
        model.eval()
        with torch.no_grad():
            full, _ = model(data.x)
            full = torch.stack(full, 1)
            for b in indep:
                one, _ = model(data.x[b:b + 1].contiguous())
                e = max(per_t(torch.stack(one, 1).cpu(), full[b:b + 1].cpu()))
                m = max(mr_rel_l2_per_t(torch.stack(one, 1).cpu(), full[b:b + 1].cpu()))
                print(f"[parity] {size}x{size} eval: sample {b} alone vs inside the B={B} batch, worst per-timestep rel-L2 {e:.2e} (mean-removed {m:.2e})")
                assert e <= 5e-3 and m <= EVAL_MR_CAP
            sp = U.StreamingPredictor(model, use_graph=True, warmup=1)
            roll = sp.rollout(data.x[:roll_b].contiguous())
            e = max(per_t(roll.cpu(), full[:roll_b].cpu()))
            m = max(mr_rel_l2_per_t(roll.cpu(), full[:roll_b].cpu()))
            print(f"[parity] {size}x{size} eval: {T}-frame stateful rollout (HIP graph) vs full-sequence forward, worst rel-L2 {e:.2e} (mean-removed {m:.2e})")
            assert e <= 5e-3 and m <= EVAL_MR_CAP
    return log


def _fwd(log, epi, shape, min_width=0):
    return [r for r in log if r[0] == "fwd" and r[1] == epi and r[2] == shape and r[3] >= min_width]


def test_config2_full_size_properties():
    """BASELINE configs[2]: cloud 128x128 seq-8, per-GPU batch 32, base_ch 64 + skip LSTMs."""
    log = _full_size_properties(128, 8, 32, torch.bfloat16, indep=(0, 31), roll_b=4)
    assert _fwd(log, 0, 2) and _fwd(log, 0, 3, 128), "patch loop / ring kernel on 128-wide images not exercised"
    assert _fwd(log, 1, 2) or _fwd(log, 2, 2), "ConvLSTM recurrence kernels not exercised"
    assert any(r[0] == "wgrad" and r[1] == 3 for r in log) and any(r[0] == "wgrad" and r[1] == 4 for r in log), "p3 / ring weight gradients"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_config3_full_size_properties(dtype):
    """BASELINE configs[3]: 256x256 seq-12, per-GPU batch 4, 4-level UNet-ConvLSTM at base_ch 64, bf16 and the fp16-MFMA twins.
    The 128x128 level has C = 128: the strip-mode patch tiles (4-row x 64-column blocks of one image) run here."""
    log = _full_size_properties(256, 12, 4, dtype, indep=(0, 3), roll_b=2)
    assert _fwd(log, 0, 2, 128), f"strip-mode patch loop (images wider than 64 pixels, C_out >= 128) not exercised: {sorted(set(log))[:12]}"
    assert any(r[0] == "wgrad" and r[1] == 3 for r in log), "256 x 256 / 8-phase weight gradient not exercised"


def test_config3_reference_batch_is_cut_at_the_descriptor_range():
    """The reference's own 256x256 setting (main.py: B = 32, T = 8): 2^31 bytes in the full-resolution activations, so the
    inc / up0 launches are cut into image ranges on BatchNorm-group boundaries (ops._img_chunks).  Reproducible training step,
    finite gradients, and eval-mode batch independence through the SPLIT launches."""
    torch.manual_seed(41)
    model = U.TemporalUNetDualView(1, 1, base_ch=64, use_skip_lstm=True).to(DEV)
    data = U.SyntheticSequences(32, 8, 256, 256, seed=42, kind="uniform")
    opt = U.FusedAdamW(model.parameters(), lr=0.0, weight_decay=0.0, max_grad_norm=None)
    model.train()
    runs = []
    for i in range(2):
        ops.LAUNCH_LOG = [] if i == 0 else None
        loss, _ = U.train_step(model, opt, data.x, data.y, None, False, clip_norm=None)
        if i == 0:
            log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
        runs.append((float(loss), opt.flat.flat_g.detach().clone()))
    splits = [r for r in log if r[0] == "split"]
    print(f"[parity] 256x256 B=32 T=8: split launches {sorted(set(splits))}; loss {runs[0][0]:.6f} / {runs[1][0]:.6f}")
    assert any(r[1] == "igemm_fwd(store)" for r in splits) and any(r[1] == "igemm_wgrad" for r in splits)
    worst = sorted(((rel_l2(runs[1][1][o:o + p.numel()].cpu(), runs[0][1][o:o + p.numel()].cpu()), k)
                    for (k, p), o in zip(model.named_parameters(), opt.flat.offsets)), reverse=True)[:4]
    print(f"[parity] 256x256 B=32 T=8: least reproducible tensors {[(k, float(f'{e:.1e}')) for e, k in worst]}")
    assert runs[0][0] == runs[1][0] and rel_l2(runs[1][1].cpu(), runs[0][1].cpu()) <= 5e-6 and bool(torch.isfinite(runs[0][1]).all())
    del runs, opt
    model.eval()
    with torch.no_grad():
        full, _ = model(data.x)
        one, _ = model(data.x[31:32].contiguous())
        got, ref = torch.stack(one, 1).cpu(), torch.stack(full, 1)[31:32].cpu()
    e, m = max(per_t(got, ref)), max(mr_rel_l2_per_t(got, ref))
    print(f"[parity] 256x256 B=32 T=8 eval: sample 31 alone vs inside the split batch: rel-L2 {e:.2e} (mean-removed {m:.2e})")
    assert e <= 5e-3 and m <= EVAL_MR_CAP


def test_config4_full_size_properties():
    """BASELINE configs[4]: 512x512 seq-20 autoregressive rollout, inference only, base_ch 64 + skip LSTMs, batch 1: the
    hipGraph-captured recurrent step equals the full-sequence forward, and graph replay equals eager frame-by-frame."""
    torch.manual_seed(51)
    model = U.TemporalUNetDualView(1, 1, base_ch=64, use_skip_lstm=True).to(DEV).eval()
    x = U.SyntheticSequences(1, 20, 512, 512, seed=52, kind="uniform").x
    with torch.no_grad():
        ops.LAUNCH_LOG = []
        full, _ = model(x)
        log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
        full = torch.stack(full, 1).cpu()
        a = U.StreamingPredictor(model, use_graph=True, warmup=1).rollout(x).cpu()
        b = U.StreamingPredictor(model, use_graph=False).rollout(x).cpu()
    assert _fwd(log, 0, 2, 128), "strip-mode patch loop not exercised by the 512x512 forward"
    e, m = max(per_t(a, full)), max(mr_rel_l2_per_t(a, full))
    print(f"[parity] 512x512 T=20 B=1: rollout (HIP graph) vs full-sequence forward rel-L2 {e:.2e} (mean-removed {m:.2e}); graph vs eager "
          f"max |diff| {float((a - b).abs().max()):.1e}")
    assert torch.equal(a, b), "graph replay differs from eager frame-by-frame"
    assert e <= 5e-3 and m <= EVAL_MR_CAP and bool(torch.isfinite(a).all())


def test_grouped_recurrence_matches_separate_launches():
    """The three ConvLSTMs as ONE group launch per timestep (ops.ConvLSTMGroup, uclstm_igemm_fwd_group) against the three
    sequences of launches (ops.ConvLSTMSeq): same arithmetic, only the K-range counts differ (f32 summation order of the split-K
    slabs), at the benchmark's width and batch."""
    torch.manual_seed(77)
    model = U.TemporalUNetDualView(1, 1, base_ch=64, use_skip_lstm=True).to(DEV)
    data = U.SyntheticSequences(32, 3, 64, 64, seed=78, kind="uniform")
    res = {}
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for grouped in (True, False):
        model.load_state_dict(sd0)          # the train-mode forward below moves the running statistics
        ops.GROUP_LSTM = grouped
        try:
            ops.KERNEL_LOG = []
            model.eval()
            with torch.no_grad():
                outs, st = model(data.x)
            ev = torch.stack(outs, 1).cpu()
            model.train()
            model.zero_grad(set_to_none=True)
            outs, _ = model(data.x)
            loss = U.compute_loss(torch.stack(outs, 1), data.y, data.mask, True)
            loss.backward()
            torch.cuda.synchronize()
            log = list(ops.KERNEL_LOG)
        finally:
            ops.KERNEL_LOG = None
            ops.GROUP_LSTM = True
        res[grouped] = (ev, st[0][0].cpu(), st[0][1].cpu(), float(loss), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}, log)
    a, b = res[True], res[False]
    assert (1, 2) in a[5] and (2, 2) in a[5]
    e_out, e_h, e_c = max(per_t(a[0], b[0])), rel_l2(a[1], b[1]), rel_l2(a[2], b[2])
    e_mr = max(mr_rel_l2_per_t(a[0], b[0]))
    worst = sorted(((rel_l2(a[4][k], b[4][k]), k) for k in a[4] if float(b[4][k].abs().max()) > 0), reverse=True)[:4]
    print(f"[parity] grouped vs separate recurrence launches: eval outputs {e_out:.2e} (mean-removed {e_mr:.2e}), final h {e_h:.2e}, c {e_c:.2e}; "
          f"train loss {a[3]:.6f} / {b[3]:.6f}; worst gradients {[(k, float(f'{e:.1e}')) for e, k in worst]}")
    assert e_out <= 1e-3 and e_mr <= 2e-2 and e_h <= 5e-3 and e_c <= 5e-3
    assert abs(a[3] - b[3]) <= 2e-4 * abs(b[3])
    fa = torch.cat([a[4][k].flatten() for k in a[4]])
    fb = torch.cat([b[4][k].flatten() for k in a[4]])
    print(f"[parity] grouped vs separate: whole gradient rel-L2 {rel_l2(fa, fb):.4f}")
    assert rel_l2(fa, fb) <= 5e-2
