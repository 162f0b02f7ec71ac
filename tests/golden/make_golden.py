#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Run in the build container only (the reference checkout does not travel):

    python tests/golden/make_golden.py [/root/reference]

It imports the reference's ``train/unet.py`` and ``main.py`` (CPU, fp32,
torch seeds stated per fixture), runs them on small seeded inputs and stores
inputs, weights and expected outputs/gradients as ``.npz`` files.  ``main.py``
imports ``train/resnet18.py`` which needs ``segmentation_models_pytorch``
(not installed): an EMPTY stub module is registered for that name so that
``compute_loss`` / the step order become importable; ``PretrainedTemporalUNet``
is never instantiated.

Fixtures are data only: no reference source text is stored.
"""
import os
import sys
import types

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.modules.setdefault("segmentation_models_pytorch", types.ModuleType("segmentation_models_pytorch"))

from train.unet import (ConvLSTMCell, ConvLSTM, DoubleConv, Down, Up, OutConv,  # noqa: E402
                        SpatialAttention, TemporalUNetDualView, NPZSequenceDataset)
import main as ref_main  # noqa: E402

torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().clone().numpy()   # clone: later in-place updates must not alias saved arrays


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(arrs)} arrays)")


def sd_arrays(mod, prefix="p/"):
    return {prefix + k: npy(v) for k, v in mod.state_dict().items()}


# ---------------------------------------------------------------- 1. cell
def gen_cell():
    out = {}
    for tag, (cin, hd, B, H, W, seed) in {
        "a": (4, 8, 2, 16, 16, 100),
        "b": (4, 5, 2, 7, 9, 101),        # hidden_dim not a multiple of anything, non-square
    }.items():
        torch.manual_seed(seed)
        cell = ConvLSTMCell(cin, hd)
        x = torch.randn(B, cin, H, W, requires_grad=True)
        h0 = (0.5 * torch.randn(B, hd, H, W)).requires_grad_(True)
        c0 = (0.5 * torch.randn(B, hd, H, W)).requires_grad_(True)
        h1, (_, c1) = cell(x, (h0, c0))
        (h1.sum() + c1.sum()).backward()
        out.update({f"{tag}/weight": npy(cell.conv.weight), f"{tag}/bias": npy(cell.conv.bias),
                    f"{tag}/x": npy(x), f"{tag}/h0": npy(h0), f"{tag}/c0": npy(c0),
                    f"{tag}/h1": npy(h1), f"{tag}/c1": npy(c1),
                    f"{tag}/gx": npy(x.grad), f"{tag}/gh0": npy(h0.grad), f"{tag}/gc0": npy(c0.grad),
                    f"{tag}/gw": npy(cell.conv.weight.grad), f"{tag}/gb": npy(cell.conv.bias.grad)})
        # state=None case
        hn, (_, cn) = cell(x.detach())
        out.update({f"{tag}/h1_none": npy(hn), f"{tag}/c1_none": npy(cn)})
    save("cell", **out)


# ---------------------------------------------------------------- 2. sequence
def gen_seq():
    torch.manual_seed(200)
    lstm = ConvLSTM(4, 8, num_layers=2)
    T, B, H, W = 5, 2, 12, 12
    xs = [torch.randn(B, 4, H, W, requires_grad=True) for _ in range(T)]
    outs, states = lstm(xs)
    loss = sum((o * o).sum() for o in outs) * 0.5 + states[0][1].sum() + states[1][0].sum()
    loss.backward()
    arr = sd_arrays(lstm)
    arr.update({f"g/{k}": npy(v.grad) for k, v in lstm.named_parameters()})
    arr["x"] = np.stack([npy(x) for x in xs])
    arr["gx"] = np.stack([npy(x.grad) for x in xs])
    arr["out"] = np.stack([npy(o) for o in outs])
    for li, (h, c) in enumerate(states):
        arr[f"h_final/{li}"] = npy(h)
        arr[f"c_final/{li}"] = npy(c)
    # continuation with carried state
    with torch.no_grad():
        outs2, states2 = lstm([x.detach() for x in xs[:2]], [(h.detach(), c.detach()) for h, c in states])
    arr["out_cont"] = np.stack([npy(o) for o in outs2])
    save("seq", **arr)


# ---------------------------------------------------------------- 3. blocks
def gen_blocks():
    arr = {}
    torch.manual_seed(300)
    dc = DoubleConv(3, 8)
    with torch.no_grad():
        for bn in (dc.net[1], dc.net[4]):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    arr.update(sd_arrays(dc, "dc/p/"))
    xa = torch.randn(4, 3, 10, 10, requires_grad=True)
    xb = torch.randn(4, 3, 10, 10)
    dc.train()
    ya = dc(xa)
    (ya * torch.linspace(0.5, 1.5, ya.numel()).view_as(ya)).sum().backward()
    arr["dc/xa"], arr["dc/ya_train"], arr["dc/gxa"] = npy(xa), npy(ya), npy(xa.grad)
    arr.update({f"dc/g/{k}": npy(v.grad) for k, v in dc.named_parameters()})
    with torch.no_grad():
        yb = dc(xb)
    arr["dc/xb"], arr["dc/yb_train"] = npy(xb), npy(yb)
    arr.update(sd_arrays(dc, "dc/p_after2/"))
    dc.eval()
    with torch.no_grad():
        arr["dc/ya_eval"] = npy(dc(xa.detach()))

    torch.manual_seed(301)
    upm = Up(16, 8)
    arr.update(sd_arrays(upm, "up/p/"))
    x1 = torch.randn(2, 16, 5, 6, requires_grad=True)
    x2 = torch.randn(2, 8, 11, 13, requires_grad=True)      # odd skip size: F.pad path
    upm.train()
    yu = upm(x1, x2)
    (yu * yu).sum().backward()
    arr.update({"up/x1": npy(x1), "up/x2": npy(x2), "up/y_train": npy(yu),
                "up/gx1": npy(x1.grad), "up/gx2": npy(x2.grad)})
    arr.update({f"up/g/{k}": npy(v.grad) for k, v in upm.named_parameters()})

    torch.manual_seed(302)
    dn = Down(8, 16)
    arr.update(sd_arrays(dn, "down/p/"))
    xd = torch.randn(2, 8, 12, 12, requires_grad=True)
    dn.train()
    yd = dn(xd)
    (yd * yd).sum().backward()
    arr.update({"down/x": npy(xd), "down/y_train": npy(yd), "down/gx": npy(xd.grad)})
    arr.update({f"down/g/{k}": npy(v.grad) for k, v in dn.named_parameters()})

    torch.manual_seed(303)
    oc = OutConv(8, 1)
    arr.update(sd_arrays(oc, "outc/p/"))
    xo = torch.randn(2, 8, 6, 6, requires_grad=True)
    yo = oc(xo)
    (yo * yo).sum().backward()
    arr.update({"outc/x": npy(xo), "outc/y": npy(yo), "outc/gx": npy(xo.grad),
                "outc/gw": npy(oc.conv.weight.grad), "outc/gb": npy(oc.conv.bias.grad)})

    torch.manual_seed(304)
    sa = SpatialAttention()
    arr.update(sd_arrays(sa, "att/p/"))
    xs = torch.randn(2, 16, 6, 6)
    with torch.no_grad():
        arr.update({"att/x": npy(xs), "att/y": npy(sa(xs))})
    save("blocks", **arr)


# ---------------------------------------------------------------- 4/6. model + one optimisation step
def gen_model(tag, use_skip, seed, lstm_layers=1, use_attention=False, base_ch=4):
    torch.manual_seed(seed)
    model = TemporalUNetDualView(1, 1, base_ch=base_ch, lstm_layers=lstm_layers,
                                 use_skip_lstm=use_skip, use_attention=use_attention)
    B, T, H, W = 2, 3, 32, 32
    x = torch.rand(B, T, 2, H, W)
    y = torch.rand(B, T, 1, H, W) * 2 - 1
    mask = (torch.rand(B, T, 1, H, W) > 0.3).float()
    arr = sd_arrays(model, "p/")
    arr.update({"x": npy(x), "y": npy(y), "mask": npy(mask)})

    # eval-mode forward on the initial weights (+ stateful continuation, no-skip variant only)
    model.eval()
    with torch.no_grad():
        outs, st = model(x)
        arr["out_eval"] = np.stack([npy(o) for o in outs], axis=1)
        outs2, _ = model(x[:, :2], st)
        arr["out_eval_cont"] = np.stack([npy(o) for o in outs2], axis=1)
        for li, (h, c) in enumerate(st):
            arr[f"state_h/{li}"], arr[f"state_c/{li}"] = npy(h), npy(c)

    # one full optimisation step exactly as main.py:91-108 (use_mask=True variant)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)     # main.py:275
    opt.zero_grad(set_to_none=True)
    output, _ = model(x)
    y_pred = torch.stack(output, dim=1)
    loss = ref_main.compute_loss(y_pred, y, mask, True)
    loss.backward()
    arr["out_train"] = npy(y_pred)
    arr["loss"] = npy(loss)
    arr.update({f"g/{k}": npy(v.grad) for k, v in model.named_parameters()})
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)               # main.py:106
    arr["grad_norm"] = npy(gn)
    opt.step()
    arr.update(sd_arrays(model, "p_after/"))
    save(tag, **arr)


# ---------------------------------------------------------------- 5. loss
def gen_loss():
    torch.manual_seed(0)
    yp = torch.randn(2, 3, 1, 8, 8, requires_grad=True)
    y = torch.randn(2, 3, 1, 8, 8)
    mask = (torch.rand(2, 3, 1, 8, 8) > 0.5).float()
    arr = {"y_pred": npy(yp), "y": npy(y), "mask": npy(mask)}
    l_un = ref_main.compute_loss(yp, y, mask, use_mask=False)
    g_un, = torch.autograd.grad(l_un, yp)
    l_m = ref_main.compute_loss(yp, y, mask, use_mask=True)
    g_m, = torch.autograd.grad(l_m, yp)
    arr.update({"loss_unmasked": npy(l_un), "grad_unmasked": npy(g_un),
                "loss_masked": npy(l_m), "grad_masked": npy(g_m)})
    print("loss known answers:", float(l_un.detach()), float(l_m.detach()))
    save("loss", **arr)


# ---------------------------------------------------------------- 7. dataset transform (8f-2)
def gen_dataset():
    rng = np.random.default_rng(7)
    N, T, H, W = 3, 4, 8, 8
    X = (rng.random((N, T, 2, H, W)) * 40).astype(np.float32)
    X[X < 8] = 0.0
    Y = (rng.standard_normal((N, T, 1, H, W)) * 3).astype(np.float32)
    path = os.path.join(OUT, "_tmp_ds.npz")
    np.savez(path, X=X, Y=Y)
    ds = NPZSequenceDataset(path)
    os.remove(path)
    x, y, m = ds[1]
    arr = {"X": X, "Y": Y, "x1": npy(x), "y1": npy(y), "mask1": npy(m),
           "norm_const": np.float64(ds.norm_const), "min_vel": np.float64(ds.min_vel),
           "max_vel": np.float64(ds.max_vel), "y_scale": np.float64(ds.y_scale),
           "trans_min": np.float64(ds.trans_min), "trans_max": np.float64(ds.trans_max),
           "denorm_y1": npy(ds.denormalize(y))}
    save("dataset", **arr)


# ---------------------------------------------------------------- 8. seed-regenerable reference runs at the BASELINE shapes
# These fixtures hold NO weights and NO inputs: the build's modules reproduce the reference's seeded initialisation bit for
# bit (tests/test_cabi_and_host.py::test_same_seed_gives_the_reference_initialisation) and the inputs come from seeded CPU
# generators, so a test re-creates both from the recorded seeds and checks them against the recorded checksums before it
# trusts the recorded outputs.  That keeps a base_ch=64 / B=32 reference run at ~1 MB instead of ~500 MB.
def checksum(tensors):
    s1 = sum(float(t.detach().double().sum()) for t in tensors)
    s2 = sum(float(t.detach().double().abs().sum()) for t in tensors)
    return np.array([s1, s2], dtype=np.float64)


def seeded_inputs(kind, B, T, H, W, seed):
    """Same generators as unet_convlstm_amd.engine.SyntheticSequences (imported from the BUILD, CPU tensors)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import unet_convlstm_amd as U
    d = U.SyntheticSequences(B, T, H, W, seed=seed, kind=kind, device="cpu")
    return d.x, d.y, d.mask


def rel_per_t(got, ref):
    return [float((got[:, t] - ref[:, t]).double().norm() / ref[:, t].double().norm()) for t in range(ref.shape[1])]


def mr_rel_per_t(got, ref):
    """Per-timestep rel-L2 of the MEAN-REMOVED frames (tests/conftest.py: mr_rel_l2_per_t is the same function)."""
    g = got.double() - got.double().mean(dim=(-2, -1), keepdim=True)
    r = ref.double() - ref.double().mean(dim=(-2, -1), keepdim=True)
    return [float((g[:, t] - r[:, t]).norm() / r[:, t].norm()) for t in range(ref.shape[1])]


def grad_stats(model):
    names = [k for k, _ in model.named_parameters()]
    return names, np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()], dtype=np.float64)


def flat_grad(model):
    return torch.cat([p.grad.detach().flatten().double() for _, p in model.named_parameters()])


def gen_seeded(tag, *, base_ch, skip, B, T, HW, kind, seed, use_mask, lstm_layers=1, with_step=True, with_autocast=True,
               with_fp16=False):
    torch.manual_seed(seed)
    model = TemporalUNetDualView(1, 1, base_ch=base_ch, lstm_layers=lstm_layers, use_skip_lstm=skip, use_attention=False)
    x, y, mask = seeded_inputs(kind, B, T, HW, HW, seed + 1)
    arr = {"cfg": np.array([base_ch, int(skip), B, T, HW, seed, int(use_mask), lstm_layers], dtype=np.int64),
           "kind": np.array(kind), "param_checksum": checksum(model.parameters()), "input_checksum": checksum([x, y, mask])}
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def eval_block(suffix):
        """Eval-mode forward in f32 plus the REFERENCE's own drift under autocast on the same weights, as raw per-timestep
        rel-L2 and MEAN-REMOVED (per-frame spatial mean subtracted from both sides: at random init the eval output is a DC
        offset -- the outc bias -- 60-230x larger than the spatial signal, so the raw figure cannot see a wrong signal)."""
        model.eval()
        with torch.no_grad():
            outs, _ = model(x)
            ref = torch.stack(outs, dim=1)
            arr["out_eval" + suffix] = npy(ref)
            sig = float((ref - ref.mean(dim=(-2, -1), keepdim=True)).std())
            print(f"{tag}{suffix}: eval |out| mean {float(ref.abs().mean()):.4f}, spatial signal std {sig:.5f}")
            for pre, adt in (("ac_", torch.bfloat16),) + ((("ac16_", torch.float16),) if with_fp16 else ()):
                with torch.autocast("cpu", dtype=adt):
                    o2, _ = model(x)
                got = torch.stack([o.float() for o in o2], dim=1)
                arr[pre + "eval_rel_l2_per_t" + suffix] = np.array(rel_per_t(got, ref), dtype=np.float64)
                arr[pre + "eval_mr_rel_l2_per_t" + suffix] = np.array(mr_rel_per_t(got, ref), dtype=np.float64)
                print(f"{tag}{suffix}: reference {pre}autocast EVAL drift per-t raw {[round(e, 6) for e in arr[pre + 'eval_rel_l2_per_t' + suffix]]} "
                      f"mean-removed {[round(e, 5) for e in arr[pre + 'eval_mr_rel_l2_per_t' + suffix]]}")

    eval_block("")
    # a second eval case with NON-TRIVIAL running statistics: three train-mode forwards first (3 x T momentum updates per
    # BatchNorm, train/unet.py:179,:196), their buffers stored (small: per-channel vectors) so that a test can load them
    model.train()
    with torch.no_grad():
        for _ in range(3):
            model(x)
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            arr["warm/" + k] = npy(v)
    eval_block("_warm")
    model.load_state_dict(sd0)
    if with_step:
        model.train()
        model.zero_grad(set_to_none=True)
        output, _ = model(x)
        y_pred = torch.stack(output, dim=1)
        loss = ref_main.compute_loss(y_pred, y, mask, use_mask)
        loss.backward()
        arr["out_train"], arr["loss"] = npy(y_pred), npy(loss)
        names, gnorms = grad_stats(model)
        arr["grad_names"], arr["grad_norms"] = np.array(names), gnorms
        arr["grad_norm"] = np.float64(np.sqrt((gnorms ** 2).sum()))
        g32 = flat_grad(model)
        per32 = [p.grad.detach().clone().double() for _, p in model.named_parameters()]
        for pre, adt in ((("ac16_", torch.float16),) if with_fp16 else ()):
            # the REFERENCE itself under fp16 autocast with a static loss scale of 2^14 (what any fp16 training does; without it
            # the 1/N loss gradient underflows binary16): the drift anchor for the fp16-MFMA twin kernels (configs[3])
            model.load_state_dict(sd0)
            model.train()
            model.zero_grad(set_to_none=True)
            with torch.autocast("cpu", dtype=adt):
                output, _ = model(x)
            y_ac = torch.stack([o.float() for o in output], dim=1)
            loss_ac = ref_main.compute_loss(y_ac, y, mask, use_mask)
            (loss_ac * 16384.0).backward()
            gac = flat_grad(model) / 16384.0
            arr[pre + "loss"] = npy(loss_ac)
            arr[pre + "grad_norm"] = np.float64(float(gac.norm()))
            arr[pre + "grad_rel_l2"] = np.float64(float((gac - g32).norm() / g32.norm()))
            arr[pre + "grad_cosine"] = np.float64(float(torch.dot(gac, g32) / (gac.norm() * g32.norm())))
            err_t = [float((y_ac[:, t] - y_pred[:, t]).detach().double().norm() / y_pred[:, t].detach().double().norm()) for t in range(T)]
            arr[pre + "out_rel_l2_per_t"] = np.array(err_t, dtype=np.float64)
            print(f"{tag}: reference fp16-autocast drift: out per-t {[round(e, 5) for e in err_t]}, loss {float(loss_ac):.6f} vs "
                  f"{float(loss):.6f}, grad rel-L2 {float(arr[pre + 'grad_rel_l2']):.4f} cosine {float(arr[pre + 'grad_cosine']):.5f}")
        if with_autocast:
            # the REFERENCE itself under bf16 autocast (its own fp32 <-> bf16 drift in train mode: the evidence the stated
            # tolerance of the HIP path is anchored in)
            model.load_state_dict(sd0)
            model.train()
            model.zero_grad(set_to_none=True)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                output, _ = model(x)
            y_ac = torch.stack([o.float() for o in output], dim=1)
            loss_ac = ref_main.compute_loss(y_ac, y, mask, use_mask)
            loss_ac.backward()
            arr["ac_out_train"], arr["ac_loss"] = npy(y_ac), npy(loss_ac)
            gac = flat_grad(model)
            arr["ac_grad_norm"] = np.float64(float(gac.norm()))
            arr["ac_grad_rel_l2"] = np.float64(float((gac - g32).norm() / g32.norm()))
            arr["ac_grad_cosine"] = np.float64(float(torch.dot(gac, g32) / (gac.norm() * g32.norm())))
            pt = []
            for (k, p_), r in zip(model.named_parameters(), per32):
                g = p_.grad.detach().double()
                pt.append(float((g - r).norm() / (r.norm() + 1e-30)))
            arr["ac_grad_rel_l2_per_tensor"] = np.array(pt, dtype=np.float64)
            arr["ac_grad_norms"] = np.array([float(p_.grad.double().norm()) for _, p_ in model.named_parameters()], dtype=np.float64)
            err_t = [float((y_ac[:, t] - y_pred[:, t]).double().norm() / y_pred[:, t].double().norm()) for t in range(T)]
            arr["ac_out_rel_l2_per_t"] = np.array(err_t, dtype=np.float64)
            print(f"{tag}: reference autocast drift: out per-t {[round(e, 5) for e in err_t]}, loss {float(loss_ac):.6f} vs "
                  f"{float(loss):.6f}, grad rel-L2 {float(arr['ac_grad_rel_l2']):.4f} cosine {float(arr['ac_grad_cosine']):.5f}")
    save(tag, **arr)


def gen_cfg0():
    """BASELINE configs[0] plumbing: 1-layer ConvLSTM(2,16) + 1x1 head on Moving-MNIST-shaped blobs, built from the
    reference's own ConvLSTM and OutConv classes (the reference ships no digits model: SURVEY.md section 8d cfg 1)."""
    torch.manual_seed(700)
    lstm = ConvLSTM(2, 16, num_layers=1)
    head = OutConv(16, 1)
    x, y, mask = seeded_inputs("blobs", 8, 20, 64, 64, 701)
    outs, st = lstm([x[:, t] for t in range(x.shape[1])])
    y_pred = torch.stack([head(o) for o in outs], dim=1)
    loss = ref_main.compute_loss(y_pred, y, mask, True)
    loss.backward()
    arr = {"param_checksum": checksum(list(lstm.parameters()) + list(head.parameters())), "input_checksum": checksum([x, y, mask]),
           "out": npy(y_pred), "loss": npy(loss), "h_final": npy(st[0][0]), "c_final": npy(st[0][1])}
    arr.update({f"g/lstm.{k}": npy(v.grad) for k, v in lstm.named_parameters()})
    arr.update({f"g/head.{k}": npy(v.grad) for k, v in head.named_parameters()})
    save("ref_cfg0_convlstm_head", **arr)


def gen_resnet_lstms():
    """the five ConvLSTM(ch, ch, num_layers=2) of train/resnet18.py:49-54,:71-74 driven exactly as
    its forward does (:101-104, :120-128): per-timestep views of a [B*T]-flattened feature tensor, list in, list out,
    torch.stack + view back.  The encoder is third-party and absent; seeded random features stand in for it."""
    arr = {}
    B, T = 2, 3
    for ch, hw, seed in ((64, 16, 800), (64, 8, 801), (128, 8, 802), (256, 4, 803), (512, 2, 804)):
        torch.manual_seed(seed)
        lstm = ConvLSTM(input_dim=ch, hidden_dim=ch, num_layers=2, kernel_size=3)
        feat = torch.randn(B * T, ch, hw, hw, generator=torch.Generator().manual_seed(seed + 50)).requires_grad_(True)
        feat_seq = feat.view(B, T, ch, hw, hw)
        lstm_in = [feat_seq[:, t] for t in range(T)]
        lstm_out_list, _ = lstm(lstm_in)
        out = torch.stack(lstm_out_list, dim=1).view(B * T, ch, hw, hw)
        (out * out).sum().backward()
        tag = f"{ch}_{hw}"
        arr[f"{tag}/param_checksum"] = checksum(lstm.parameters())
        arr[f"{tag}/feat_checksum"] = checksum([feat])
        arr[f"{tag}/out"] = npy(out)
        arr[f"{tag}/gfeat"] = npy(feat.grad)
        arr[f"{tag}/grad_norms"] = np.array([float(p.grad.double().norm()) for p in lstm.parameters()], dtype=np.float64)
        arr[f"{tag}/gbias0"] = npy(lstm.layers[0].conv.bias.grad)
        arr[f"{tag}/gw1_slice"] = npy(lstm.layers[1].conv.weight.grad[:8, :8])
    save("ref_resnet_lstms", **arr)


def gen_cell_k():
    """ConvLSTM with 5x5 and 7x7 gate convolutions (the reference accepts any kernel_size, train/unet.py:15-19; every call
    site uses 3): T=3 sequence, outputs, final state, all gradients."""
    arr = {}
    for k, seed in ((5, 1000), (7, 1001)):
        torch.manual_seed(seed)
        lstm = ConvLSTM(4, 8, num_layers=1, kernel_size=k)
        xs = [torch.randn(2, 4, 10, 9, requires_grad=True) for _ in range(3)]
        outs, st = lstm(xs)
        (sum((o * o).sum() for o in outs) * 0.5 + st[0][1].sum()).backward()
        arr.update({f"k{k}/p/{n}": npy(v) for n, v in lstm.state_dict().items()})
        arr.update({f"k{k}/g/{n}": npy(v.grad) for n, v in lstm.named_parameters()})
        arr[f"k{k}/x"] = np.stack([npy(x) for x in xs])
        arr[f"k{k}/gx"] = np.stack([npy(x.grad) for x in xs])
        arr[f"k{k}/out"] = np.stack([npy(o) for o in outs])
        arr[f"k{k}/c_final"] = npy(st[0][1])
    save("cell_k", **arr)


SEEDED = {
    # BASELINE configs[1] at the benchmark's own batch (the kernel plan the driver times): base_ch 64 + skip LSTMs, B=32
    "ref_cfg1_b32": dict(base_ch=64, skip=True, B=32, T=2, HW=64, kind="uniform", seed=900, use_mask=False),
    # SURVEY 8(d) cfg 2 names "P and S": the reference's class defaults (train/unet.py:132: base_ch 32, no skip LSTMs) at the same batch
    "ref_cfgS_b32": dict(base_ch=32, skip=False, B=32, T=2, HW=64, kind="uniform", seed=905, use_mask=False),
    # well-sized autocast anchor (BatchNorm statistics over >= 256 values per channel at every level)
    "ref_autocast_b16": dict(base_ch=8, skip=True, B=16, T=3, HW=64, kind="uniform", seed=910, use_mask=True, with_fp16=True),
    # Moving-MNIST-shaped blobs through the full model
    "ref_blobs64": dict(base_ch=16, skip=True, B=8, T=4, HW=64, kind="blobs", seed=920, use_mask=True),
    # configs[2]: cloud 128x128
    "ref_cloud128": dict(base_ch=16, skip=True, B=4, T=3, HW=128, kind="blobs", seed=930, use_mask=True),
    # configs[3]: 256x256
    "ref_256": dict(base_ch=8, skip=True, B=2, T=2, HW=256, kind="uniform", seed=940, use_mask=False, with_fp16=True),
    # configs[4]: 512x512 rollout (inference only).  Two seeds: 950 draws a small output bias (|out| = 0.004), so its RAW
    # rel-L2 reads the signal-relative error (1.05e-2 on the HIP path in round 2); 952 draws a DC offset of 0.3.  Both are
    # judged on the mean-removed metric against the reference's own autocast eval drift.
    "ref_512": dict(base_ch=8, skip=True, B=1, T=3, HW=512, kind="uniform", seed=952, use_mask=False, with_step=False),
    "ref_512_s950": dict(base_ch=8, skip=True, B=1, T=3, HW=512, kind="uniform", seed=950, use_mask=False, with_step=False),
}


if __name__ == "__main__":
    only = [a for a in sys.argv[2:]]
    if only:
        for name in only:
            if name in SEEDED:
                gen_seeded(name, **SEEDED[name])
            else:
                globals()[name]()
        sys.exit(0)
    for name, kw in SEEDED.items():
        gen_seeded(name, **kw)
    gen_cfg0()
    gen_resnet_lstms()
    gen_cell()
    gen_seq()
    gen_blocks()
    gen_loss()
    gen_model("model_noskip", use_skip=False, seed=400)
    gen_model("model_skip", use_skip=True, seed=401)
    gen_model("model_2layer_att", use_skip=False, seed=402, lstm_layers=2, use_attention=True, base_ch=2)
    gen_dataset()
