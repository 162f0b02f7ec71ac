"""Weight-panel pack / unpack kernels against a plain numpy restatement of the descriptor's index map (bit-exact: the panel
is the f32 weight rounded to bf16 at the mapped position, zero elsewhere), for every panel family the model uses --
forward conv (one and two sources), ConvLSTM gate-interleaved (full and the hoisted x / h halves), the transposed
input-gradient panels (conv, ConvLSTM gates), ConvTranspose forward / input-gradient, the pre-gathered first layer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import unet_convlstm_amd as U
    from unet_convlstm_amd import ops
    L = U._lib

DEV = "cuda"


def index_map(d):
    """(valid [N,Ktot] bool, offset [N,Ktot] int64) of the descriptor, straight from include/uclstm.h's definition."""
    N, K = d.N, d.Ktot
    n = np.arange(N)[:, None]
    k = np.arange(K)[None, :]
    tapn = np.zeros_like(n)
    if d.n_mode == L.NMODE_IDENTITY:
        n_ent, ok_n = n, n < d.n_valid
    elif d.n_mode == L.NMODE_LSTM:
        hb, gate, j = n >> 6, (n & 63) >> 4, n & 15
        hc = hb * 16 + j
        n_ent, ok_n = gate * d.n_valid + hc, hc < d.n_valid
    else:
        tapn = n // d.n_cp
        n_ent = n - tapn * d.n_cp
        ok_n = n_ent < d.n_valid
    per_tap = d.kseg[0] + d.kseg[1]
    tap = k // per_tap
    kr = k - tap * per_tap
    s = (kr >= d.kseg[0]).astype(np.int64)
    c = np.where(s == 1, kr - d.kseg[0], kr)
    cvalid = np.where(s == 1, d.cvalid[1], d.cvalid[0])
    choff = np.where(s == 1, d.choff[1], d.choff[0])
    if d.k_mode == L.KMODE_IDENTITY:
        ok_k, k_ent = c < cvalid, choff + c
        ntap = d.taps
    elif d.k_mode == L.KMODE_GATES:
        gate, hc = c // d.k_hdp, c % d.k_hdp
        ok_k, k_ent = (gate < 4) & (hc < d.k_hd), choff + gate * d.k_hd + hc
        ntap = d.taps
    else:
        tk = c // d.k_hd
        ok_k, k_ent, tap, ntap = tk < d.k_hdp, c - tk * d.k_hd, tk, d.k_hdp
    tap_eff = (ntap - 1 - tap) if d.tap_flip else tap
    off = n_ent * d.stride_n + k_ent * d.stride_k + tap_eff * d.stride_tap + tapn * d.stride_ntap
    valid = ok_n & ok_k
    return valid, np.where(valid, off, 0)


CASES = {
    "conv fwd 1 source": lambda: (ops.conv_pack_desc(40, 24, [24], [24]), (40, 24, 3, 3), 0),
    "conv fwd 2 sources": lambda: (ops.conv_pack_desc(72, 80, [56, 24], [56, 24]), (72, 80, 3, 3), 0),
    "conv fwd wide": lambda: (ops.conv_pack_desc(128, 320, [320], [320]), (128, 320, 3, 3), 0),
    "conv dgrad": lambda: (ops.conv_dgrad_pack_desc(72, 80, 56), (72, 80, 3, 3), 0),
    "conv dgrad second source": lambda: (ops.conv_dgrad_pack_desc(72, 80, 24), (72, 80, 3, 3), 56 * 9),
    "conv dgrad wide": lambda: (ops.conv_dgrad_pack_desc(200, 136, 136), (200, 136, 3, 3), 0),
    "lstm fwd": lambda: (ops.lstm_pack_desc(40, 24), (160, 64, 3, 3), 0),
    "lstm x half": lambda: (ops.lstm_half_pack_desc(40, 24, "x"), (160, 64, 3, 3), 0),
    "lstm h half": lambda: (ops.lstm_half_pack_desc(40, 24, "h"), (160, 64, 3, 3), 0),
    "lstm dgrad h": lambda: (ops.lstm_dgrad_pack_desc(40, 24, 40), (160, 64, 3, 3), 24 * 9),
    "lstm dgrad x": lambda: (ops.lstm_dgrad_pack_desc(72, 136, 136), (288, 208, 3, 3), 0),
    "convT fwd": lambda: (ops.convt_pack_desc(48, 24), (48, 24, 2, 2), 0),
    "convT dgrad": lambda: (ops.convt_dgrad_pack_desc(48, 24), (48, 24, 2, 2), 0),
    "first layer": lambda: (ops.im2col_pack_desc(24, 2, 24), (24, 2, 3, 3), 0),
    "lstm 1x1": lambda: (ops.lstm_pack_desc(16, 8, 1), (64, 24, 1, 1), 0),
}


@pytest.mark.parametrize("name", list(CASES))
def test_pack_matches_the_descriptor_index_map(name):
    d, wshape, elem_off = CASES[name]()
    g = torch.Generator().manual_seed(1)
    w = torch.randn(wshape, generator=g)
    wp = ops.pack_weights(d, w.to(DEV), elem_off).cpu()
    valid, off = index_map(d)
    flat = w.flatten().numpy()
    want = np.where(valid, flat[np.minimum(off + elem_off, flat.size - 1)], 0.0).astype(np.float32)
    assert int(valid.sum()) > 0 and (off + elem_off)[valid].max() < flat.size
    want = torch.from_numpy(want).to(torch.bfloat16)
    assert wp.shape == want.shape
    assert torch.equal(wp.view(torch.int16), want.view(torch.int16)), f"{name}: {int((wp != want).sum())} panel elements differ"


@pytest.mark.parametrize("name,nslab", [("conv fwd 1 source", 1), ("conv fwd 2 sources", 5), ("conv fwd 1 source", 170), ("lstm fwd", 3),
                                        ("convT fwd", 2), ("first layer", 300)])
def test_unpack_adds_the_slabs_into_the_reference_layout(name, nslab):
    """uclstm_unpack_wgrad: grad = (accumulate ? grad : 0) + sum of slabs, scattered through the index map; with more than 64
    slabs they are folded in place first (slab_fold_kernel, replaces round 1's ATen reduction)."""
    d, wshape, _ = CASES[name]()
    if name == "lstm fwd":
        d = ops.lstm_wgrad_unpack_desc(40, 24)
    g = torch.Generator().manual_seed(2)
    slabs = torch.randn((nslab, d.N, d.Ktot), generator=g)
    valid, off = index_map(d)
    like = torch.zeros(wshape)
    base = torch.randn(wshape, generator=g)
    tot = slabs.double().sum(0).numpy()
    want = np.zeros(like.numel(), dtype=np.float64)
    np.add.at(want, off[valid], tot[valid])
    for acc in (0, 1):
        grad = base.clone().to(DEV)
        sd = slabs.clone().to(DEV)
        ns, st = ops._slabs_of(sd)
        L.check(L.lib.uclstm_unpack_wgrad(d, sd.data_ptr(), ns, st, grad.data_ptr(), acc, None), "unpack")
        ref = torch.from_numpy(want).view(wshape) + (base.double() if acc else 0)
        torch.testing.assert_close(grad.cpu().double(), ref, rtol=1e-5, atol=1e-4 if nslab > 64 else 1e-5)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_batched_packing_equals_the_single_panel_kernels(dtype):
    """uclstm_pack_weights_batched (one launch per kernel family over a device job table, the look-ahead packing of a training
    step) must write bit for bit what uclstm_pack_weights writes panel by panel -- every family, both 16-bit types."""
    dt = torch.bfloat16 if dtype == "bf16" else torch.float16
    g = torch.Generator().manual_seed(3)
    items = []
    for name, mk in CASES.items():
        d, wshape, off = mk()
        w = torch.randn(wshape, generator=g).to(DEV)
        items.append(((w.data_ptr(), off, bytes(d), dt), d, w, off))
    batch = ops._PackBatch(items)
    fams = sorted({f for launches in batch.segments for _, f, _, _, _ in launches})
    assert len(batch.segments) > 1 and fams == [0, 1, 2, 3], fams       # every family the model's panels fall into (4: unused)
    for wp, _ in batch.panels.values():
        wp.fill_(7.0)
    batch.launch(torch.cuda.current_stream())
    assert all(e is not None for e in batch.events)
    torch.cuda.synchronize()
    for (key, d, w, off), name in zip(items, CASES):
        want = ops.pack_weights(d, w, off, dtype=dt)
        got = batch.panels[key][0]
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), name


def test_look_ahead_packing_hands_out_the_batched_panels():
    """prepack_begin() replays the previous step's pack calls as batched launches; pack_weights() then returns those panels."""
    g = torch.Generator().manual_seed(4)
    calls = []
    for name in ("conv fwd 2 sources", "conv dgrad", "lstm fwd", "convT fwd", "first layer"):
        d, wshape, off = CASES[name]()
        calls.append((d, torch.randn(wshape, generator=g).to(DEV), off))
    try:
        ops.prepack_begin()                                   # step 0: records
        first = [ops.pack_weights(d, w, off).clone() for d, w, off in calls]
        ops.prepack_end()
        for _, w, _ in calls:
            w.mul_(0.5)                                       # "optimiser step"
        ops.prepack_begin()                                   # step 1: packs ahead
        assert ops._PACK_BATCH is not None and len(ops._PACK_READY) == len(calls)
        second = [ops.pack_weights(d, w, off) for d, w, off in calls]
        assert all(p.data_ptr() == ops._PACK_BATCH.panels[k][0].data_ptr() for p, k in zip(second, ops._PACK_BATCH.sig))
        ops.prepack_end()
        torch.cuda.synchronize()
        for a, b, (d, w, off) in zip(first, second, calls):
            assert torch.equal(b.view(torch.int16), ops.pack_weights(d, w, off).view(torch.int16))
            assert not torch.equal(a, b)
    finally:
        ops.prepack_end()
        ops._PACK_PLAN.clear()
