"""CPU-only tests: the C-ABI library loads and exports every symbol of include/uclstm.h, argument contracts are
enforced before any launch, the host-side mirror keeps the reference's module surface, and the flat-buffer
data-parallel path is correct with world_size 2 on gloo.  No kernel is launched here."""
import ctypes as C
import inspect
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, sub

import unet_convlstm_amd as U
from unet_convlstm_amd import _lib as L
from unet_convlstm_amd import ops
from unet_convlstm_amd.optim import FlatParams
from unet_convlstm_amd.ddp import FlatDDP


# ---------------------------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------------------------
def test_library_exports_every_header_symbol():
    syms = L.header_symbols()
    assert len(syms) >= 29
    bound = set(L._PROTOS) | {n + "_f16" for n in L.F16_TWINS}          # the fp16 twins share their prototype's binding
    assert set(syms) == bound, (set(syms) ^ bound)
    assert len(L.F16_TWINS) == 29 and all(n in L._PROTOS for n in L.F16_TWINS)
    assert L.kernels(torch.bfloat16) is L.lib and L.kernels(torch.float16).uclstm_igemm_fwd is not L.lib.uclstm_igemm_fwd
    assert L.kernels(torch.float16).uclstm_bn_finalize is L.lib.uclstm_bn_finalize            # shared (f32-only) entry point
    raw = C.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), s
    hdr = open(L.HEADER_PATH).read()
    assert L.lib.uclstm_abi_version() == L.ABI_VERSION == int(re.search(r"#define UCLSTM_ABI_VERSION (\d+)", hdr).group(1))
    assert L.lib.uclstm_build_arch() == b"gfx950"


def test_library_reports_the_sources_of_this_tree():
    """Ship exactly what is tracked: the loaded library was built from csrc/*, include/uclstm.h and build.py's flags as they are
    in this tree (the loader would have refused it otherwise), and the sidecar binds that hash to this very file."""
    from unet_convlstm_amd import build as B
    assert L.lib.uclstm_source_hash().decode() == B.source_hash()
    assert B.library_hash(L.LIB_PATH) == B.source_hash() and not B.needs_rebuild()
    stray = [f for f in os.listdir(os.path.join(ROOT, "unet-convlstm_amd", "build")) if f.endswith(".so") or "whatif" in f] \
        if os.path.isdir(os.path.join(ROOT, "unet-convlstm_amd", "build")) else []
    assert not stray, f"stale A/B builds would ship to the GPU box: {stray} (keep them under unet-convlstm_amd/ab/)"


def test_build_recompiles_when_a_source_byte_changes(tmp_path):
    """build() decides by content hash, not by mtime: an edited source byte (with the old mtime restored) recompiles, an
    untouched tree does not, and a library next to a foreign sidecar counts as stale."""
    from unet_convlstm_amd import build as B
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    hdr = tmp_path / "api.h"
    hdr.write_text("int probe_value(void);\n")
    src = csrc / "probe.hip"
    src.write_text('#include <hip/hip_runtime.h>\n__global__ void k(int* p) { *p = 1; }\nextern "C" int probe_value(void) { return 41; }\n')
    lib = str(tmp_path / "libprobe.so")
    kw = dict(csrc=str(csrc), header=str(hdr), lib=lib, sources=["probe.hip"], f16_sources=[], verbose=False)
    B.build(**kw)
    h0 = B.source_hash(str(csrc), str(hdr), ["probe.hip"], [])
    so = C.CDLL(lib)
    so.uclstm_source_hash.restype = C.c_char_p
    assert so.uclstm_source_hash().decode() == h0 == B.library_hash(lib) and so.probe_value() == 41
    t_lib, t_obj = os.path.getmtime(lib), os.path.getmtime(tmp_path / "build" / "probe.o")
    B.build(**kw)                                                    # nothing changed: nothing rebuilt
    assert os.path.getmtime(lib) == t_lib and os.path.getmtime(tmp_path / "build" / "probe.o") == t_obj
    st = os.stat(src)
    src.write_text(src.read_text().replace("return 41", "return 42"))
    os.utime(src, (st.st_atime, st.st_mtime))                        # an mtime-based check would not see this edit
    assert B.needs_rebuild(lib, str(csrc), str(hdr), ["probe.hip"], [])
    B.build(**kw)
    h1 = B.source_hash(str(csrc), str(hdr), ["probe.hip"], [])
    assert h1 != h0 and B.library_hash(lib) == h1
    lib2 = str(tmp_path / "libprobe2.so")                            # dlopen caches by path: load the rebuilt file under a new name
    os.link(lib, lib2)
    so2 = C.CDLL(lib2)
    so2.uclstm_source_hash.restype = C.c_char_p
    assert so2.uclstm_source_hash().decode() == h1 and so2.probe_value() == 42
    with open(lib, "ab") as f:                                       # a library that is not the one the sidecar describes
        f.write(b"\0")
    assert B.library_hash(lib) is None and B.needs_rebuild(lib, str(csrc), str(hdr), ["probe.hip"], [])


def test_struct_layouts_match_the_header():
    # sizes follow from the header's field lists (natural alignment, 8-byte pointers)
    assert C.sizeof(L.Src) == 32
    assert C.sizeof(L.Seg) == 48
    assert C.sizeof(L.PackDesc) == 4 * 17 + 4 + 8 * 4      # 17 ints (+4 pad) + 4 int64
    d = L.IgemmDesc()
    assert C.sizeof(d) % 8 == 0 and type(d).src.offset == 32 and type(d).wp.offset == 96
    # v7: pre_add sits between the LSTM outputs and the split-K block
    assert type(d).pre_add.offset == type(d).gates_out.offset + 8 and type(d).acc_out.offset == type(d).pre_add.offset + 8
    h = ops.lstm_half_pack_desc(40, 24, "h")
    x = ops.lstm_half_pack_desc(40, 24, "x")
    full = ops.lstm_pack_desc(40, 24)
    assert h.N == x.N == full.N and h.Ktot + x.Ktot == full.Ktot and (h.choff[0], x.choff[0]) == (24, 0)


def test_argument_contracts_are_checked_before_launch():
    # null / malformed descriptors must come back as UCLSTM_E_BADARG (-1) without touching a device
    assert L.lib.uclstm_igemm_fwd(None, None) == -1
    assert L.lib.uclstm_igemm_wgrad(None, None) == -1
    d = L.IgemmDesc()
    d.n_img, d.H, d.W, d.groups, d.ktap, d.scale, d.pad, d.nsrc = 4, 8, 8, 3, 3, 1, 1, 1      # 4 % 3 != 0
    assert L.lib.uclstm_igemm_fwd(C.byref(d), None) == -1
    assert L.lib.uclstm_igemm_tiles_per_group(4, 8, 8, 3, 128) == -1
    assert L.lib.uclstm_igemm_tiles_per_group(640, 64, 64, 20, 128) == 1024     # 128 x 128-pixel blocks
    assert L.lib.uclstm_igemm_tiles_per_group(640, 64, 64, 20, 64) == 512       # narrow panel: 64 x 256-pixel blocks
    assert L.lib.uclstm_igemm_tiles_per_group(4, 8, 8, 1, 128) == 2             # small layer: 128 x 128-pixel blocks
    assert L.lib.uclstm_bn_apply_relu(None, None, None, None, 10, 10, 8, None) == -1
    assert L.lib.uclstm_maxpool2_fwd(None, None, 1, 4, 4, 8, None) == -1
    assert L.lib.uclstm_adamw_step(None, None, None, None, 10, None, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == -1
    pd = ops.conv_pack_desc(8, 8, [8], [8])
    pd.Ktot += 64
    assert L.lib.uclstm_pack_weights(C.byref(pd), None, None, None) == -1
    with pytest.raises(L.UclstmError):
        L.check(-1, "x")


def test_pack_descriptors_geometry():
    d = ops.conv_pack_desc(40, 24 + 8, [24, 8], [24, 8])
    assert (d.N, d.taps, d.kseg[0], d.kseg[1], d.Ktot) == (40, 9, 64, 64, 9 * 128)
    assert (d.choff[0], d.choff[1], d.stride_n, d.stride_k) == (0, 24, 32 * 9, 9)
    d = ops.lstm_pack_desc(1024, 1024)
    assert (d.N, d.Ktot, d.n_mode) == (4096, 9 * 2048, L.NMODE_LSTM)
    d = ops.lstm_pack_desc(5, 4)
    assert (d.N, d.kseg[0], d.kseg[1]) == (64, 64, 64)
    d = ops.lstm_dgrad_pack_desc(5, 4, 5)
    assert (d.N, d.kseg[0], d.k_hdp, d.k_hd, d.tap_flip) == (8, 64, 8, 5, 1)
    d = ops.lstm_wgrad_unpack_desc(1024, 1024)
    assert (d.N, d.Ktot, d.stride_ntap) == (4096, 9 * 2048, 1024 * 2048 * 9)
    d = ops.convt_pack_desc(1024, 512)
    assert (d.N, d.Ktot, d.n_cp, d.stride_n, d.stride_k, d.stride_ntap) == (2048, 1024, 512, 4, 2048, 1)
    d = ops.convt_dgrad_pack_desc(1024, 512)
    assert (d.N, d.taps, d.Ktot) == (1024, 4, 4 * 512)
    d = ops.im2col_pack_desc(64, 2, 24)
    assert (d.N, d.Ktot, d.k_mode, d.k_hd, d.k_hdp) == (64, 64, L.KMODE_IM2COL, 2, 9)


# ---------------------------------------------------------------------------------------------
# module surface
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kw", [
    ("model_noskip", dict(base_ch=4, use_skip_lstm=False)),
    ("model_skip", dict(base_ch=4, use_skip_lstm=True)),
    ("model_2layer_att", dict(base_ch=2, lstm_layers=2, use_attention=True)),
])
def test_state_dict_keys_shapes_dtypes_equal_the_reference(name, kw):
    ref = sub(load_golden(name), "p/")
    m = U.TemporalUNetDualView(1, 1, **kw)
    sd = m.state_dict()
    assert list(sd.keys()) == sorted(sd.keys(), key=list(sd.keys()).index)       # deterministic
    assert set(sd.keys()) == set(ref.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k].shape) and v.dtype == ref[k].dtype, k
    m.load_state_dict(ref, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k]), k              # bit-exact round trip of the reference layout


def test_constructor_signatures_and_public_attributes():
    def params(cls):
        return [(p.name, p.default) for p in list(inspect.signature(cls.__init__).parameters.values())[1:]]
    assert params(U.ConvLSTMCell) == [("input_dim", inspect._empty), ("hidden_dim", inspect._empty), ("kernel_size", 3), ("bias", True)]
    assert params(U.ConvLSTM) == [("input_dim", inspect._empty), ("hidden_dim", inspect._empty), ("num_layers", 1), ("kernel_size", 3)]
    assert params(U.DoubleConv) == params(U.Down) == params(U.Up) == params(U.OutConv) == [("in_ch", inspect._empty), ("out_ch", inspect._empty)]
    assert params(U.SpatialAttention) == [("kernel_size", 7)]
    assert params(U.TemporalUNetDualView) == [("in_channels_per_sat", 1), ("out_channels", 1), ("base_ch", 32), ("lstm_layers", 1),
                                             ("use_skip_lstm", False), ("use_attention", False)]
    cell = U.ConvLSTMCell(3, 5)
    assert cell.hidden_dim == 5 and isinstance(cell.conv, torch.nn.Conv2d) and cell.conv.weight.shape == (20, 8, 3, 3)
    lstm = U.ConvLSTM(3, 5, num_layers=2)
    assert len(lstm.layers) == 2 and lstm.layers[1].conv.in_channels == 10
    m = U.TemporalUNetDualView(use_skip_lstm=True)
    assert m.use_skip_lstm and not m.use_attention and callable(m.encode_once) and U.UNet is U.TemporalUNetDualView
    up = U.Up(16, 8)
    assert up.up.weight.shape == (16, 8, 2, 2)


def test_same_seed_gives_the_reference_initialisation():
    """Parameter containers are stock torch modules created in the reference's order, so a seeded construction
    reproduces the reference's random init bit-for-bit (fixture 'model_skip' was built with seed 401)."""
    ref = sub(load_golden("model_skip"), "p/")
    torch.manual_seed(401)
    m = U.TemporalUNetDualView(1, 1, base_ch=4, lstm_layers=1, use_skip_lstm=True, use_attention=False)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k]), k


def test_no_cpu_fallback():
    m = U.TemporalUNetDualView(base_ch=4)
    with pytest.raises(U.UclstmError):
        m(torch.rand(1, 2, 2, 16, 16))
    with pytest.raises(U.UclstmError):
        U.ConvLSTMCell(2, 4)(torch.rand(1, 2, 8, 8))
    with pytest.raises(U.UclstmError):
        U.compute_loss(torch.rand(1, 2, 1, 8, 8), torch.rand(1, 2, 1, 8, 8))
    with pytest.raises(RuntimeError):
        U.FusedAdamW([torch.nn.Parameter(torch.zeros(3))])


def test_npz_dataset_matches_reference_fixture(tmp_path):
    g = load_golden("dataset")
    path = tmp_path / "ds.npz"
    np.savez(path, X=g["X"].numpy(), Y=g["Y"].numpy())
    ds = U.NPZSequenceDataset(str(path))
    for k in ("norm_const", "min_vel", "max_vel", "y_scale", "trans_min", "trans_max"):
        assert abs(getattr(ds, k) - float(g[k])) <= 1e-6 * max(1.0, abs(float(g[k]))), k
    x, y, m = ds[1]
    torch.testing.assert_close(x, g["x1"])
    torch.testing.assert_close(y, g["y1"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(m, g["mask1"])
    torch.testing.assert_close(ds.denormalize(y).double(), g["denorm_y1"].double(), rtol=1e-5, atol=1e-5)
    assert len(ds) == 3


def test_synthetic_sequences_shapes_and_determinism():
    a = U.SyntheticSequences(2, 5, 32, 32, seed=3, kind="blobs", device="cpu")
    b = U.SyntheticSequences(2, 5, 32, 32, seed=3, kind="blobs", device="cpu")
    assert a.x.shape == (2, 5, 2, 32, 32) and a.y.shape == (2, 5, 1, 32, 32) and a.mask.shape == a.y.shape
    assert torch.equal(a.x, b.x) and torch.equal(a.y, b.y)
    assert torch.equal(a.x[:, :, 0], a.x[:, :, 1])                       # frame duplicated into both views
    assert float(a.y.abs().max()) <= 1.0 and float(a.x.max()) <= 1.0
    u = U.SyntheticSequences(2, 3, 16, 16, seed=1, kind="uniform", device="cpu")
    assert 0.0 <= float(u.x.min()) and float(u.x.max()) < 1.0 and -1.0 <= float(u.y.min()) and float(u.y.max()) < 1.0


# ---------------------------------------------------------------------------------------------
# flat parameters + data-parallel gradient exchange (gloo, world_size 2)
# ---------------------------------------------------------------------------------------------
def test_flat_params_views_and_zero_grad():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    before = [p.detach().clone() for p in net.parameters()]
    fp = FlatParams(net.parameters())
    assert fp.numel == sum(p.numel() for p in net.parameters())
    for p, b, o in zip(net.parameters(), before, fp.offsets):
        assert torch.equal(p, b) and p.data_ptr() == fp.flat_p.data_ptr() + 4 * o
        assert p.grad.data_ptr() == fp.flat_g.data_ptr() + 4 * o
    net(torch.randn(4, 5)).sum().backward()
    assert float(fp.flat_g.abs().sum()) > 0
    for p, o in zip(net.parameters(), fp.offsets):                       # autograd accumulated IN PLACE into the flat buffer
        assert p.grad.data_ptr() == fp.flat_g.data_ptr() + 4 * o
    fp.zero_grad()
    assert float(fp.flat_g.abs().sum()) == 0


def _ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                          # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
    net.register_buffer("stat", torch.full((3,), float(rank)))
    fp = FlatParams(net.parameters())
    ddp = FlatDDP(net, fp, bucket_mb=30 * 4 / (1 << 20))   # >= 30 floats per bucket -> three buckets
    p0 = fp.flat_p.clone()
    launched_during_backward = []
    orig = ddp._launch

    def spy(bi):
        launched_during_backward.append(bi)
        orig(bi)
    ddp._launch = spy
    torch.manual_seed(7 + rank)
    x = torch.randn(8, 6)
    fp.zero_grad()
    ddp.reset()
    net(x).pow(2).mean().backward()
    n_early = len(launched_during_backward)
    ddp.finalize()
    # numpy copies: pickled by value (torch tensors would travel as shared-memory handles that die with this process)
    q.put((rank, p0.numpy().copy(), fp.flat_g.numpy().copy(), x.numpy().copy(), n_early, len(ddp.buckets), net.stat.numpy().copy()))
    dist.destroy_process_group()


def test_flat_ddp_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [tuple(torch.from_numpy(v) if isinstance(v, np.ndarray) else v for v in r) for r in res]
    (_, p_a, g_a, x_a, early_a, nb, stat_a), (_, p_b, g_b, x_b, early_b, _, stat_b) = res
    assert torch.equal(p_a, p_b)                           # parameters broadcast from rank 0
    assert torch.equal(stat_a, stat_b) and float(stat_a[0]) == 0.0      # buffers too
    assert torch.equal(g_a, g_b)                           # every rank holds the same averaged gradient
    assert nb >= 3 and early_a == nb and early_b == nb     # every bucket was launched from a backward hook (overlap)
    # reference: mean of the two ranks' local gradients
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
    fp = FlatParams(net.parameters())
    fp.flat_p.copy_(p_a)
    gs = []
    for x in (x_a, x_b):
        fp.zero_grad()
        net(x).pow(2).mean().backward()
        gs.append(fp.flat_g.clone())
    torch.testing.assert_close(g_a, (gs[0] + gs[1]) / 2, rtol=1e-6, atol=1e-7)


def _ddp_bf16_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
    fp = FlatParams(net.parameters())
    ddp = FlatDDP(net, fp, bucket_mb=30 * 4 / (1 << 20), grad_dtype=torch.bfloat16)
    torch.manual_seed(7 + rank)
    x = torch.randn(8, 6)
    fp.zero_grad()
    ddp.reset()
    net(x).pow(2).mean().backward()
    local = fp.flat_g.clone()
    ddp.finalize()
    q.put((rank, local.numpy().copy(), fp.flat_g.numpy().copy()))
    dist.destroy_process_group()


def test_flat_ddp_bf16_buckets_two_ranks_gloo():
    """grad_dtype=bfloat16: every bucket is exchanged as bf16 (half the bytes), result = mean of the bf16-rounded local
    gradients, identical on both ranks, back in the f32 gradient buffer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, la, ga), (_, lb, gb) = [tuple(torch.from_numpy(v) if isinstance(v, np.ndarray) else v for v in r) for r in res]
    assert torch.equal(ga, gb)
    want = ((la.bfloat16() + lb.bfloat16()).float() / 2)            # gloo sums in bf16, the mean is taken in f32
    torch.testing.assert_close(ga, want, rtol=1e-2, atol=1e-6)
    exact = (la + lb) / 2
    assert float((ga - exact).norm() / exact.norm()) <= 1e-2        # bf16 exchange: <= 2^-8 relative per element


def test_flat_ddp_counts_uses_of_side_written_gradients():
    """A parameter whose gradient is written by the operators themselves announces once per USE: the bucket may only be
    released after as many announcements as the forward pass reported uses (a module called twice under one backward),
    and ``no_sync()`` passes neither count nor launch."""
    if dist.is_initialized():
        dist.destroy_process_group()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(33500 + (os.getpid() % 2000))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        net = torch.nn.Linear(4, 3)
        fp = FlatParams(net.parameters())
        ddp = FlatDDP(net, fp, bucket_mb=1.0, first_bucket_mb=1e-9)
        launched = []
        orig = ddp._launch
        ddp._launch = lambda bi: (launched.append(bi), orig(bi))[1]
        w, b = net.weight, net.bias
        iw, ib = ddp._index_of[id(w)], ddp._index_of[id(b)]
        fp.zero_grad()
        ddp.reset()
        U.ops.note_use(w)
        U.ops.note_use(w)                                  # the module ran twice before backward
        U.ops.grad_written(w)
        assert not launched and not ddp._done[iw]          # first announcement of two: nothing may start
        U.ops.grad_written(w)
        assert ddp._done[iw]
        ddp._on_ready(ib)                                  # autograd's accumulator: once per backward, completes at once
        assert sorted(launched) == list(range(len(ddp.buckets)))
        ddp.finalize()
        with pytest.raises(RuntimeError, match="announced 3 gradients for 2 recorded uses"):
            U.ops.grad_written(w)                          # a third announcement for two uses is a protocol error
        # gradient accumulation: passes inside no_sync() are invisible
        launched.clear()
        ddp.reset()
        with ddp.no_sync():
            U.ops.note_use(w)
            U.ops.grad_written(w)
            ddp._on_ready(ib)
        assert not launched and ddp._uses[iw] == 0
        U.ops.note_use(w)
        U.ops.grad_written(w)
        ddp._on_ready(ib)
        assert sorted(launched) == list(range(len(ddp.buckets)))
        ddp.finalize()
        # a forward inside no_sync() whose backward runs outside (or a forward before reset()): the number of uses of this
        # pass is unknown, so a side announcement must NOT release the bucket on the first of T per-timestep gradients --
        # it is left to finalize() (uses 0 -> expected max(1, 0) = 1 released early, then asserted)
        launched.clear()
        ddp.reset()
        with ddp.no_sync():
            U.ops.note_use(w)
            U.ops.note_use(w)
        U.ops.grad_written(w)
        U.ops.grad_written(w)
        assert not launched and not ddp._done[iw]
        ddp.reset()
        U.ops.grad_written(w)                              # gradient of a forward that ran before reset(): never early
        assert not launched and not ddp._done[iw]
        ddp.finalize()
        assert sorted(launched) == list(range(len(ddp.buckets)))
        # one gradient buffer, one wrapper
        with pytest.raises(RuntimeError, match="already belong"):
            FlatDDP(net, fp)
        ddp.remove_hooks()
        assert not U.ops.USE_HOOKS and not U.ops.GRAD_SIDE_HOOKS
        # a wrapper dropped WITHOUT remove_hooks() takes its global hooks with it
        d2 = FlatDDP(net, fp)
        assert len(U.ops.USE_HOOKS) == 1 and len(U.ops.GRAD_SIDE_HOOKS) == 1
        for h in d2._hooks:
            h.remove()
        d2._hooks = []
        del d2
        import gc
        gc.collect()
        assert not U.ops.USE_HOOKS and not U.ops.GRAD_SIDE_HOOKS
        d3 = FlatDDP(net, fp)
        assert d3.describe() == {"buckets": len(d3.buckets), "allreduce_bytes": 4 * fp.numel, "dtype": "f32", "world_size": 1,
                                 "bucket_bytes": [4 * (hi - lo) for lo, hi in d3.ranges]}
        d3.remove_hooks()
    finally:
        dist.destroy_process_group()


def test_bench_self_launch_starts_n_ranks():
    """``python bench.py --gpus 2`` without torchrun must start two ranks itself (the driver's N>1 invocation) -- checked
    with --dry-launch, which rendezvouses over gloo on the CPU and reports what every rank saw."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dry_launch"] and out["n_gpus"] == 2
    assert sorted(x["rank"] for x in out["ranks"]) == [0, 1] and all(x["world_size"] == 2 for x in out["ranks"])
    assert out["scaling"] == "weak" and all(x["per_gpu_batch"] == 32 for x in out["ranks"]) and out["global_batch"] == 64
    # strong scaling: the global batch is fixed and divided over the ranks
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch", "--scaling", "strong", "--global-batch", "32"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["scaling"] == "strong" and out["global_batch"] == 32 and all(x["per_gpu_batch"] == 16 for x in out["ranks"])
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch", "--scaling", "strong", "--global-batch", "33"],
                       env={**env, "RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"}, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "multiple" in (r.stderr + r.stdout)
    # the launcher parent never loads torch (a process that starts GPU programs makes no GPU call, not even a device count)
    probe = ("import sys, runpy; sys.argv=['bench.py','--gpus','2','--dry-launch'];\n"
             "import subprocess; subprocess.run=lambda *a, **k: type('R', (), {'returncode': 0})()\n"
             "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit: pass\n"
             "print('TORCH_LOADED' if 'torch' in sys.modules else 'TORCH_ABSENT')" % os.path.join(root, "bench.py"))
    r = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=120)
    assert "TORCH_ABSENT" in r.stdout, (r.stdout, r.stderr[-1000:])
    # a rank count that does not match --gpus is refused, never silently run as one rank
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"],
                       env={**env, "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_launch_plan_splits_tensors_beyond_the_descriptor_range():
    """A single GEMM launch addresses < 2 GiB per tensor (bit 31 of a buffer offset = outside).  The reference's own
    256x256 setting (B=32, T=8, base_ch=64: 2^31 bytes in the inc / up0 layers) must be cut into image ranges on
    BatchNorm-group boundaries instead of failing with 'bad argument'."""
    ops = U.ops
    per_img = 256 * 256 * 64 * 2
    n_img, groups = 8 * 32, 8
    assert n_img * per_img == 1 << 31
    ch = ops._img_chunks(n_img, groups, per_img, "t", whole_groups=True)
    assert len(ch) >= 2 and ch[0][0] == 0 and ch[-1][1] == n_img
    for (a, b), (c, _) in zip(ch, ch[1:] + [(n_img, n_img)]):
        assert b == c and (b - a) % (n_img // groups) == 0 and (b - a) * per_img < ops.LAUNCH_BYTES_LIMIT
    assert ops._img_chunks(48, 12, 256 * 256 * 64 * 2, "t", whole_groups=True) == [(0, 48)]        # config 4 (B=4, T=12) fits
    free = ops._img_chunks(n_img, 1, per_img, "t")
    assert all((b - a) * per_img < ops.LAUNCH_BYTES_LIMIT for a, b in free) and free[-1][1] == n_img
    with pytest.raises(U.UclstmError, match="exceeds"):
        ops._img_chunks(2, 2, 1 << 31, "t", whole_groups=True)


def test_panel_cache_is_owned_and_invalidated_by_weight_updates():
    ops = U.ops
    c = ops.PanelCache()
    assert not c.stale() and ops._ACTIVE_CACHE is None
    with c:
        assert ops._ACTIVE_CACHE is c
    assert ops._ACTIVE_CACHE is None
    ops.weights_changed()                                  # what FusedAdamW.step() does
    assert c.stale() and not ops.PanelCache().stale()
    c.clear()
    assert not c.stale()
    assert not hasattr(ops, "_PANEL_CACHE")                # no module-global cache any more


def test_pack_job_table_families_and_segments():
    """Host side of the batched look-ahead packing: uclstm_pack_job_init (a host function) classifies a panel into its kernel
    family and sizes its block range as the single-panel launch would; ops.pack_segments cuts the panels into segments at
    cumulative byte fractions."""
    import ctypes as C
    cases = [(ops.conv_pack_desc(72, 80, [56, 24], [56, 24]), 1, (1 + 1) * 72),               # rows, 9 taps: (chunks0 + chunks1) x N
             (ops.lstm_pack_desc(40, 24), 1, None),
             (ops.convt_dgrad_pack_desc(48, 24), 2, None),                                    # rows, 4 taps
             (ops.conv_dgrad_pack_desc(200, 136, 136), 3, ((200 + 63) // 64) * ((136 + 15) // 16)),   # transposed: 64 K columns x 16 rows
             (ops.lstm_dgrad_pack_desc(40, 24, 40), 3, None),
             (ops.convt_pack_desc(48, 24), 0, None)]                                          # generic
    w = (C.c_float * 4)()
    for d, family, nblocks in cases:
        job = L.PackJob()
        rc = L.lib.uclstm_pack_job_init(C.byref(job), C.byref(d), C.cast(w, C.c_void_p), C.cast(w, C.c_void_p), 5)
        assert rc == family == job.family, (rc, family)
        assert job.block0 == 5 and job.nblocks >= 1 and job.gx >= 1 and job.nblocks % job.gx == 0
        if nblocks is not None:
            assert job.nblocks == nblocks, (job.nblocks, nblocks)
        assert bytes(job.d) == bytes(d) and job.div[2] == d.Ktot
    assert L.lib.uclstm_pack_job_init(None, C.byref(cases[0][0]), C.cast(w, C.c_void_p), C.cast(w, C.c_void_p), 0) == -1
    assert L.lib.uclstm_pack_job_init(C.byref(L.PackJob()), C.byref(cases[0][0]), None, C.cast(w, C.c_void_p), 0) == -1
    sizes = [1, 1, 8, 10, 30, 50]
    assert ops.pack_segments(sizes, (0.02, 0.2, 0.5)) == [0, 0, 1, 1, 2, 3]
    assert ops.pack_segments(sizes, ()) == [0] * 6
    seg = ops.pack_segments([7] * 40, ops.PACK_SEGMENTS)
    assert seg[0] == 0 and seg == sorted(seg) and seg[-1] == len(ops.PACK_SEGMENTS)


def test_store_split_k_long_k_rule_is_off_by_default_and_narrow_when_on(monkeypatch):
    """ops.store_split_k: K ranges of a statistics-free store convolution.  The long-K rule is OFF by default: alone it makes one launch
    of the headline step 23 % faster, in the overlapped step the backward phase does not gain (profiles/round3_store_splitk*.txt).
    Switched on (288 K-steps) it must pick 2 ranges only for the two measured winners and leave every shape that LOST alone."""
    assert ops.SPLITK_STORE_LONGK_STEPS == 0 or "UCLSTM_SPLITK_STORE_LONGK" in os.environ
    monkeypatch.setattr(ops, "SPLITK_STORE_LONGK_STEPS", 0)
    assert ops.store_split_k(640 * 16, 1024, 36864 // 64) == 1
    monkeypatch.setattr(ops, "SPLITK_STORE_LONGK_STEPS", 288)
    assert ops.store_split_k(640 * 16, 1024, 36864 // 64) == 2                       # temporal ConvLSTM input gradient, 64 x 64 seq-20 B=32
    assert ops.store_split_k(48 * 256, 1024, 36864 // 64) == 2                        # the same layer at 256 x 256 seq-12 B=4: 384 tiles, measured
    lost = [(640 * 64, 512, 4608), (640 * 16, 1024, 9216), (640 * 64, 512, 18432), (640 * 16, 1024, 4608), (640 * 16, 512, 9216),
            (640 * 64, 256, 4608), (640 * 64, 512, 2304), (640 * 256, 256, 2304)]
    for pixels, N, K in lost:
        assert ops.store_split_k(pixels, N, K // 64) == 1, (pixels, N, K)
    assert ops.store_split_k(256 * 64, 1024, 36864 // 64) == 1                        # 128 x 128 seq-8 B=32: 512 tiles = two exact rounds
    assert ops.store_split_k(1 * 16 * 16, 256, 2304 // 64) > 1                         # the old small-grid rule (batch-1 inference) still applies
