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


def mr_rel_l2_per_t(got, ref):
    """Per-timestep rel-L2 of MEAN-REMOVED frames ([B,T,C,H,W]: the per-frame spatial mean is subtracted from both sides).
    At random init the eval-mode output of the model is a DC offset (the outc bias) 60-230x larger than the spatial signal;
    the raw rel-L2 is blind to a wrong signal there, this one is an error relative to the signal."""
    g = got.double() - got.double().mean(dim=(-2, -1), keepdim=True)
    r = ref.double() - ref.double().mean(dim=(-2, -1), keepdim=True)
    return [float((g[:, t] - r[:, t]).norm() / (r[:, t].norm() + 1e-30)) for t in range(ref.shape[1])]


EVAL_MR_CAP = 5e-2          # stated: eval-mode error <= 5 % of the spatial signal wherever the reference's own anchor is looser


def eval_parity(name, model, g, sd, forward, pre="ac_", raw_tol=1e-2, cap=EVAL_MR_CAP):
    """Eval-mode forward of ``model`` (``forward()`` -> [B,T,1,H,W] CPU tensor) against a seeded reference fixture, at random
    init and with the reference's warmed running statistics loaded: raw per-timestep rel-L2 <= raw_tol and mean-removed
    <= min(1.25 x the reference's own autocast eval drift (``pre``: ac_ = bf16, ac16_ = fp16), cap).  Restores ``sd``."""
    worst = 0.0
    for variant in ("", "_warm"):
        if variant == "_warm":
            model.load_state_dict({**sd, **{k[len("warm/"):]: v for k, v in g.items() if k.startswith("warm/")}})
        got, ref = forward(), g["out_eval" + variant]
        # raw line: the stated tolerance, or the reference's own raw autocast drift where that is larger (seed 950: the output
        # is all signal, |out| = 0.004, and the reference's bf16 autocast itself sits at 1.12e-2)
        tol = max(raw_tol, 1.25 * max(float(v) for v in g[pre + "eval_rel_l2_per_t" + variant]))
        e_raw = [rel_l2(got[:, t], ref[:, t]) for t in range(ref.shape[1])]
        e_mr = mr_rel_l2_per_t(got, ref)
        anchor = [float(v) for v in g[pre + "eval_mr_rel_l2_per_t" + variant]]
        print(f"[parity] {name}{variant}: eval forward vs reference per-timestep rel-L2 raw {[round(e, 6) for e in e_raw]} (tol {raw_tol}), "
              f"mean-removed {[round(e, 5) for e in e_mr]}; the reference's own {pre}autocast eval drift, mean-removed "
              f"{[round(e, 5) for e in anchor]} (bound min(1.25 x, {cap}))")
        assert max(e_raw) <= tol, f"{name}{variant}: raw eval rel-L2 {max(e_raw):.3e} > {tol:.3e}"
        for t, (e, r) in enumerate(zip(e_mr, anchor)):
            assert e <= min(1.25 * r, cap), f"{name}{variant} t={t}: eval drift {e:.4f} of the signal > min(1.25 x {r:.4f}, {cap})"
        worst = max(worst, max(e_mr))
    model.load_state_dict(sd)
    return worst


def per_tensor_grad_report(names, got, want, ref_norms, anchor_rel, floor=2e-2, tiny=1e-6, factor=2.5, median_factor=1.25):
    """Per-parameter gradient parity against the reference's OWN per-tensor autocast drift (``anchor_rel``):

      * every tensor: rel-L2(got_i, want_i) <= max(factor x anchor_rel[i], floor), and its norm within that band of the
        reference's norm;
      * the MEDIAN over tensors of rel-L2_i / anchor_rel[i] <= median_factor -- the population of tensors drifts like the
        reference's own reduced-precision run does;
      * tensors whose reference gradient is analytically zero (conv bias in front of BatchNorm: the reference holds ~1e-9
        noise there) must be ~zero.

    Why 2.5 per tensor where the whole-vector and the median checks use 1.25: one tensor's drift is ONE draw of the chaotic
    amplification of bf16 roundings through batch-statistics BatchNorm at random init (the reference's own per-tensor autocast
    drift spans 0.02 .. 0.57 inside one model), and the small bias sums are cancellation-dominated; two correct bf16
    implementations differ per tensor by far more than in the aggregate.  Measured on MI355X over 88 tensors x 5 cases
    (gpurun_out/r3_t3.log): median ratio 0.989 / 0.993 / 0.998 / 1.006 / 1.109, worst single tensor 2.09 (up0.up.bias at
    256 x 256: rel-L2 0.104 against the reference's 0.050 -- the ConvTranspose bias gradients are sums over every output pixel
    of a gradient the following BatchNorm has nearly centred), next 1.64 and 1.50.
    A wrong gradient (sign, scale, missing term, wrong tensor) is O(1) and fails both lines.
    Returns (failures, rows sorted by slack, median ratio)."""
    total = float(np.sqrt(sum(float(n) ** 2 for n in ref_norms)))
    rows, bad, ratios = [], [], []
    for i, k in enumerate(names):
        g, w = got[k].double().flatten(), want[k].double().flatten()
        rn = float(ref_norms[i])
        if rn <= tiny * total:
            gn = float(g.norm())
            rows.append((0.0, k, gn, rn, 0.0, "zero"))
            if gn > 10 * tiny * total:
                bad.append(f"{k}: reference gradient is ~0 ({rn:.2e}) but got norm {gn:.2e}")
            continue
        rel = float((g - w).norm() / (w.norm() + 1e-30))
        bound = max(factor * float(anchor_rel[i]), floor)
        dn = abs(float(g.norm()) - rn) / rn
        rows.append((rel / bound, k, rel, bound, dn, ""))
        if float(anchor_rel[i]) > 1e-4:
            ratios.append(rel / float(anchor_rel[i]))
        if rel > bound:
            bad.append(f"{k}: rel-L2 {rel:.4f} > {bound:.4f} ({factor} x reference autocast {float(anchor_rel[i]):.4f})")
        if dn > bound:
            bad.append(f"{k}: norm off by {dn:.4f} > {bound:.4f}")
    rows.sort(reverse=True)
    med = float(np.median(ratios)) if ratios else 0.0
    if med > median_factor:
        bad.append(f"median over {len(ratios)} tensors of rel-L2 / reference autocast drift = {med:.3f} > {median_factor}")
    return bad, rows, med


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
