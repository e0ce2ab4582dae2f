"""Helpers shared by the parity tests: seeded sampling of big tensors against the committed fixtures."""
import os

import numpy as np
import torch

from transception_amd.seeded_init import _stream

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NSAMP = 2048


def sample_idx(tag: str, numel: int, n: int = NSAMP) -> np.ndarray:
    g = _stream(f"sample:{tag}", 3)
    return g.integers(0, numel, size=min(n, numel), dtype=np.int64)


def load(name: str):
    return np.load(os.path.join(GOLDEN, name))


def check_packed(store, tag: str, t: torch.Tensor, atol: float, rtol: float = 0.0, sum_rtol: float = 1e-4,
                 scale_rel: float = 0.0):
    """Compare tensor `t` with the (shape, samples, checksums) triple stored under `tag`."""
    a = t.detach().contiguous().float().cpu().reshape(-1).numpy()
    shape = tuple(int(s) for s in store[tag + "/shape"])
    assert tuple(t.shape) == shape, f"{tag}: shape {tuple(t.shape)} != golden {shape}"
    idx = sample_idx(tag, a.size)
    want = store[tag + "/samples"]
    got = a[idx]
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want) + scale_rel * float(np.abs(want).max())
    assert np.all(err <= tol), f"{tag}: max err {err.max():.3e} (tol {atol:g}+{rtol:g}*|x|), worst at {int(err.argmax())}"
    s, sa = store[tag + "/sum"]
    gsa = np.abs(a.astype(np.float64)).sum()
    assert abs(gsa - sa) <= sum_rtol * max(sa, 1e-12) + atol * a.size * 0.01, f"{tag}: |x| checksum {gsa} vs {sa}"
    return float(err.max())


def check_packed_l2(store, tag: str, t: torch.Tensor, rel_l2: float, floor: float):
    """16-bit paths: the sampled entries of `t` against the stored samples in relative L2 (rounding noise of a long bf16 sum is a few
    per cent of the tensor's norm and lands anywhere, so a per-entry bound relative to the largest sample is the wrong shape);
    tensors whose samples have a norm below `floor` are held to an absolute L2 of floor * rel_l2 ... floor instead."""
    a = t.detach().contiguous().float().cpu().reshape(-1).numpy()
    shape = tuple(int(s) for s in store[tag + "/shape"])
    assert tuple(t.shape) == shape, f"{tag}: shape {tuple(t.shape)} != golden {shape}"
    want = store[tag + "/samples"].astype(np.float64)
    got = a[sample_idx(tag, a.size)].astype(np.float64)
    d, nw = float(np.linalg.norm(got - want)), float(np.linalg.norm(want))
    assert d <= rel_l2 * max(nw, floor), f"{tag}: sampled rel L2 {d / max(nw, 1e-300):.3e} (|want| {nw:.3e}, floor {floor:.3e}, budget {rel_l2})"
    return d / max(nw, floor)


def check_all_grads(named, ref_grads, atol: float = 2e-6, rtol: float = 2e-3, what: str = ""):
    """EVERY parameter gradient of the HIP model against a reference gradient (the CPU oracle's, evaluated in the calling test):
    per tensor |got - ref|_inf <= atol + rtol |ref|_inf.  `named`: dict(model.named_parameters()); `ref_grads`: name -> tensor or
    None.  A parameter without a HIP gradient must have none (or an all-zero one) in the reference.  Returns the number of live
    tensors checked and the worst (error / bound, name); fails listing every tensor out of bound."""
    bad, worst, n = [], (0.0, ""), 0
    for key, p in named.items():
        ref = ref_grads.get(key)
        if p.grad is None:
            if ref is not None and float(ref.abs().max()) != 0.0:
                bad.append(f"{key}: no HIP gradient, reference max {float(ref.abs().max()):.3e}")
            continue
        if ref is None:
            if float(p.grad.abs().max()) != 0.0:
                bad.append(f"{key}: HIP gradient max {float(p.grad.abs().max()):.3e}, none in the reference")
            continue
        n += 1
        got = p.grad.detach().float().cpu()
        ref = ref.detach().float().cpu().reshape(got.shape)
        err, bound = float((got - ref).abs().max()), atol + rtol * float(ref.abs().max())
        if not (err <= bound):
            bad.append(f"{key}: err {err:.3e} > {bound:.3e} (|ref|max {float(ref.abs().max()):.3e})")
        if err / bound > worst[0]:
            worst = (err / bound, key)
    assert not bad, f"{what}{len(bad)} of {n} gradient tensors out of bound:\n  " + "\n  ".join(bad[:20])
    return n, worst


def family(key: str) -> str:
    """Parameter family of a gradient tensor: its name with the indices of repeated same-shape modules (MB paths, MHCA layers, RIPM steps,
    stage-1 / decoder transformer blocks, bridge layers) replaced by '#'; indices that change the shape (stage, window, MixFFN scale,
    decoder) stay.  The 1217 live tensors fall into ~330 families."""
    import re
    return re.sub(r"(mhca_blks|MHCA_layers|patch_embeds|block1)\.\d+|(layer_former_|bridge_layer)\d+", lambda m: (m.group(1) + ".#") if m.group(1) else (m.group(2) + "#"), key)


BUDGET_FILE = os.path.join(GOLDEN, "lowp_grad_budget.json")
BUDGET_MARGIN, BUDGET_FLOOR = 1.5, 0.015         # a family's bound = 1.5 x its measured worst relative L2, at least 0.015


def load_budget(config: str):
    """Per-family bounds of the 16-bit gradient check for one test configuration (tests/golden/lowp_grad_budget.json, measured on MI355X by
    running the same tests with TC_WRITE_BUDGET=<json path>): family -> (rel L2 bound, cosine bound)."""
    import json
    if not os.path.exists(BUDGET_FILE):
        return None
    tab = json.load(open(BUDGET_FILE)).get(config)
    if tab is None:
        return None
    # (the cosine deficit 1 - cos goes with the SQUARE of the relative error -- 1 - cos ~ rel^2 / 2 for an error orthogonal to the gradient --
    # so the margin on it is 1.5^2: a plain 1.5 made the cosine the tighter of the two bounds, at 1.22 x the measured relative error)
    return {fam: (max(BUDGET_MARGIN * v[0], BUDGET_FLOOR), 1.0 - max(BUDGET_MARGIN ** 2 * (1.0 - v[1]), 0.5 * BUDGET_FLOOR ** 2)) for fam, v in tab.items()}


def check_all_grads_lowp(named_lp, named_ref, rel_l2: float, cos_min: float, what: str = "", zero_rel: float = 1e-6, zero_abs: float = 1e-5,
                         config: str = ""):
    """16-bit storage path against the fp32 HIP path, EVERY gradient tensor: relative L2 error and cosine within the bound of the tensor's
    FAMILY (load_budget(config): 1.5 x the family's measured worst value; a family without an entry, or config == "", falls back to
    the global rel_l2 / cos_min).  Tensors whose fp32 gradient is numerically zero (norm below zero_rel x the global gradient norm:
    biases in front of a softmax over their own axis or of a BatchNorm, whose exact gradient is 0) are held to |diff| <= zero_abs x
    the global norm instead.  With TC_WRITE_BUDGET=<path> the measured per-family worst values of this call are merged into that
    json file under `config` (the generator of tests/golden/lowp_grad_budget.json)."""
    gn = float(sum(float((q.grad.double() ** 2).sum()) for q in named_ref.values() if q.grad is not None) ** 0.5)
    budget = load_budget(config) if config else None
    measured = {}
    bad, worst, n, nz = [], (0.0, ""), 0, 0
    for key, p in named_lp.items():
        q = named_ref[key]
        assert (p.grad is None) == (q.grad is None), key
        if p.grad is None:
            continue
        n += 1
        a, b = p.grad.detach().double().flatten(), q.grad.detach().double().flatten()
        nb = float(b.norm())
        d = float((a - b).norm())
        if nb < zero_rel * gn:
            nz += 1
            if d > zero_abs * gn:
                bad.append(f"{key}: |ref| {nb:.2e} (numerically zero), |diff| {d:.2e} > {zero_abs * gn:.2e}")
            continue
        rel = d / nb
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
        fam = family(key)
        m = measured.setdefault(fam, [0.0, 1.0, 0])
        m[0], m[1], m[2] = max(m[0], rel), min(m[1], cos), m[2] + 1
        b_rel, b_cos = budget.get(fam, (rel_l2, cos_min)) if budget is not None else (rel_l2, cos_min)
        b_rel, b_cos = min(b_rel, rel_l2), max(b_cos, cos_min)           # (never looser than the global bound)
        if not (rel <= b_rel and cos >= b_cos):
            bad.append(f"{key}: rel L2 {rel:.3e} (bound {b_rel:.3e}), cosine {cos:.5f} (bound {b_cos:.5f})")
        if rel > worst[0]:
            worst = (rel, key)
    wb = os.environ.get("TC_WRITE_BUDGET")
    if wb and config:
        import json
        tab = json.load(open(wb)) if os.path.exists(wb) else {}
        old = tab.get(config, {})
        for fam, v in measured.items():
            o = old.get(fam)
            old[fam] = [round(max(v[0], o[0]) if o else v[0], 5), round(min(v[1], o[1]) if o else v[1], 6), v[2]]
        tab[config] = old
        with open(wb, "w") as f:
            json.dump(tab, f, indent=0, sort_keys=True)
        #  write mode is not a free pass: the measured errors it records must still sit inside the global bounds
        over = [f"{fam}: rel L2 {v[0]:.3e} / cosine {v[1]:.5f}" for fam, v in measured.items() if not (v[0] <= rel_l2 and v[1] >= cos_min)]
        assert not over, f"{what}TC_WRITE_BUDGET run: families outside the global bound ({rel_l2}, {cos_min}):\n  " + "\n  ".join(over[:20])
        return n, worst
    assert not bad, f"{what}{len(bad)} of {n} gradient tensors out of bound:\n  " + "\n  ".join(bad[:20])
    return n, worst


def gelu_model(u: torch.Tensor, dtype) -> torch.Tensor:
    """GELU as the kernels of storage type `dtype` evaluate it, in the precision of `u`: exact erf for fp32; for bf16 / fp16 the forward kernels
    take Phi(x) ~ 0.5 + t P(t^2), t = clamp(x, -X0, X0) (csrc/tc_common.h TC_PHI_*: within 1.02e-5 of the normal CDF, pinned by
    tests/test_ops_gpu.py::test_gelu_16bit_polynomial_against_erf) -- the fp64 rounding models use the same polynomial, read from the header."""
    if dtype == torch.float32:
        return torch.nn.functional.gelu(u)
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "transception_amd", "csrc", "tc_common.h")).read()
    val = lambda name: float(re.search(r"#define %s\s+(-?[0-9.eE+-]+)f" % name, hdr).group(1))
    x0, c = val("TC_PHI_X0"), [val("TC_PHI_C%d" % k) for k in range(9)]
    t = u.clamp(-x0, x0)
    s2 = t * t
    pl = torch.full_like(u, c[8])
    for k in range(7, -1, -1):
        pl = pl * s2 + c[k]
    return u * (t * pl + 0.5)
