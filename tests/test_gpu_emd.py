"""approx-EMD on the GPU vs the CPU restatement (parity UNPINNED by the reference: no CPU path,
no test, __expf in the CUDA kernels -> tolerance parity + structural invariants)."""
import numpy as np
import pytest
import torch

from oracle import structural as S
from oracle.gen_golden import chamfer_inputs

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
    return BK


@pytest.mark.parametrize("shape", [(2, 64, 64), (2, 128, 64), (1, 48, 96), (3, 300, 300), (2, 257, 1024)])
def test_approxmatch_matchcost_vs_oracle(shape):
    BK = _gpu()
    B, n, m = shape
    a, b = chamfer_inputs(500 + n + m, B, n, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    match, temp = BK.ApproxMatch(ta, tb)
    cost = BK.MatchCost(ta, tb, match)
    g1, g2 = BK.MatchCostGrad(ta, tb, match)
    torch.cuda.synchronize()
    rmatch, _ = S.approxmatch(a, b)
    rcost = S.matchcost(a, b, rmatch)
    np.testing.assert_allclose(cost.cpu().numpy(), rcost, rtol=1e-4)
    # fast-exp vs expf: compare the matching in aggregate (row/column mass) and elementwise loosely
    gm = match.cpu().numpy()
    np.testing.assert_allclose(gm.sum(1), rmatch.sum(1), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(gm.sum(2), rmatch.sum(2), rtol=1e-3, atol=1e-4)
    # elementwise: the auction's max(0, .)/min(., 1) clamps make isolated entries sensitive to the
    # last bits of exp() (v_exp_f32 here, expf in the oracle, __expf in the reference).  r03 (VERDICT r02 weak #2:
    # "stays loose"), measured by tests/diag/emd_elementwise_probe.py: EVERY entry is within 5e-3 |ref| + 1e-4 at every
    # shape (worst: 3.5e-4 absolute on one 64 x 64 case), and at most 1.5e-3 of them leave 1e-3 |ref| + 1e-5
    err = np.abs(gm - rmatch)
    assert not (err > 5e-3 * np.abs(rmatch) + 1e-4).any(), float(err.max())
    assert (err > 1e-3 * np.abs(rmatch) + 1e-5).mean() < 5e-3
    # cost / grads of the GPU's own matching vs the oracle fed the same matching (isolates those kernels)
    np.testing.assert_allclose(cost.cpu().numpy(), S.matchcost(a, b, gm), rtol=2e-5)
    r1, r2 = S.matchcostgrad(a, b, gm)
    np.testing.assert_allclose(g1.cpu().numpy(), r1, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g2.cpu().numpy(), r2, rtol=1e-4, atol=1e-5)


class _matrix_path:
    """dpf_emd_set_matrix_path(on) for the duration of a block."""
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        from dpf_nets_amd._lib import lib
        self.prev = lib().dpf_emd_set_matrix_path(1 if self.on else 0)

    def __exit__(self, *exc):
        from dpf_nets_amd._lib import lib
        lib().dpf_emd_set_matrix_path(self.prev)


def _rmw(BK, ta, tb):
    BK.EMD_RMW = True
    try:
        return BK.ApproxMatch(ta, tb)
    finally:
        BK.EMD_RMW = False


def test_deferred_materialisation_is_bit_identical_to_rmw():
    """The deferred path's packed-VALU kernels (dpf_emd_set_matrix_path(0)) against the read-modify-write path: same bits."""
    BK = _gpu()
    for (B, n, m) in ((2, 64, 64), (3, 300, 257), (2, 1024, 2048)):
        a, b = chamfer_inputs(700 + n, B, n, m)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        m_rmw, t_rmw = _rmw(BK, ta, tb)
        with _matrix_path(False):
            m_def, t_def = BK.ApproxMatch(ta, tb)
        assert torch.equal(m_rmw, m_def)
        assert torch.equal(t_rmw[:, :n + m], t_def[:, :n + m])        # remainL / remainR


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 48, 96), (3, 300, 257), (2, 1024, 2048), (2, 2048, 2048), (5, 7, 3), (1, 130, 33)])
def test_matrix_core_passes_against_the_difference_form(shape):
    """r05: the level passes and the materialisation on the matrix cores (expanded-form d^2 as one MFMA per 32 x 32 pairs,
    csrc/emd.hip) against the packed-VALU / read-modify-write kernels on the same input: cost within 1e-5 (contract against
    the oracle: 1e-4), every entry of the matching within 2e-3 of its scale 1 (measured worst 9.5e-4 at 2048 x 2048: the
    auction carries a 1e-3 change of the steep levels' weights through), row and column mass within 1e-4, the scratch
    vectors remainL / remainR within 1e-3 -- and NOT the same bits (the matrix path did run)."""
    BK = _gpu()
    B, n, m = shape
    a, b = chamfer_inputs(700 + n, B, n, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    m_rmw, t_rmw = _rmw(BK, ta, tb)
    c_rmw = BK.MatchCost(ta, tb, m_rmw)
    m_mx, t_mx, c_mx = BK.ApproxMatchCost(ta, tb)
    m_mx2, _, c_mx2 = BK.ApproxMatchCost(ta, tb)
    assert torch.equal(m_mx, m_mx2) and torch.equal(c_mx, c_mx2)                   # deterministic
    assert not torch.equal(m_mx, m_rmw)
    np.testing.assert_allclose(c_mx.cpu().numpy(), c_rmw.cpu().numpy(), rtol=1e-5)
    assert float((m_mx - m_rmw).abs().max()) <= 2e-3
    np.testing.assert_allclose(m_mx.sum(1).cpu().numpy(), m_rmw.sum(1).cpu().numpy(), atol=1e-4)
    np.testing.assert_allclose(m_mx.sum(2).cpu().numpy(), m_rmw.sum(2).cpu().numpy(), atol=1e-4)
    np.testing.assert_allclose(t_mx[:, :n + m].cpu().numpy(), t_rmw[:, :n + m].cpu().numpy(), atol=1e-3)
    # the cost the materialisation pass sums is the cost of the matching it wrote
    np.testing.assert_allclose(c_mx.cpu().numpy(), S.matchcost(a, b, m_mx.cpu().numpy()), rtol=2e-5)


def test_matrix_core_passes_repeat_bit_for_bit():
    """Repeated calls on the same clouds: the same bits in every level's ratio vectors (read out of the workspace) and in
    `match`.  r05 history (csrc/emd.hip, pair_exponents): with the literal-zero C operand the sparse-regime pass 2 returned a few
    different sums per million from run to run at 2048^2; the inline-asm MFMA that cured that flickered itself at 8192^2 with two
    or more waves per SIMD (~1 sum per call at B = 2, ~50 at B = 16).  These are the sizes and regimes that showed it."""
    _gpu()
    from dpf_nets_amd._lib import lib, check, current_stream
    L = lib()
    for (B, n, m, reps) in ((2, 2048, 2048, 6), (3, 1500, 900, 6), (2, 8192, 8192, 5), (6, 8192, 8192, 3)):
        a, b = chamfer_inputs(700 + n, B, n, m)
        if n == 8192:
            b = (a[:, ::-1] + 0.03 * b).astype(np.float32).copy()       # the matching of the specified-size test
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        match = torch.empty((B, m, n), device="cuda")
        temp = torch.empty((B, (n + m) * 2), device="cuda")
        nbytes = L.dpf_approxmatch_workspace_bytes(B, n, m)
        ws = torch.zeros((nbytes,), dtype=torch.uint8, device="cuda")
        first = None
        for it in range(reps):
            check(L.dpf_approxmatch_ws(B, n, m, ta.data_ptr(), tb.data_ptr(), match.data_ptr(), temp.data_ptr(), ws.data_ptr(),
                                       nbytes, current_stream()), "approxmatch_ws")
            torch.cuda.synchronize()
            got = (ws[:9 * B * (n + m) * 4].clone(), match.view(torch.int32).clone())
            if first is None:
                first = got
            else:
                assert torch.equal(got[0], first[0]), (B, n, it)
                assert torch.equal(got[1], first[1]), (B, n, it)
        del match, ws, first, got
        torch.cuda.empty_cache()


def emd_fuzz_case(rng, max_n=1500, max_m=1024):
    """One case of the approx-EMD fuzz (the generator of tests/diag/emd_matrix_fuzz.py, r05, plus the kinds VERDICT r05 #1 names):
    random ragged shapes, n != m in both directions, and cloud kinds -- uniform, gauss, jitter (a perturbed re-sampling), clustered,
    offset (a far common origin), line (collinear: near-ties are the rule), plane, dup (many exactly repeated points), grid (a
    lattice: exact ties).  -> (a (B, n, 3), b (B, m, 3), kind)"""
    B = int(rng.integers(1, 4))
    n = int(rng.choice([1, 3, 31, 32, 33, 64, 100, 127, 128, 129, 255, 300, 500, 777, 1024, 1500][:16 if max_n >= 1500 else 13]))
    m = int(rng.choice([1, 2, 32, 33, 63, 96, 128, 130, 257, 400, 512, 900, 1024][:13 if max_m >= 1024 else 10]))
    kind = str(rng.choice(["uniform", "gauss", "jitter", "clustered", "offset", "line", "plane", "dup", "grid"]))
    a = rng.random((B, n, 3), dtype=np.float32) - 0.5
    b = rng.random((B, m, 3), dtype=np.float32) - 0.5
    if kind == "gauss":
        a = (0.2 * rng.standard_normal((B, n, 3))).astype(np.float32)
        b = (0.2 * rng.standard_normal((B, m, 3))).astype(np.float32)
    elif kind == "clustered":
        a = (a * 0.05 + rng.integers(0, 3, (B, n, 1)) * 0.3).astype(np.float32)
    elif kind == "line":
        a[:, :, 1:] = 0
        b[:, :, 1:] = 0
    elif kind == "plane":
        a[:, :, 2] = 0
        b[:, :, 2] = 0.01
    elif kind == "jitter":
        b = (a[:, rng.integers(0, n, m)] + 0.02 * rng.standard_normal((B, m, 3))).astype(np.float32)
    elif kind == "dup":
        a = a[:, rng.integers(0, max(1, n // 8), n)]
        b = (a[:, rng.integers(0, n, m)] + np.float32(0.001) * (rng.random((B, m, 3), dtype=np.float32) - 0.5)).astype(np.float32)
    elif kind == "grid":
        a = (rng.integers(-8, 9, (B, n, 3)) / 16.0).astype(np.float32)
        b = (rng.integers(-8, 9, (B, m, 3)) / 16.0).astype(np.float32)
    elif kind == "offset":
        a, b = a + 5.0, b + 5.0
    return np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32), kind


def collinear_case():
    """tests/diag/emd_collinear_case.py (r05): 4 clouds of 1500 collinear points against 128 -- the one measured input on which the
    r05 matrix-core passes left the contract (1.15e-4 of the oracle's cost)."""
    rng = np.random.default_rng(11)
    for _ in range(67):
        B = int(rng.integers(1, 5))
        n = int(rng.choice([1, 3, 31, 32, 33, 64, 100, 127, 128, 129, 255, 300, 500, 777, 1024, 1500, 2048, 3000]))
        m = int(rng.choice([1, 2, 32, 33, 63, 96, 128, 130, 257, 400, 512, 900, 1024, 2048, 2500]))
        kind = rng.choice(["uniform", "gauss", "jitter", "clustered", "offset", "line"])
        a = rng.random((B, n, 3), dtype=np.float32) - 0.5
        if kind == "gauss": a = (0.2 * rng.standard_normal((B, n, 3))).astype(np.float32)
        if kind == "clustered": a = (a * 0.05 + rng.integers(0, 3, (B, n, 1)) * 0.3).astype(np.float32)
        if kind == "line": a[:, :, 1:] = 0
        if kind == "jitter":
            idx = rng.integers(0, n, m)
            b = (a[:, idx] + 0.02 * rng.standard_normal((B, m, 3))).astype(np.float32)
        else:
            b = (rng.random((B, m, 3), dtype=np.float32) - 0.5) if kind != "gauss" else (0.2 * rng.standard_normal((B, m, 3))).astype(np.float32)
            if kind == "line": b[:, :, 1:] = 0
        if kind == "offset": a, b = a + 5.0, b + 5.0
    assert (B, n, m, kind) == (4, 1500, 128, "line")
    return np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)


def _auction64(a1, b1):
    """approxmatch.cu:3-182 + matchcost in float64 (whole passes as matrix expressions; tests/test_oracle_golden.py holds the C
    oracle to this reading): exact arithmetic for practical purposes -- what the fp32 oracle is compared with to tell how well
    conditioned an input is."""
    n, m = len(a1), len(b1)
    d2 = ((b1[:, None, :].astype(np.float64) - a1[None, :, :].astype(np.float64)) ** 2).sum(2)
    remL = np.full(n, 1.0 if n >= m else float(m // n))
    remR = np.full(m, float(n // m) if n >= m else 1.0)
    match = np.zeros((m, n))
    for j in range(7, -2, -1):
        e = np.exp(-(4.0 ** j) * d2)
        ratioL = remL / (1e-9 + remR @ e)
        sumr = (e @ ratioL) * remR
        ratioR = np.minimum(remR / (sumr + 1e-9), 1.0) * remR
        remR = np.maximum(0.0, remR - sumr)
        w = e * ratioR[:, None] * ratioL[None, :]
        match += w
        remL = np.maximum(0.0, remL - w.sum(0))
    return float((match * np.sqrt(d2)).sum())


def _oracle_bars(BK, a, b, tag):
    """The contract of the module header against the CPU oracle: cost rtol 1e-4, row / column mass 1e-3, finite, repeatable."""
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    match, _, cost = BK.ApproxMatchCost(ta, tb)
    m2, _, c2 = BK.ApproxMatchCost(ta, tb)
    torch.cuda.synchronize()
    assert torch.equal(match, m2) and torch.equal(cost, c2), ("not repeatable",) + tag
    assert torch.isfinite(match).all() and torch.isfinite(cost).all(), ("non-finite",) + tag
    rmatch, _ = S.approxmatch(a, b)
    rcost = S.matchcost(a, b, rmatch)
    gm = match.cpu().numpy()
    got = cost.cpu().numpy()
    err = np.abs(got - rcost) / np.maximum(np.abs(rcost), 1e-6)
    if (err > 1e-4).any():
        # Only where the comparison measures parity: the auction divides by (1e-9 + a sum of weights), and fp32 cannot resolve
        # `remain - consumed` of a size-1 quantity to 1e-9 -- on clouds where some point's neighbours have all been consumed
        # (exact duplicates, near-duplicates) EVERY fp32 evaluation returns its own rounding noise amplified: the fp32 oracle sits
        # up to 4e-4 from the same auction in float64 there and both kernel families up to 5e-4 from the oracle
        # (tools/emd_oracle_fuzz.py, profiles/r06_emd_oracle_fuzz_800.txt: 11 of 1 604 clouds).  The oracle's own distance
        # from exact arithmetic is the measure of that; a cloud further than 1e-5 gets 1e-4 + 4 x that distance.
        for i in np.nonzero(err > 1e-4)[0]:
            cond = abs(float(rcost[i]) - _auction64(a[i], b[i])) / max(abs(float(rcost[i])), 1e-6)
            assert cond > 1e-5 and err[i] <= 1e-4 + 4.0 * cond, (tag, int(i), float(err[i]), cond)
    np.testing.assert_allclose(gm.sum(1), rmatch.sum(1), rtol=1e-3, atol=1e-3, err_msg=repr(tag))
    np.testing.assert_allclose(gm.sum(2), rmatch.sum(2), rtol=1e-3, atol=1e-3, err_msg=repr(tag))
    return float(err.max())


def test_collinear_clouds_vs_oracle():
    """the collinear case against the CPU oracle at the stated bars, on the default (matrix-core) path.  Red with
    the r05 operand format (1.15e-4); the r06 format's exponents are exact where it matters (csrc/emd.hip, section header)."""
    BK = _gpu()
    a, b = collinear_case()
    worst = _oracle_bars(BK, a, b, ("collinear", 4, 1500, 128))
    assert worst <= 2e-5, worst                 # (measured r06: the class of the packed-VALU kernels' 1e-6, not the contract's edge)


def test_emd_fuzz_all_kinds_vs_oracle():
    """the fuzz of tests/diag/emd_matrix_fuzz.py inside the suite and against the CPU ORACLE (not against the
    other GPU kernel family): seeded, time-boxed (the C oracle is one host core: ~27 n m exp per cloud), every cloud kind at
    least twice, both kernel families (the matrix-core default and the packed-VALU kernels) at cost rtol 1e-4, mass 1e-3."""
    import time
    BK = _gpu()
    rng = np.random.default_rng(2026)
    t0 = time.time()
    seen, worst = {}, {}
    for it in range(120):
        a, b, kind = emd_fuzz_case(rng)
        if time.time() - t0 > 60 and all(seen.get(k, 0) >= 2 for k in ("uniform", "gauss", "jitter", "clustered", "offset", "line", "plane", "dup", "grid")):
            break
        seen[kind] = seen.get(kind, 0) + 1
        tag = (it, kind) + a.shape[:2] + b.shape[1:2]
        worst[kind] = max(worst.get(kind, 0.0), _oracle_bars(BK, a, b, tag))
        if it % 4 == 0:
            with _matrix_path(False):
                _oracle_bars(BK, a, b, tag + ("packed VALU",))
    assert len(seen) == 9 and min(seen.values()) >= 2, seen
    print("emd fuzz: cases per kind", seen, "worst cost error per kind", {k: "%.1e" % v for k, v in worst.items()})


def test_matrix_core_exponent_error_bound():
    """The r06 operand format's claim, MEASURED on the device (csrc/emd.hip, section header): dpf_debug_emd_exponents returns the
    exp2 arguments of every pair of a cloud pair exactly as the passes form them; against float64 of the same formula, over the
    pairs whose weight can matter (argument > -150), the error stays below 2e-5 + 2e-7 |argument| at EVERY level on unit-size
    clouds of every kind (the r05 format: 4e-3 at the steepest level; the fp32 difference form of the packed-VALU kernels:
    5e-6 + 1.2e-7 |argument|), and below 2e-4 at the edge of the range the gate admits (|x - c| = 3.3)."""
    _gpu()
    from dpf_nets_amd._lib import lib, check, current_stream
    L = lib()
    rng = np.random.default_rng(5)
    cases = []
    for kind in ("uniform", "line", "jitter", "grid", "tiny", "wide", "offset"):
        n, m = 700, 333
        a = rng.random((n, 3), dtype=np.float32) - 0.5
        b = rng.random((m, 3), dtype=np.float32) - 0.5
        if kind == "line":
            a[:, 1:] = 0; b[:, 1:] = 0
        elif kind == "jitter":
            b = (a[rng.integers(0, n, m)] + 0.003 * rng.standard_normal((m, 3))).astype(np.float32)
        elif kind == "grid":
            a = (rng.integers(-8, 9, (n, 3)) / 16.0).astype(np.float32); b = (rng.integers(-8, 9, (m, 3)) / 16.0).astype(np.float32)
        elif kind == "tiny":
            a, b = a * np.float32(1e-3), b * np.float32(1e-3)
        elif kind == "wide":
            a, b = a * np.float32(3.8), b * np.float32(3.8)      # the cube's corner at log2(e) |x - c|^2 = 15.6 of the gate's 16
        elif kind == "offset":
            a, b = a + np.float32(7.0), b + np.float32(7.0)
        cases.append((kind, np.ascontiguousarray(a), np.ascontiguousarray(b)))
    for kind, a, b in cases:
        n, m = len(a), len(b)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        nbytes = L.dpf_approxmatch_workspace_bytes(1, n, m)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
        out = torch.empty((m, n), device="cuda")
        meta = torch.empty((8,), device="cuda")
        d2 = ((b.astype(np.float64)[:, None, :] - a.astype(np.float64)[None, :, :]) ** 2).sum(2)
        for j in (7, 6, 5, 3, 0, -1):
            check(L.dpf_debug_emd_exponents(n, m, ta.data_ptr(), tb.data_ptr(), j, out.data_ptr(), meta.data_ptr(), ws.data_ptr(),
                                            nbytes, current_stream()), "debug_emd_exponents")
            torch.cuda.synchronize()
            assert float(meta[3]) == 0.0, (kind, "out of range")
            ref = -(4.0 ** j) * 1.4426950408889634 * d2
            got = out.cpu().numpy().astype(np.float64)
            live = ref > -150.0
            err = np.abs(got - ref)[live]
            bar = (2e-4 if kind == "wide" else 2e-5) + 2e-7 * np.abs(ref)[live]
            assert (err <= bar).all(), (kind, j, float(err.max()), float((err / bar).max()), float(meta[4]), float(meta[5]))
            # pairs that cannot matter must not come out alive: an argument below -150 stays below -126 (exp2 -> 0)
            assert (got[~live] < -126.0).all(), (kind, j)


def test_matrix_core_passes_do_not_read_stale_workspace():
    """every product caller hands the kernels a torch.empty workspace while r05's repeat test handed them
    zeros.  Here the SAME call runs on a zero-filled workspace, on one filled with 0xFF bytes (fp16 / fp32 NaN patterns, index -1),
    on one filled with 0x7B (large finite numbers, huge indices) and on recycled memory (a freed `match` of another call):
    identical bits in `match`, the scratch vectors and every level's ratio vectors."""
    _gpu()
    from dpf_nets_amd._lib import lib, check, current_stream
    L = lib()
    for (B, n, m) in ((2, 2048, 2048), (3, 1500, 900), (2, 300, 1000), (1, 4096, 4096)):
        a, b = chamfer_inputs(800 + n, B, n, m)
        if n == m:
            b = (a[:, ::-1] + 0.03 * b).astype(np.float32).copy()
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        nbytes = L.dpf_approxmatch_workspace_bytes(B, n, m)
        first = None
        for fill in ("zeros", 0xFF, 0x7B, "recycled"):
            if fill == "recycled":
                junk = torch.full((B, m, n), float("nan"), device="cuda")
                junk2 = torch.full((nbytes // 4 + 16,), -3.0e38, device="cuda")
                del junk, junk2                                   # the caching allocator hands these blocks out again below
                ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
                match = torch.empty((B, m, n), device="cuda")
                temp = torch.empty((B, (n + m) * 2), device="cuda")
            else:
                ws = torch.zeros((nbytes,), dtype=torch.uint8, device="cuda") if fill == "zeros" else \
                    torch.full((nbytes,), fill, dtype=torch.uint8, device="cuda")
                match = torch.full((B, m, n), float("nan"), device="cuda")
                temp = torch.full((B, (n + m) * 2), float("nan"), device="cuda")
            cost = torch.empty((B,), device="cuda")
            check(L.dpf_approxmatch_cost_ws(B, n, m, ta.data_ptr(), tb.data_ptr(), match.data_ptr(), temp.data_ptr(), cost.data_ptr(),
                                            ws.data_ptr(), nbytes, current_stream()), "approxmatch_cost_ws")
            torch.cuda.synchronize()
            got = (ws[:9 * B * (n + m) * 4].clone(), match.view(torch.int32).clone(), temp[:, :n + m].view(torch.int32).clone(),
                   cost.view(torch.int32).clone())
            assert torch.isfinite(match).all() and torch.isfinite(cost).all(), (B, n, m, fill)
            if first is None:
                first = got
            else:
                for x, y, what in zip(got, first, ("ratio vectors", "match", "remainL / remainR", "cost")):
                    assert torch.equal(x, y), (B, n, m, fill, what)
            del ws, match, temp, got
        del first
        torch.cuda.empty_cache()


@pytest.mark.parametrize("kind", ["scale", "offset", "inf", "nan"])
def test_matrix_core_passes_leave_out_of_range_clouds_to_the_valu_kernels(kind):
    """The matrix-core passes take clouds with log2(e) |x - c|^2 <= 16 for every point (c = cloud 1's centroid; beyond that the
    mixed terms' rounding would leave the 1e-4 class); a call with any point beyond that, or not finite, is decided ON THE DEVICE
    for the packed-VALU kernels: the read-modify-write path's bits.  A far common offset is inside."""
    BK = _gpu()
    B, n, m = 2, 300, 257
    a, b = chamfer_inputs(77, B, n, m)
    if kind == "scale":
        a, b = a * 60.0, b * 60.0
    elif kind == "offset":
        a, b = a + 1000.0, b + 1000.0          # centred away: stays on the matrix cores
    elif kind == "inf":
        b[1, 5, 2] = np.inf
    else:
        a[0, 7, 0] = np.nan
    a, b = a.astype(np.float32), b.astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    if kind in ("inf", "nan"):
        # (non-finite input: the packed-VALU deferred kernels are the yardstick -- the read-modify-write path adds `match` up
        # level by level in memory, the deferred one in registers: the same values, but not the same NaN payloads / signs of zero)
        with _matrix_path(False):
            m_rmw, t_rmw = BK.ApproxMatch(ta, tb)
    else:
        m_rmw, t_rmw = _rmw(BK, ta, tb)
    m_def, t_def = BK.ApproxMatch(ta, tb)
    if kind == "offset":
        assert not torch.equal(m_rmw, m_def)
        assert float((m_def - m_rmw).abs().max()) <= 2e-3
    else:
        assert torch.equal(m_rmw.view(torch.int32), m_def.view(torch.int32))          # (bit patterns: NaN == NaN)
        assert torch.equal(t_rmw[:, :n + m].view(torch.int32), t_def[:, :n + m].view(torch.int32))


def test_emd_full_size_invariants_and_autograd():
    BK = _gpu()
    from dpf_nets_amd.networks.utils import emd_approx
    B, n = 8, 2048
    a, b = chamfer_inputs(91, B, n, n)
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tb = torch.from_numpy(b).cuda()
    match, _ = BK.ApproxMatch(ta.detach(), tb)
    assert (match >= 0).all()
    assert (match.sum(1) <= 1 + 1e-3).all() and (match.sum(2) <= 1 + 1e-3).all()
    assert match.sum() > 0.95 * n * B
    emd = emd_approx(ta, tb)
    assert emd.shape == (B,) and (emd > 0).all()
    emd.sum().backward()
    assert torch.isfinite(ta.grad).all() and ta.grad.abs().sum() > 0
    same = emd_approx(tb, tb.clone())
    assert (same < 0.1 * emd.detach()).all()                   # a cloud matched with itself costs ~nothing


def test_one_pass_gradients_match_two_pass():
    """dpf_matchcostgrad_ws (match read once) vs dpf_matchcostgrad (two kernels): same sums in a different order."""
    BK = _gpu()
    for (B, n, m) in ((2, 64, 64), (3, 300, 257), (128, 1024, 500), (70, 2048, 700), (600, 200, 129), (1100, 200, 90)):
        a, b = chamfer_inputs(900 + n, B, n, m)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        match, _ = BK.ApproxMatch(ta, tb)
        g1, g2 = BK.MatchCostGrad(ta, tb, match)
        assert torch.equal(g1, BK.MatchCostGrad(ta, tb, match)[0])          # deterministic
        BK.EMD_GRAD_TWO_PASS = True
        try:
            h1, h2 = BK.MatchCostGrad(ta, tb, match)
        finally:
            BK.EMD_GRAD_TWO_PASS = False
        np.testing.assert_allclose(g1.cpu().numpy(), h1.cpu().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(g2.cpu().numpy(), h2.cpu().numpy(), rtol=1e-4, atol=1e-5)
        r1, r2 = S.matchcostgrad(a, b, match.cpu().numpy())
        np.testing.assert_allclose(g1.cpu().numpy(), r1, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(g2.cpu().numpy(), r2, rtol=1e-4, atol=1e-5)


def test_fused_cost_matches_separate_matchcost():
    """dpf_approxmatch_cost_ws: the same matching bit for bit, the cost within fp32 summation order of dpf_matchcost
    (and of the oracle fed that matching), deterministic from run to run; match_cost() uses it."""
    BK = _gpu()
    from dpf_nets_amd.metrics.StructuralLosses.match_cost import match_cost
    for (B, n, m) in ((2, 64, 64), (3, 300, 257), (2, 1024, 2048), (1, 100, 37)):
        a, b = chamfer_inputs(900 + n, B, n, m)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        match, temp = BK.ApproxMatch(ta, tb)
        cost = BK.MatchCost(ta, tb, match)
        m2, t2, c2 = BK.ApproxMatchCost(ta, tb)
        m3, t3, c3 = BK.ApproxMatchCost(ta, tb)
        assert torch.equal(match, m2) and torch.equal(temp[:, :n + m], t2[:, :n + m])
        assert torch.equal(c2, c3) and torch.equal(m2, m3)
        np.testing.assert_allclose(c2.cpu().numpy(), cost.cpu().numpy(), rtol=2e-5)
        np.testing.assert_allclose(c2.cpu().numpy(), S.matchcost(a, b, match.cpu().numpy()), rtol=2e-5)
        if n == m:                                                    # emd_approx asserts N == M
            np.testing.assert_allclose(match_cost(ta, tb).cpu().numpy(), c2.cpu().numpy(), rtol=0, atol=0)


def test_pairwise_emd_cd_and_emd_cd_callers():
    """metrics/evaluation_metrics.py (_pairwise_EMD_CD_, EMD_CD: lib/metrics/evaluation_metrics.py:48-121) against the
    reference-shaped loops over nn_distance / match_cost (expand + contiguous per row block) and the oracle's Chamfer."""
    BK = _gpu()
    from dpf_nets_amd.metrics import evaluation_metrics as EM
    from dpf_nets_amd.metrics.StructuralLosses.nn_distance import nn_distance
    from dpf_nets_amd.metrics.StructuralLosses.match_cost import match_cost
    a, b = chamfer_inputs(77, 5, 96, 96)
    b = np.concatenate([b, a[:2] + 0.01]).astype(np.float32)               # 7 references
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    cd, emd = EM._pairwise_EMD_CD_(ta, tb, batch_size=3)
    assert cd.shape == emd.shape == (5, 7)
    for i in range(5):
        cds, emds = [], []
        for r0 in range(0, 7, 3):
            ref = tb[r0:r0 + 3]
            exp = ta[i].view(1, -1, 3).expand(ref.shape[0], -1, -1).contiguous()
            dl, dr = nn_distance(exp, ref)
            cds.append(dl.mean(1) + dr.mean(1))
            emds.append(match_cost(exp, ref) / 96.0)
        np.testing.assert_allclose(cd[i].cpu().numpy(), torch.cat(cds).cpu().numpy(), rtol=1e-6, atol=1e-9)
        assert torch.equal(emd[i], torch.cat(emds))
    d1, _, d2, _ = S.nndistance(np.repeat(a[3:4], 7, 0), b)
    np.testing.assert_allclose(cd[3].cpu().numpy(), d1.mean(1) + d2.mean(1), rtol=1e-5)
    res = EM.EMD_CD(ta, tb[:5], batch_size=2, reduced=False)["MMD-CD"]
    dl, dr = nn_distance(ta, tb[:5].contiguous())
    np.testing.assert_allclose(res.cpu().numpy(), (dl.mean(1) + dr.mean(1)).cpu().numpy(), rtol=1e-6)
    assert abs(float(EM.EMD_CD(ta, tb[:5], batch_size=4)["MMD-CD"]) - float(res.mean())) < 1e-7
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        EM._pairwise_EMD_CD_(ta.cpu(), tb.cpu(), 3)


def test_emd_at_the_specified_size_vs_oracle_and_properties():
    """BASELINE.json configs[4]: N = M = 8192.  One cloud against the C oracle (approxmatch.cu:3-224 restated; ~20 s of one
    host core) -- cost, row/column mass and the elementwise fraction as above -- then B = 2 through the properties the
    domain offers (mass conservation, nonnegativity, symmetry of the cost under swapping the clouds within the auction's
    tolerance, match_cost() == MatchCost(ApproxMatch()), packed-VALU deferred == read-modify-write bits, matrix-core passes
    within 2e-3 of them elementwise)."""
    BK = _gpu()
    n = 8192
    a, b = chamfer_inputs(4242, 2, n, n)
    b = (a[:, ::-1] + 0.03 * b).astype(np.float32).copy()              # a jittered, re-ordered copy: a non-trivial matching
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    match, temp, cost = BK.ApproxMatchCost(ta, tb)
    torch.cuda.synchronize()
    rmatch, _ = S.approxmatch(a[:1], b[:1])
    rcost = S.matchcost(a[:1], b[:1], rmatch)
    np.testing.assert_allclose(cost[:1].cpu().numpy(), rcost, rtol=1e-4)
    gm = match[:1].cpu().numpy()
    np.testing.assert_allclose(gm.sum(1), rmatch.sum(1), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(gm.sum(2), rmatch.sum(2), rtol=1e-3, atol=1e-4)
    err = np.abs(gm - rmatch)
    assert (err > 5e-3 * np.abs(rmatch) + 1e-4).mean() < 1e-6, float(err.max())     # (67 M entries: a handful at most)
    assert (err > 1e-3 * np.abs(rmatch) + 1e-5).mean() < 1e-4
    del gm, rmatch, err
    # properties at B = 2
    assert (match >= 0).all()
    assert (match.sum(1) <= 1 + 1e-3).all() and (match.sum(2) <= 1 + 1e-3).all() and match.sum() > 0.95 * 2 * n
    assert torch.allclose(BK.MatchCost(ta, tb, match), cost, rtol=1e-5)
    m_rmw, _ = _rmw(BK, ta, tb)
    assert float((m_rmw - match).abs().max()) <= 2e-3                  # the matrix-core passes against the difference form
    with _matrix_path(False):
        m_valu, _, c_valu = BK.ApproxMatchCost(ta, tb)
    assert torch.equal(m_rmw, m_valu)
    assert torch.allclose(c_valu, cost, rtol=1e-5)
    del m_rmw, m_valu
    _, _, cost_swapped = BK.ApproxMatchCost(tb, ta)
    assert torch.allclose(cost_swapped, cost, rtol=5e-2)               # the auction is not symmetric, its optimum nearly is
