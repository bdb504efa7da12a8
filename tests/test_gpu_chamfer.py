"""Chamfer parity on the GPU: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import detrng
from oracle import structural as S
from oracle.gen_golden import chamfer_inputs

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
    return BK


# "small" = the LDS-staged one-query-per-lane kernel of rank-sized batches, forced for every shape whose clouds fit in LDS
IMPLS = ["auto", "mfma", "brute", "small"]


def _run(BK, a, b, impl=None):
    from dpf_nets_amd._lib import lib
    old = BK.NN_IMPL
    BK.NN_IMPL = "brute" if impl == "small" else (impl or old)
    old_small = lib().dpf_nn_small_mode(1 if impl == "small" else -1)
    try:
        d1, i1, d2, i2 = BK.NNDistance(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    finally:
        BK.NN_IMPL = old
        lib().dpf_nn_small_mode(old_small)
    torch.cuda.synchronize()
    return d1.cpu().numpy(), i1.cpu().numpy(), d2.cpu().numpy(), i2.cpu().numpy()


def _assert_bit_exact(got, ref, tag):
    for g, r, name in zip(got, ref, ("dist1", "idx1", "dist2", "idx2")):
        assert g.dtype == r.dtype and g.shape == r.shape, (tag, name)
        if g.dtype == np.float32:
            assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), \
                (tag, name, float(np.abs(g - r).max()), int((g != r).sum()))
        else:
            assert np.array_equal(g, r), (tag, name, int((g != r).sum()))


# shapes chosen to hit every launch variant (candidate split KS = 4 / 2 / 1), ragged tails
# (n, m not multiples of 8 / 64 / 128 / 512) and degenerate sizes
SHAPES = [(3, 257, 257), (2, 130, 515), (1, 1, 1), (2, 7, 3), (1, 64, 8), (5, 128, 1000), (32, 1024, 1024),
          (32, 2048, 2048), (48, 2500, 2048), (64, 2048, 2048)]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("shape", SHAPES)
def test_nndistance_bit_exact_vs_oracle(shape, impl):
    BK = _gpu()
    B, n, m = shape
    a, b = chamfer_inputs(1000 + n + m, B, n, m)
    _assert_bit_exact(_run(BK, a, b, impl), S.nndistance(a, b), shape)


@pytest.mark.parametrize("impl", ["mfma", "brute", "small"])
@pytest.mark.parametrize("kind", ["same_x", "two_planes", "line", "clustered", "surface", "far_offset", "all_equal",
                                  "big_coords"])
def test_pruned_search_adversarial_distributions(kind, impl):
    """Distributions that stress the matrix-core filter's tolerance window and tie queue: identical x, duplicated
    planes (massive exact ties), clusters, a thin surface, clouds far from the origin, all candidates equal."""
    BK = _gpu()
    B, n, m = 3, 700, 1100
    a, b = chamfer_inputs(2000 + len(kind), B, n, m)
    if kind == "same_x":
        a[..., 0] = 0.125; b[..., 0] = 0.125
    elif kind == "two_planes":
        a[..., 0] = np.where(a[..., 0] > 0, 0.25, -0.25); b[..., 0] = np.where(b[..., 0] > 0, 0.25, -0.25)
        b[:, 500:600] = b[:, 100:200]
    elif kind == "line":
        a[..., 1:] = 0; b[..., 1:] = 0
    elif kind == "clustered":
        a = (np.round(a * 4) / 4 + a * 0.01).astype(np.float32); b = (np.round(b * 4) / 4 + b * 0.01).astype(np.float32)
    elif kind == "surface":
        a[..., 2] = np.sin(a[..., 0] * 9) * 0.1; b[..., 2] = np.sin(b[..., 0] * 9) * 0.1
    elif kind == "far_offset":          # clouds far from the origin: |p|^2 >> d, the filter's window is wide
        a += np.float32(37.0); b += np.float32(37.0)
    elif kind == "all_equal":           # every candidate ties: queue overflow -> exact rescan path
        b[:] = b[:, :1]
    elif kind == "big_coords":
        a *= np.float32(1000.0); b *= np.float32(1000.0)
    _assert_bit_exact(_run(BK, a, b, impl), S.nndistance(a, b), kind)


def _adversarial(kind, a, b):
    """the distributions of test_pruned_search_adversarial_distributions, in place on copies"""
    a, b = a.copy(), b.copy()
    m = b.shape[1]
    if kind == "same_x":
        a[..., 0] = 0.125; b[..., 0] = 0.125
    elif kind == "two_planes":
        a[..., 0] = np.where(a[..., 0] > 0, 0.25, -0.25); b[..., 0] = np.where(b[..., 0] > 0, 0.25, -0.25)
        b[:, m // 2:m // 2 + 100] = b[:, 100:200]
    elif kind == "line":
        a[..., 1:] = 0; b[..., 1:] = 0
    elif kind == "clustered":
        a = (np.round(a * 4) / 4 + a * 0.01).astype(np.float32); b = (np.round(b * 4) / 4 + b * 0.01).astype(np.float32)
    elif kind == "surface":
        a[..., 2] = np.sin(a[..., 0] * 9) * 0.1; b[..., 2] = np.sin(b[..., 0] * 9) * 0.1
    elif kind == "far_offset":
        a += np.float32(37.0); b += np.float32(37.0)
    elif kind == "all_equal":
        b[:] = b[:, :1]
    elif kind == "big_coords":
        a *= np.float32(1000.0); b *= np.float32(1000.0)
    elif kind == "tiny_coords":
        a *= np.float32(1e-6); b *= np.float32(1e-6)
    return a, b


@pytest.mark.parametrize("kind", ["uniform", "same_x", "two_planes", "line", "clustered", "surface", "far_offset", "all_equal",
                                  "big_coords", "tiny_coords"])
def test_filter_surrogate_error_bound(kind):
    """the matrix-core filter is exact BECAUSE its surrogate s = |c|^2 - 2 q.c (on centred coordinates, bf16 hi/lo
    operands, one MFMA per 32 x 32 pairs) stays within E = 2^-14 R2 of the value it stands for, d - |q|^2: the tile filter keeps
    everything within tau = 4 E of the running minimum.  The bound is derived in csrc/chamfer_mfma.hip's header; here it is
    MEASURED: dpf_debug_nn_surrogate forms s with the kernel's own centring, fragments and MFMA for every pair of a cloud pair,
    and float64 says what it should have been."""
    _gpu()
    from dpf_nets_amd._lib import lib, current_stream
    n, m = 700, 1100
    a, b = chamfer_inputs(2000 + len(kind), 1, n, m)
    a, b = _adversarial(kind, a, b)
    for q, c in ((a[0], b[0]), (b[0], a[0])):                       # both directions of the pair
        tq, tc = torch.from_numpy(np.ascontiguousarray(q)).cuda(), torch.from_numpy(np.ascontiguousarray(c)).cuda()
        s = torch.empty((q.shape[0], c.shape[0]), device="cuda")
        out4 = torch.empty(4, device="cuda")
        rc = lib().dpf_debug_nn_surrogate(q.shape[0], tq.data_ptr(), c.shape[0], tc.data_ptr(), s.data_ptr(), out4.data_ptr(), current_stream())
        assert rc == 0
        torch.cuda.synchronize()
        mu, r2 = out4[:3].cpu().numpy(), float(out4[3])
        qc = (q - mu).astype(np.float32).astype(np.float64)         # fl(q - mu), as the kernel centres
        cc = (c - mu).astype(np.float32).astype(np.float64)
        assert r2 >= max((qc * qc).sum(1).max(), (cc * cc).sum(1).max()) * (1 - 1e-6)
        exact = (cc * cc).sum(1)[None, :] - 2.0 * qc @ cc.T         # d - |q|^2 on the centred coordinates
        err = float(np.abs(s.cpu().numpy().astype(np.float64) - exact).max())
        E = 2.0 ** -14 * r2
        # (+ 2^-20 R2: the fp32 rounding of |c|^2 and of the MFMA's accumulation, which the derivation leaves to tau's slack)
        assert err <= E + 2.0 ** -20 * r2, (kind, err / E if E > 0 else err)
        # the centring itself: fl(c - mu) against the real difference moves d - |q|^2 by < 2^-21 R2
        true = ((c.astype(np.float64) - mu) ** 2).sum(1)[None, :] - 2.0 * (q.astype(np.float64) - mu) @ (c.astype(np.float64) - mu).T
        assert float(np.abs(exact - true).max()) <= 2.0 ** -21 * r2 + 1e-300, kind


FUZZ_N = [1, 7, 31, 32, 33, 64, 100, 257, 500, 1000, 1024, 2047, 2048, 2049, 2080, 2500, 3000, 4100]
FUZZ_M = [1, 5, 32, 63, 65, 300, 777, 1024, 2048, 2111, 2500, 4096, 5000]


def test_nndistance_fuzz_all_kernels_vs_oracle():
    """a seeded, time-boxed run of tools/nn_fuzz.py inside the suite.  Random batch sizes, ragged tile counts, clouds
    that spill into a second pass of fragments, lattices (exact ties), duplicates, far offsets, 1e-6 .. 1e6 scales: the matrix-core
    filter, the SGPR-fed scan and the LDS-staged scan must return the CPU oracle's bits on every case it can afford (<= 4e7
    pair evaluations: nine cases in ten) and each other's on the rest."""
    BK = _gpu()
    import time
    rng = np.random.default_rng(20260501)
    S.set_threads(8)
    t0, cases, pinned, budget, want = time.time(), 0, 0, 100.0, 6000
    try:
        while cases < want and time.time() - t0 < budget:
            B = int(rng.integers(1, 9))
            n, m = int(rng.choice(FUZZ_N)), int(rng.choice(FUZZ_M))
            if cases % 2:                                            # half of them small: cheap for the oracle
                n, m = min(n, 500), min(m, 1024)
            kind = str(rng.choice(["uniform", "normal", "lattice", "dups", "offset", "line", "tiny", "huge"]))
            a = rng.random((B, n, 3), dtype=np.float32) - 0.5
            b = rng.random((B, m, 3), dtype=np.float32) - 0.5
            if kind == "normal":
                a = rng.standard_normal((B, n, 3)).astype(np.float32) * 0.2; b = rng.standard_normal((B, m, 3)).astype(np.float32) * 0.2
            elif kind == "lattice":
                a = np.round(a * 8) / 8; b = np.round(b * 8) / 8
            elif kind == "dups":
                b[:, m // 2:] = b[:, :m - m // 2]
                a[:, ::3] = b[:, :1]
            elif kind == "offset":
                a += np.float32(11.0); b += np.float32(11.0)
            elif kind == "line":
                a[..., 1:] = 0; b[..., 1:] = 0
            elif kind == "tiny":
                a *= np.float32(1e-6); b *= np.float32(1e-6)
            elif kind == "huge":
                a *= np.float32(1e6); b *= np.float32(1e6)
            a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
            tag = (cases, B, n, m, kind)
            ref = _run(BK, a, b, "brute")
            if B * n * m <= 4e7:
                _assert_bit_exact(ref, S.nndistance(a, b), tag)
                pinned += 1
            _assert_bit_exact(_run(BK, a, b, "mfma"), ref, tag + ("mfma",))
            if max(n, m) <= 8192:
                _assert_bit_exact(_run(BK, a, b, "small"), ref, tag + ("small",))
            cases += 1
    finally:
        S.set_threads(1)
    print("nndistance fuzz: %d cases in %.0f s, %d of them pinned on the CPU oracle" % (cases, time.time() - t0, pinned))
    assert cases >= 1500 and pinned >= 1000, (cases, pinned)          # (a fresh box runs all 6000 in a fraction of the budget)


def _assert_same_with_nans(got, ref, tag):
    """bit-exact where the reference value is a number; NaN where it is NaN (the payload / sign of a NaN differs between
    x86 and gfx950 arithmetic, its position does not)"""
    for g, r, name in zip(got, ref, ("dist1", "idx1", "dist2", "idx2")):
        assert g.dtype == r.dtype and g.shape == r.shape, (tag, name)
        if g.dtype == np.float32:
            rn = np.isnan(r)
            assert np.array_equal(np.isnan(g), rn), (tag, name, int((np.isnan(g) != rn).sum()))
            assert np.array_equal(g.view(np.uint32)[~rn], r.view(np.uint32)[~rn]), (tag, name, int((g[~rn] != r[~rn]).sum()))
        else:
            assert np.array_equal(g, r), (tag, name, int((g != r).sum()), np.argwhere(g != r)[:4].tolist())


@pytest.mark.parametrize("impl", IMPLS)
def test_finite_query_out_where_the_padding_sits(impl):
    """the LDS-staged kernels pad their candidate tiles with points at (3e38, 3e38, 3e38), "so far away that the
    distance is +inf" -- true unless a FINITE query sits out there too (distance 0 to the padding).  Every squared distance of
    such a query overflows; the reference answers (inf, 0) and so must every kernel."""
    BK = _gpu()
    B, n, m = 4, 300, 1100                      # 1100 candidates: a ragged last tile, i.e. padding
    a, b = chamfer_inputs(77, B, n, m)
    a[0, 5] = np.float32(3.0e38); a[1, 7] = (np.float32(2.9e38), np.float32(-3.0e38), np.float32(3.0e38))
    a[2, 0] = (np.float32(3.0e38), np.float32(0.0), np.float32(0.0)); b[3, 1099] = np.float32(-3.0e38)
    _assert_same_with_nans(_run(BK, a, b, impl), S.nndistance(a, b), impl)


NONFINITE_SHAPES = [(8, 700, 1100), (12, 2048, 2048), (8, 2500, 4100), (9, 100, 37)]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("shape", NONFINITE_SHAPES)
def test_nonfinite_input_gives_the_reference_result(shape, impl):
    """NaN / Inf coordinates.  The reference's scan (nndistance.cu:5-122) takes the first candidate of every
    batch of 512 unconditionally and compares with strict '<' / '>', so a NaN at candidate 0 latches (NaN, 0), at candidate
    512 k it makes batch k lose, elsewhere it is skipped; an Inf coordinate gives d = inf or NaN.  The oracle restates that
    loop; every implementation must return the same (values bit for bit where they are numbers, NaN where NaN, indices
    exactly) -- one scenario per cloud of the batch, the other clouds of the same launch staying finite."""
    BK = _gpu()
    B, n, m = shape
    a, b = chamfer_inputs(4000 + n + m, B, n, m)
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    def at(k, size):
        return min(k, size - 1)
    b[0, 0, 1] = nan                                   # candidate 0 of direction 0 (= query 0 of direction 1)
    b[1, at(5, m), 0] = nan                            # mid-batch candidate
    b[2, at(512, m), 2] = nan                          # first candidate of batch 1
    b[2, at(513, m), 0] = nan
    a[3, at(512, n), 0] = nan; a[3, at(1024, n), 1] = nan; a[3, n - 1, 2] = nan
    a[4, at(7, n), 0] = inf; b[4, at(9, m), 1] = -inf  # infinities on both sides
    b[5, 0, 0] = inf                                   # d = inf at candidate 0: replaced by any finite d
    a[5, 0, :] = inf                                   # a query at infinity: inf - inf = NaN against b[5, 0]
    a[6] *= np.float32(3.0e19); b[6] *= np.float32(3.0e19)   # finite coordinates, every squared distance overflows
    a[7, at(3, n)] = nan; b[7, at(3, m)] = nan; b[7, at(600, m), 0] = inf
    if B > 8:
        b[8, :, 0] = nan                               # every candidate NaN
    with np.errstate(all="ignore"):
        ref = S.nndistance(a, b)
    assert np.isnan(ref[0][0]).all() and (ref[1][0] == 0).all()          # the latch of nndistance.cu:26 at work
    _assert_same_with_nans(_run(BK, a, b, impl), ref, (shape, impl))


def test_nonfinite_input_through_the_cd_and_pairwise_entry_points():
    """The same contract through dpf_nndistance_cd (the benchmark step's Chamfer call: distances, indices and the
    per-cloud CD, NaN where the reference's `dl.mean(1) + dr.mean(1)` is NaN) and dpf_pairwise_cd."""
    BK = _gpu()
    B, n, m = 32, 2048, 2048
    a, b = chamfer_inputs(4500, B, n, m)
    b[3, 0, 0] = np.nan; a[7, 100, 1] = np.inf; b[9, 512, 2] = np.nan
    with np.errstate(all="ignore"):
        ref = S.nndistance(a, b)
        ref_cd = ref[0].mean(1, dtype=np.float64) + ref[2].mean(1, dtype=np.float64)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    d1, i1, d2, i2, cd = BK.NNDistanceCD(ta, tb)
    torch.cuda.synchronize()
    _assert_same_with_nans((d1.cpu().numpy(), i1.cpu().numpy(), d2.cpu().numpy(), i2.cpu().numpy()), ref, "cd")
    cd = cd.cpu().numpy()
    bad = ~np.isfinite(ref_cd)
    assert bad.sum() >= 2 and np.array_equal(~np.isfinite(cd), bad)
    np.testing.assert_allclose(cd[~bad], ref_cd[~bad], rtol=2e-6)
    from dpf_nets_amd.networks.utils import pairwise_CD
    mat = pairwise_CD(ta[:6], tb[:5]).cpu().numpy()
    with np.errstate(all="ignore"):
        for i in range(6):
            for j in range(5):
                r = S.nndistance(a[i:i + 1], b[j:j + 1])
                want = r[0].mean(dtype=np.float64) + r[2].mean(dtype=np.float64)
                if np.isfinite(want):
                    assert abs(mat[i, j] - want) <= 2e-6 * abs(want), (i, j)
                else:
                    assert not np.isfinite(mat[i, j]), (i, j)


def test_nndistance_first_minimum_on_exact_ties():
    BK = _gpu()
    B, n, m = 2, 197, 530
    a = np.round(detrng.uniform_f32(31, (B, n, 3), -3, 3))
    b = np.round(detrng.uniform_f32(32, (B, m, 3), -3, 3))
    b[:, 250:260] = b[:, 10:20]
    b[:, 520:530] = b[:, 3:13]
    got = _run(BK, a, b)
    _assert_bit_exact(got, S.nndistance(a, b), "ties")
    for impl in IMPLS:
        _assert_bit_exact(_run(BK, a, b, impl), S.nndistance(a, b), "ties-" + impl)
    dd = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1)
    assert np.array_equal(got[1], dd.argmin(2)) and np.array_equal(got[3], dd.argmin(1))


def test_nndistance_vs_reference_golden(golden_dir):
    BK = _gpu()
    gold = np.load(os.path.join(golden_dir, "chamfer.npz"))
    a, b = chamfer_inputs(21, 3, 257, 257)
    d1, i1, d2, i2 = _run(BK, a, b)
    np.testing.assert_allclose(d1, gold["eq257/ref_per_a"], rtol=1e-4, atol=2e-6)   # reference distChamfer
    np.testing.assert_allclose(d2, gold["eq257/ref_per_b"], rtol=1e-4, atol=2e-6)
    ok = gold["eq257/bf_gap1"] > 1e-5
    assert np.array_equal(i1[ok], gold["eq257/bf_idx1"][ok])


def test_nndistance_full_size_properties():
    """BASELINE.json sizes (B=32, N=2048 and the N=8192 stress config) through size-independent properties."""
    BK = _gpu()
    for (B, n) in ((32, 2048), (4, 8192)):
        a, b = chamfer_inputs(77, B, n, n)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        d1, i1, d2, i2 = BK.NNDistance(ta, tb)
        e1, j1, e2, j2 = BK.NNDistance(tb, ta)              # swapping the clouds swaps the outputs
        assert torch.equal(d1, e2) and torch.equal(i1, j2) and torch.equal(d2, e1) and torch.equal(i2, j1)
        assert (d1 >= 0).all() and (i1 >= 0).all() and (i1 < n).all()
        # the reported index really attains the reported distance, and nothing is closer
        nn = torch.gather(tb, 1, i1.long().unsqueeze(2).expand(-1, -1, 3))
        dx = nn - ta
        dd = (dx[..., 0] * dx[..., 0] + dx[..., 1] * dx[..., 1]) + dx[..., 2] * dx[..., 2]
        assert torch.equal(dd, d1)
        s1, si1, _, _ = BK.NNDistance(ta, ta.clone())      # a cloud against itself: distance 0 at its own index
        assert (s1 == 0).all() and torch.equal(si1.long(), torch.arange(n, device="cuda").expand(B, n))
        # permuting the candidates does not change the distances
        perm = torch.randperm(n, device="cuda")
        p1, _, _, _ = BK.NNDistance(ta, tb[:, perm].contiguous())
        assert torch.equal(p1, d1)


def test_nndistancegrad_vs_oracle_and_autograd():
    BK = _gpu()
    from dpf_nets_amd.metrics.StructuralLosses import nn_distance
    for (B, n, m) in ((2, 40, 55), (3, 300, 257), (8, 2048, 2048)):
        a, b = chamfer_inputs(41 + n, B, n, m)
        gd1 = detrng.normal_f32(42, (B, n)); gd2 = detrng.normal_f32(43, (B, m))
        d1, i1, d2, i2 = S.nndistance(a, b)
        r1, r2 = S.nndistancegrad(a, b, i1, i2, gd1, gd2)
        ta = torch.from_numpy(a).cuda().requires_grad_(True)
        tb = torch.from_numpy(b).cuda().requires_grad_(True)
        o1, o2 = nn_distance(ta, tb)
        ((o1 * torch.from_numpy(gd1).cuda()).sum() + (o2 * torch.from_numpy(gd2).cuda()).sum()).backward()
        np.testing.assert_allclose(ta.grad.cpu().numpy(), r1, rtol=1e-5, atol=1e-5)   # atomics: order-dependent rounding
        np.testing.assert_allclose(tb.grad.cpu().numpy(), r2, rtol=1e-5, atol=1e-5)


def test_input_checks_mirror_reference():
    BK = _gpu()
    a = torch.zeros(2, 8, 3)
    with pytest.raises(RuntimeError):
        BK.NNDistance(a, a)                                    # CPU tensor: CHECK_CUDA
    c = torch.zeros(2, 3, 8, device="cuda").transpose(1, 2)
    with pytest.raises(RuntimeError):
        BK.NNDistance(c, c)                                    # non-contiguous: CHECK_CONTIGUOUS


def test_pairwise_cd_and_fscore_match_the_expand_and_call_form():
    """utils.py:38-42 (f_score) and :90-117 (pairwise_CD) against the reference's own formulation
    evaluated with the oracle."""
    _gpu()
    from dpf_nets_amd.networks.utils import pairwise_CD, f_score, chamfer_distance
    N1, N2, n, m = 5, 7, 130, 200
    a, _ = chamfer_inputs(301, N1, n, n)
    b, _ = chamfer_inputs(302, N2, m, m)
    cds = pairwise_CD(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), bs=4).cpu().numpy()
    ref = np.zeros((N1, N2), np.float32)
    for i in range(N1):
        ai = np.ascontiguousarray(np.broadcast_to(a[i], (N2, n, 3)))       # utils.py:104-105
        d1, _, d2, _ = S.nndistance(ai, b)
        ref[i] = d1.mean(1) + d2.mean(1)
    np.testing.assert_allclose(cds, ref, rtol=2e-6)
    p, t = chamfer_inputs(303, 4, 300, 300)
    p = (t + 0.01 * p).astype(np.float32)
    f1 = f_score(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda(), threshold=0.001).cpu().numpy()
    ld, _, rd, _ = S.nndistance(p, t)
    prec, rec = 100.0 * (rd < 0.001).mean(1), 100.0 * (ld < 0.001).mean(1)
    np.testing.assert_allclose(f1, 2 * prec * rec / (prec + rec + 1e-7), rtol=1e-5)
    cd = chamfer_distance(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda())
    np.testing.assert_allclose(float(cd), float((ld.mean(1) + rd.mean(1)).mean()), rtol=1e-5)


def test_pairwise_cd_big_rows_take_the_matrix_core_kernel_and_keep_the_bits():
    """A row of pairwise_CD at evaluation size (40 clouds of 2048 points against one broadcast cloud) is served by
    the matrix-core filtered kernel through dpf_nndistance_strided_auto: distances and indices must equal the VALU
    scan's (dpf_nndistance_strided) bit for bit, and the CD row the expand-and-call form of utils.py:104-107."""
    BK = _gpu()
    from dpf_nets_amd._lib import lib, check, current_stream
    from dpf_nets_amd.networks.utils import pairwise_CD
    N2, n, m = 40, 2048, 2048
    a = detrng.uniform_f32(401, (2, n, 3), -0.3, 0.3)
    b = detrng.normal_f32(402, (N2, m, 3), 0.0, 0.15)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    outs = []
    for fn in (lib().dpf_nndistance_strided, lib().dpf_nndistance_strided_auto):
        d1 = torch.empty((N2, n), dtype=torch.float32, device="cuda"); d2 = torch.empty((N2, m), dtype=torch.float32, device="cuda")
        i1 = torch.empty((N2, n), dtype=torch.int32, device="cuda"); i2 = torch.empty((N2, m), dtype=torch.int32, device="cuda")
        check(fn(N2, n, ta[1].data_ptr(), 0, m, tb.data_ptr(), m * 3, d1.data_ptr(), i1.data_ptr(), d2.data_ptr(),
                 i2.data_ptr(), current_stream()), "strided")
        outs.append((d1, i1, d2, i2))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    cds = pairwise_CD(ta, tb)
    ref = torch.stack([torch.stack([sum(v.mean(1) for v in BK.NNDistance(ta[i:i + 1].contiguous(), tb[j:j + 1].contiguous())[::2])[0]
                                    for j in range(0, N2, 13)]) for i in range(2)])
    assert torch.allclose(cds[:, ::13], ref, rtol=2e-6, atol=0)


def test_pairwise_cd_one_launch_matrix():
    """dpf_pairwise_cd: the whole (N1, N2) matrix from one launch -- against per-pair nn_distance + mean (the oracle for the
    distances is pinned above; here the in-kernel means), ragged sizes (tails in both clouds, n != m, one tile and
    several), deterministic, rows of a row-sharded call identical to the full call."""
    BK = _gpu()
    from dpf_nets_amd.networks.utils import pairwise_CD
    for (N1, N2, n, m) in ((3, 5, 33, 600), (4, 3, 1100, 520), (2, 70, 2048, 2048)):
        a = detrng.normal_f32(500 + n, (N1, n, 3), 0.0, 0.2)
        b = detrng.uniform_f32(600 + m, (N2, m, 3), -0.4, 0.4)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        cds = pairwise_CD(ta, tb)
        assert cds.shape == (N1, N2) and torch.equal(cds, pairwise_CD(ta, tb))
        ref = torch.empty_like(cds)
        for i in range(N1):
            d1, _, d2, _ = BK.NNDistance(ta[i:i + 1].expand(N2, n, 3).contiguous(), tb)       # utils.py:104-107
            ref[i] = d1.double().mean(1).float() + d2.double().mean(1).float()
        assert torch.allclose(cds, ref, rtol=3e-6, atol=0), float((cds - ref).abs().max())
        assert torch.equal(pairwise_CD(ta, tb, bs=1), cds)                 # chunked rows: same bits
        assert torch.equal(pairwise_CD(ta, tb, shard_rows=True), cds)      # world size 1: the whole matrix
    # the C ABI rejects what it cannot serve
    from dpf_nets_amd._lib import lib
    assert lib().dpf_pairwise_cd(1, 1, 8, 64, ta.data_ptr(), tb.data_ptr(), cds.data_ptr(), cds.data_ptr(), 1 << 20, None) == -2
    assert lib().dpf_pairwise_cd(1, 1, 64, 64, ta.data_ptr(), tb.data_ptr(), cds.data_ptr(), cds.data_ptr(), 4, None) == -1


def test_nndistance_cd_fused_reduction():
    """dpf_nndistance_cd: the same four outputs as NNDistance, bit for bit, and cd = dist1.mean(1) + dist2.mean(1) -- on the
    matrix-core path (workgroup sums + finish) and on the fallback (small problem: scan + chamfer_reduce)."""
    BK = _gpu()
    # (4, 2048, 2048), (2, 2048, 1500): a rank's share of an 8-GPU job -- the LDS-staged scan with its own in-kernel finish
    for (B, n, m) in ((32, 2048, 2048), (3, 257, 300), (8, 2500, 2048), (4, 2048, 2048), (2, 2048, 1500), (1, 1024, 4000)):
        a, b = chamfer_inputs(900 + n, B, n, m)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        d1, i1, d2, i2, cd = BK.NNDistanceCD(ta, tb)
        for x, y in zip((d1, i1, d2, i2), BK.NNDistance(ta, tb)):
            assert torch.equal(x, y)
        ref = d1.double().mean(1) + d2.double().mean(1)
        assert torch.allclose(cd.double(), ref, rtol=2e-6, atol=0)
        assert torch.equal(cd, BK.NNDistanceCD(ta, tb)[4])              # deterministic
        ws = BK.CDWorkspace(B, n, m, ta.device)                          # caller-owned scratch: same bits, tickets back at zero
        for _ in range(3):
            assert torch.equal(cd, BK.NNDistanceCD(ta, tb, ws)[4])
        assert int(ws.tickets().abs().sum()) == 0 and not ws.dirty
    with pytest.raises(RuntimeError):
        BK.NNDistanceCD(ta, tb, BK.CDWorkspace(B + 1, n, m, ta.device))


def test_package_evaluation_path_is_the_single_launch_cd():
    """the fused search + CD call must be what the package's own evaluation helpers run, not a benchmark-only path.
    networks.utils.ChamferEvaluator owns a CDWorkspace per (shape, stream); metrics.evaluation_metrics.EMD_CD and
    chamfer_cd_per_cloud go through it and return the bits of an explicit NNDistanceCD call with a caller-owned workspace."""
    BK = _gpu()
    from dpf_nets_amd.networks.utils import ChamferEvaluator, chamfer_cd_per_cloud
    from dpf_nets_amd.metrics.evaluation_metrics import EMD_CD
    B, n, m = 12, 2048, 2048
    a, b = chamfer_inputs(4242, B, n, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    want = BK.NNDistanceCD(ta, tb, BK.CDWorkspace(B, n, m, ta.device))
    ev = ChamferEvaluator()
    for _ in range(3):
        got = ev(ta, tb)
        for x, y in zip(got, want):
            assert torch.equal(x, y)
    (ws,) = ev._ws.values()                                               # one workspace for the one (shape, stream), reused
    assert ws.shape == (B, n, m) and not ws.dirty and int(ws.tickets().abs().sum()) == 0
    assert torch.equal(chamfer_cd_per_cloud(ta, tb), want[4])
    res = EMD_CD(ta, tb, batch_size=B, reduced=False)["MMD-CD"]
    assert torch.equal(res, want[4])
    res2 = EMD_CD(ta, tb, batch_size=5, reduced=False)["MMD-CD"]           # ragged batches: three shapes, three workspaces
    assert torch.allclose(res2, want[4], rtol=2e-6, atol=0)


@pytest.mark.parametrize("B", [32, 4])
def test_nndistance_cd_ticket_finish_under_load(B):
    """The in-kernel finish of dpf_nndistance_cd reads other workgroups' sums (other CUs, other XCDs, lines that held the
    PREVIOUS call's sums): 150 calls on changing data, the same workspace, a second stream keeping the chip unevenly busy --
    every cd must equal the mean of the distances the same call wrote, and the tickets must be back at zero."""
    BK = _gpu()
    n = 2048                 # B = 32: the matrix-core kernel's finish; B = 4: the LDS-staged scan's (64 sums per cloud)
    g = torch.Generator(device="cuda").manual_seed(7)
    base = torch.randn(B, n, 3, device="cuda", generator=g) * 0.2
    other = torch.randn(B, n, 3, device="cuda", generator=g) * 0.2
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device="cuda")
    bad = 0
    ws = BK.CDWorkspace(B, n, n, "cuda:0")
    for it in range(150):
        a = (base * (1.0 + 0.01 * it)).contiguous()
        b = (other + 0.003 * it).contiguous()
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk2 = junk @ junk                      # uneven load on part of the chip while the call runs
        d1, i1, d2, i2, cd = BK.NNDistanceCD(a, b, ws if it % 5 else None)     # every fifth call: fresh scratch, cleared in-call
        ref = d1.double().mean(1) + d2.double().mean(1)
        bad += int((~torch.isclose(cd.double(), ref, rtol=3e-6, atol=0)).sum())
    torch.cuda.synchronize()
    assert bad == 0, bad
    assert int(ws.tickets().abs().sum()) == 0                             # tickets left at zero
    # a workspace left dirty (a call that raised, a ticket that a fault left behind) is cleared by the next call
    ws.tickets().fill_(3)
    ws.dirty = True
    d1, i1, d2, i2, cd = BK.NNDistanceCD(a, b, ws)
    assert torch.allclose(cd.double(), d1.double().mean(1) + d2.double().mean(1), rtol=3e-6, atol=0)
    assert int(ws.tickets().abs().sum()) == 0 and not ws.dirty


def test_nndistance_cd_first_call_inside_a_graph_capture():
    """A call without a caller-owned workspace INSIDE torch.cuda.graph takes its scratch from the graph's private pool and
    the library clears the tickets itself (a kernel node -- csrc/zero_fill.h), so replays on new data stay correct; nothing
    is cached anywhere (r04: there is no process-wide table any more)."""
    BK = _gpu()
    B, n = 32, 2048
    a, b = chamfer_inputs(4242, B, n, n)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert not hasattr(BK, "_CD_WORKSPACES")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            outs = BK.NNDistanceCD(ta, tb)
    for it in range(3):
        ta.mul_(1.0 + 0.05 * it)
        tb.add_(0.01 * it)
        graph.replay()
        torch.cuda.synchronize()
        ref = BK.NNDistance(ta, tb)
        for x, y in zip(outs[:4], ref):
            assert torch.equal(x, y)
        want = ref[0].double().mean(1) + ref[2].double().mean(1)
        assert torch.allclose(outs[4].double(), want, rtol=3e-6, atol=0)


def test_generative_evaluation_fragment_on_the_mirror():
    """evaluating.py:245-257 with the mirror's utils: three pairwise_CD matrices on the GPU -> COV, MMD, 1-NN accuracy; JSD of
    the clouds.  The matrices against the oracle's per-pair Chamfer (expand-and-call form), the metrics computed from the
    device matrices against the same functions on the oracle's matrices (exact: they are order statistics of the entries
    unless two entries tie within the matrices' 2e-6 agreement, which these continuous inputs do not)."""
    _gpu()
    from dpf_nets_amd.networks import utils as U
    Ng, Nr, n = 11, 8, 160
    gen = detrng.normal_f32(701, (Ng, n, 3), 0.0, 0.12)
    ref = (detrng.normal_f32(702, (Nr, n, 3), 0.0, 0.15) + np.float32(0.03)).astype(np.float32)
    tg, tr = torch.from_numpy(gen).cuda(), torch.from_numpy(ref).cuda()
    gg, tt, gt = U.pairwise_CD(tg, tg), U.pairwise_CD(tr, tr), U.pairwise_CD(tg, tr)

    def oracle_matrix(a, b):
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        for i in range(a.shape[0]):
            d1, _, d2, _ = S.nndistance(np.ascontiguousarray(np.broadcast_to(a[i], (b.shape[0],) + a.shape[1:])), b)
            out[i] = d1.mean(1) + d2.mean(1)
        return out
    ogg, ott, ogt = oracle_matrix(gen, gen), oracle_matrix(ref, ref), oracle_matrix(gen, ref)
    for got, want in ((gg, ogg), (tt, ott), (gt, ogt)):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=3e-6, atol=1e-9)
    o = [torch.from_numpy(x) for x in (ogg, ogt, ott)]
    assert U.COV(gt) == U.COV(o[1]) and U.COV(gt, axis=0) == U.COV(o[1], axis=0)
    assert abs(U.MMD(gt) - U.MMD(o[1])) <= 3e-6 * U.MMD(o[1])
    for k in (1, 3):
        assert U.KNN(gg, gt, tt, k) == U.KNN(o[0], o[1], o[2], k)
    jsd = U.JSD(tg.cpu().numpy(), tr.cpu().numpy(), warning=False)
    assert 0.0 < jsd < 1.0


def test_compiled_extension_gives_the_python_backends_bits():
    """csrc/torch_ext/structural_losses_backend.cpp -- the five functions of the reference's pybind module, compiled over the C
    ABI -- against the ctypes module of the same names: identical tensors for Chamfer forward / backward, identical match /
    cost / gradients for approx-EMD; a non-contiguous or mistyped tensor is refused as the reference refuses it."""
    BK = _gpu()
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend_native as native
    a, b = chamfer_inputs(77, 6, 700, 1100)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    got, ref = native.NNDistance(ta, tb), BK.NNDistance(ta, tb)
    for x, y in zip(got, ref):
        assert x.dtype == y.dtype and torch.equal(x, y)
    _assert_bit_exact([t.cpu().numpy() for t in got], S.nndistance(a, b), "native")
    g1, g2 = torch.rand_like(got[0]), torch.rand_like(got[2])
    for x, y in zip(native.NNDistanceGrad(ta, tb, got[1], got[3], g1, g2), BK.NNDistanceGrad(ta, tb, ref[1], ref[3], g1, g2)):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-7)          # (the scattered term is added with float atomics: last bits vary)
    sa, sb = ta[:3, :256].contiguous(), tb[:3, :320].contiguous()
    m1, t1 = native.ApproxMatch(sa, sb)
    m2, t2 = BK.ApproxMatch(sa, sb)
    assert torch.equal(m1, m2) and m1.shape == (3, 320, 256) and t1.shape == (3, 2 * (256 + 320))
    assert torch.allclose(native.MatchCost(sa, sb, m1), BK.MatchCost(sa, sb, m2), rtol=2e-6, atol=0)      # (partial sums meet in a different order)
    for x, y in zip(native.MatchCostGrad(sa, sb, m1), BK.MatchCostGrad(sa, sb, m2)):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-7)
    with pytest.raises(RuntimeError, match="contiguous"):
        native.NNDistance(ta[:, ::2], tb)
    with pytest.raises(RuntimeError, match="float32"):
        native.NNDistance(ta.double(), tb)
