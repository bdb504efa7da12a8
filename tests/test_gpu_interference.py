"""r06: the product's kernels beside a neighbour that issues MFMAs (profiles/r06_packed_f32_vs_mfma.txt).

On gfx950 a packed fp32 VALU instruction whose LOW half reads the HIGH word of a VGPR source pair (op_sel on a VGPR operand) loses that
half in lanes 48-63 while ANOTHER wave of its SIMD issues MFMAs at certain distances (tools/ubench/pk_vs_mfma_forms.hip) -- found behind
approx-EMD's run-to-run differing bits.  The other wave need not belong to the same kernel: tools/interfere/mfma_interferer.hip runs on a
second stream and corrupts a self-checking chain of such instructions on the first (the POSITIVE CONTROL below: a harness in which it
stays clean proves nothing).  csrc/Makefile keeps that instruction form -- and any packed fp32 beside MFMAs -- out of every object;
here the kernels that still hold (plain) packed fp32, and the ones that issue MFMAs themselves, must return the bits they return alone."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import flow_oracle as FO
from oracle import gprior_oracle as GO
from oracle.gen_golden import chamfer_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GAPS = (4, 15)            # wait states between the neighbour's MFMAs: the two cadences that hit hardest in the micro-test


def _interferer():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    idir = os.path.join(ROOT, "tools", "interfere")
    so = os.path.join(idir, "libmfma_interferer.so")
    src = os.path.join(idir, "mfma_interferer.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so], check=True)
    I = ctypes.CDLL(so)
    I.interferer_launch.restype = ctypes.c_int
    I.interferer_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    I.pk_chain_check.restype = ctypes.c_int
    I.pk_chain_check.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return I


class Neighbour:
    """2 048 single-wave workgroups (two per SIMD) of MFMAs with `gap` wait states between them on a side stream, until told to stop
    (or after 40 000 x 512 MFMAs per wave: it can not hang the box)."""
    def __init__(self, I, gap):
        self.I, self.gap = I, gap
        self.side = torch.cuda.Stream()
        self.stop = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.sink = torch.zeros(2048, device="cuda")

    def __enter__(self):
        torch.cuda.synchronize()
        assert self.I.interferer_launch(2048, self.gap, self.stop.data_ptr(), 40000, self.sink.data_ptr(), self.side.cuda_stream) == 0
        return self

    def __exit__(self, *a):
        self.stop.fill_(1)
        torch.cuda.synchronize()
        self.iterations = float(self.sink.min())        # < 40 000: the flag stopped it, i.e. it was running to the end


def _bits(ts):
    return [t.detach().clone().view(torch.int32) if t.dtype == torch.float32 else t.detach().clone() for t in ts]


def _beside(I, op, reps):
    """op's results alone (twice) and `reps` times beside the neighbour, per gap -> asserts bit equality; returns the neighbour's
    iteration counts"""
    ref = _bits(op())
    torch.cuda.synchronize()
    again = _bits(op())
    assert all(torch.equal(a, b) for a, b in zip(ref, again)), "differs from itself without any neighbour"
    alive = []
    for gap in GAPS:
        nb = Neighbour(I, gap)
        with nb:
            for r in range(reps):
                out = _bits(op())
                torch.cuda.current_stream().synchronize()          # (this stream only: a device-wide wait would wait for the neighbour)
                bad = sum(int((a != b).sum()) for a, b in zip(ref, out))
                assert bad == 0, "gap %d, repeat %d: %d elements differ beside the MFMA neighbour" % (gap, r, bad)
        alive.append(nb.iterations)
    return alive


def test_the_harness_interferes():
    """POSITIVE CONTROL: chains of `v_pk_fma_f32 ... op_sel:[0,1,0]` that check themselves against v_fma_f32 -- clean alone, wrong
    beside the neighbour (only low halves, only lanes 48-63: pk_vs_mfma_waves2.hip); if this stays clean the tests below prove nothing."""
    I = _interferer()
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")

    def launch():
        bad.zero_()
        assert I.pk_chain_check(256, 200, bad.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.current_stream().synchronize()
        return int(bad)
    assert launch() == 0 and launch() == 0
    hit = {}
    for gap in GAPS + (5, 3, 31):              # (which cadence hits hardest differs from box to box: 4 / 15 here, 3 .. 5 on another)
        with Neighbour(I, gap):
            hit[gap] = sum(launch() for _ in range(10))
        if hit[gap] > 1000:
            break
    assert launch() == 0                       # ... and clean again once the neighbour is gone
    if max(hit.values()) <= 1000:              # (measured: ~1.7e6 wrong lane results per launch of 2.6e7)
        pytest.skip("the neighbour did not disturb the self-checking chain on this box (%r): the tests below still run, but say less" % hit)


def _nn(impl, B, n, m, seed):
    from dpf_nets_amd._lib import lib
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
    a, b = chamfer_inputs(seed, B, n, m)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()

    def run():
        old = BK.NN_IMPL
        BK.NN_IMPL = "brute" if impl == "small" else impl
        old_small = lib().dpf_nn_small_mode(1 if impl == "small" else -1)
        try:
            return list(BK.NNDistance(ta, tb))
        finally:
            BK.NN_IMPL = old
            lib().dpf_nn_small_mode(old_small)
    return run


@pytest.mark.parametrize("impl,B", [("brute", 8), ("small", 4), ("mfma", 32)])
def test_chamfer_search_beside_an_mfma_neighbour(impl, B):
    """the SGPR-fed scan and the LDS-staged scan keep packed fp32 (plain forms and op_sel on SGPR pairs: measured immune); the
    filter issues MFMAs itself"""
    I = _interferer()
    alive = _beside(I, _nn(impl, B, 2048, 2048, 40 + B), 12)
    assert min(alive) > 0


@pytest.mark.parametrize("matrix", [False, True])
def test_approx_emd_beside_an_mfma_neighbour(matrix):
    """the packed-VALU family (packed fp32 throughout, op_sel on SGPR pairs only) and the matrix-core family (no packed fp32 at all)"""
    I = _interferer()
    from dpf_nets_amd._lib import lib
    from dpf_nets_amd.metrics.StructuralLosses import StructuralLossesBackend as BK
    a, b = chamfer_inputs(77, 4, 1024, 1024)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()

    def run():
        prev = lib().dpf_emd_set_matrix_path(1 if matrix else 0)
        try:
            match, temp, cost = BK.ApproxMatchCost(ta, tb)
        finally:
            lib().dpf_emd_set_matrix_path(prev)
        return [match, cost]
    _beside(I, run, 8)


def test_latent_prior_flow_beside_an_mfma_neighbour():
    """csrc/gprior.hip held the affected form until r06 (twelve `v_pk_fma_f32 ... op_sel:[0,1,0]` from a vector-times-scalar
    expression in a remainder loop; never seen wrong -- 40 launches of the r05 object beside the neighbour repeat too -- but the form is
    gone and csrc/Makefile's gate keeps it out)"""
    I = _interferer()
    from dpf_nets_amd import networks as nets
    n_flows, nf, G, B = 7, 128, 128, 64
    dec = nets.GlobalRNVPDecoder(n_flows, nf, G)
    dec.load_state_dict(FO.to_torch(GO.make_gprior_state(5, n_flows, nf, G)), strict=True)
    dec = dec.cuda().eval()
    g = torch.from_numpy(GO.gprior_inputs(5, B, G)).cuda()

    def run():
        with torch.no_grad():
            out = dec(g, mode="direct")
        flat = []
        for o in out:
            flat += list(o) if isinstance(o, (list, tuple)) else [o]
        return [t for t in flat if torch.is_tensor(t)]
    _beside(I, run, 20)


def test_flow_stack_and_encoder_beside_an_mfma_neighbour():
    """the fused coupling stack (MFMAs + scalar VALU) and the PointNet encoder (its accumulator sum was a packed add until r06)"""
    I = _interferer()
    from dpf_nets_amd import networks as nets
    sd = FO.to_torch(FO.make_decoder_state(31, 5, 64, 128))
    tgt, z, g = FO.synthetic_inputs(31, 8, 2048, 128)
    dec = nets.LocalCondRNVPDecoder(5, 64, 128, weight_std=0.01)
    dec.load_state_dict(sd, strict=True)
    dec = dec.cuda().eval()
    tz, tg = torch.from_numpy(z).cuda(), torch.from_numpy(g).cuda()
    enc = nets.PointNetCloudEncoder(3, 64, [128, 256, 512]).cuda().eval()

    def run():
        with torch.no_grad():
            ps, mus, lvs = dec(tz, tg, mode="direct")
            e = enc(tz)
        return [ps[-1], mus[-1], lvs[-1]] + ([e] if torch.is_tensor(e) else [t for t in e if torch.is_tensor(t)])
    _beside(I, run, 10)


def test_training_step_beside_an_mfma_neighbour():
    """forward + backward of the 63-layer training stack (315 dependent launches of flow_train.hip's kernels, BatchNorm statistics,
    FiLM training kernels) and the latent prior flow's training kernels: outputs, input gradients and every parameter gradient"""
    I = _interferer()
    from dpf_nets_amd import networks as nets
    torch.manual_seed(0)
    dec = nets.LocalCondRNVPDecoder(21, 64, 128).cuda().train()
    dec.flatten_parameters()
    tgt, z, g = FO.synthetic_inputs(7, 8, 2048, 128)
    tp, tg = torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()
    prior = nets.GlobalRNVPDecoder(3, 64, 128, weight_std=0.05).cuda().train()
    gg = torch.from_numpy(GO.gprior_inputs(9, 64, 128)).cuda()

    def run():
        dec.zero_grad(set_to_none=True)
        prior.zero_grad(set_to_none=True)
        tpi = tp.clone().requires_grad_(True)
        ps, mus, lvs = dec(tpi, tg, mode="inverse")
        (ps[0].square().mean() + sum(lvs).mean()).backward()
        out = prior(gg, mode="direct")
        flat = []
        for o in out:
            flat += list(o) if isinstance(o, (list, tuple)) else [o]
        flat = [t for t in flat if torch.is_tensor(t)]
        sum(t.square().mean() for t in flat).backward()
        return [ps[0].detach(), tpi.grad] + [p.grad for p in dec.parameters() if p.grad is not None] + \
               [t.detach() for t in flat] + [p.grad for p in prior.parameters() if p.grad is not None]
    _beside(I, run, 4)
