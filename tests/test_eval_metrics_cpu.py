"""The consumers of the Chamfer matrices in the reference's generative evaluation (lib/networks/utils.py: COV :120, MMD :124,
KNN :128, get_voxel_occ_dist :45, JSD :83, AverageMeter :8) as mirrored by dpf_nets_amd.networks.utils, against vectors
captured from the reference's own functions (oracle/gen_golden_metrics.py -> tests/golden/eval_metrics.npz)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_metrics.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def U():
    from dpf_nets_amd.networks import utils
    return utils


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_cov_mmd_knn_vs_reference_golden(gold, U, tag):
    gt, gg, tt = (torch.from_numpy(gold[tag + "/" + k]) for k in ("gt", "gg", "tt"))
    assert U.COV(gt) == float(gold[tag + "/cov1"]) and U.COV(gt, axis=0) == float(gold[tag + "/cov0"])
    assert U.MMD(gt) == float(gold[tag + "/mmd1"]) and U.MMD(gt, axis=0) == float(gold[tag + "/mmd0"])
    assert U.KNN(gg, gt, tt, 1) == float(gold[tag + "/knn1"])
    assert U.KNN(gg, gt, tt, 3) == float(gold[tag + "/knn3"])
    assert U.KNN(gg, gt, tt, 2, sqrt=True) == float(gold[tag + "/knn2_sqrt"])


@pytest.mark.parametrize("tag", ["v1", "v2"])
def test_voxel_occupancy_and_jsd_vs_reference_golden(gold, U, tag):
    c1, c2 = gold[tag + "/c1"], gold[tag + "/c2"]
    occ = U.get_voxel_occ_dist(c1, warning=False)
    assert occ.dtype == np.float64 and occ.shape == (28, 28, 28)
    assert np.array_equal(occ, gold[tag + "/occ1"])                 # every point in the reference's voxel (edges, faces, NaN)
    assert abs(U.JSD(c1, c2, warning=False) - float(gold[tag + "/jsd"])) <= 1e-12
    assert U.JSD(c1, c1, warning=False) == pytest.approx(0.0, abs=1e-12)


def test_average_meter(U):
    m = U.AverageMeter()
    m.update(2.0)
    m.update(4.0, n=3)
    assert (m.val, m.sum, m.count, m.avg) == (4.0, 14.0, 4, 3.5)
    m.reset()
    assert (m.val, m.sum, m.count, m.avg) == (0, 0, 0, 0)


def test_mirror_exports_what_evaluating_py_imports(U):
    """evaluating.py:9-10: from lib.networks.utils import AverageMeter, distChamferCUDA, f_score, pairwise_CD, JSD, COV, MMD, KNN"""
    for name in ("AverageMeter", "distChamferCUDA", "f_score", "pairwise_CD", "JSD", "COV", "MMD", "KNN"):
        assert callable(getattr(U, name)), name
    # training.py:7 and train_ae.py:16 take save_model / cnt_params from the same module
    lin = torch.nn.Linear(3, 2)
    lin.bias.requires_grad_(False)
    assert U.cnt_params(lin.parameters()) == 6
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        U.save_model({"w": lin.weight.detach()}, os.path.join(d, "m.pkl"))
        assert torch.equal(torch.load(os.path.join(d, "m.pkl"), weights_only=False)["w"], lin.weight.detach())
