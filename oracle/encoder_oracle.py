"""CPU restatement of dpf-nets' PointNet cloud encoder and the max-pool the models apply to it.

TEST INFRASTRUCTURE -- the checker, never the thing measured or shipped.  Only tests/, bench.py's cpu_baseline
leg and __graft_entry__.smoke() may import it.

Parity status: PINNED.  tests/test_oracle_golden.py checks it against vectors captured from the reference's own
`PointNetCloudEncoder` imported on CPU (oracle/gen_golden.py -> tests/golden/encoder.npz).

  PointNetCloudEncoder.__init__/forward   lib/networks/encoders.py:9-28
      features = init_sd (SharedDot 3->64, no bias) . BatchNorm1d . ReLU, then for 128, 256, 512:
                 sd{i} (SharedDot, no bias) . BatchNorm1d . ReLU                       on (B, C, N)
  the max over the points                  lib/networks/models.py:85,131,175 (torch.max(features, dim=2)[0])
"""
import math

import numpy as np
import torch

from . import detrng
from .flow_oracle import shared_dot, batch_norm

CHANNELS = (3, 64, 128, 256, 512)


def layer_names(n_features=(128, 256, 512)):
    return ["init_sd"] + ["sd%d" % i for i in range(len(n_features))]


def make_encoder_state(seed, init_n_channels=3, init_n_features=64, n_features=(128, 256, 512)):
    """Deterministic non-trivial weights (numpy dict, names as encoders.py:15-25): kaiming-uniform SharedDot
    magnitudes (layers.py:33), BatchNorm affine / running statistics randomised so eval BN is not the identity."""
    st = {}
    cin = init_n_channels
    for name, cout in zip(layer_names(n_features), (init_n_features,) + tuple(n_features)):
        b = math.sqrt(6.0 / (cout * cin))
        st["features.%s.weight" % name] = detrng.uniform_f32(detrng.key(seed, name + ".w"), (1, cout, cin), -b, b) * 2.0
        bn = "features.%s_bn." % name
        st[bn + "weight"] = detrng.uniform_f32(detrng.key(seed, name + ".g"), (cout,), 0.5, 1.5)
        st[bn + "bias"] = detrng.normal_f32(detrng.key(seed, name + ".b"), (cout,), 0.0, 0.1)
        st[bn + "running_mean"] = detrng.normal_f32(detrng.key(seed, name + ".rm"), (cout,), 0.0, 0.1)
        st[bn + "running_var"] = detrng.uniform_f32(detrng.key(seed, name + ".rv"), (cout,), 0.5, 1.5)
        st[bn + "num_batches_tracked"] = np.array(0, dtype=np.int64)
        cin = cout
    return st


def encoder_inputs(seed, B, N):
    """(B,3,N) clouds at the scale the training configs feed the encoder (all_scaled.yaml:21-22)."""
    return detrng.uniform_f32(detrng.key(seed, "enc_x"), (B, 3, N), -0.5, 0.5)


def encoder_features(state, x, training=False, stats_out=None, n_features=(128, 256, 512)):
    """encoders.py:27-28: (B,3,N) -> (B,512,N).  state: dict of torch tensors."""
    h = x
    for name in layer_names(n_features):
        bn = "features.%s_bn." % name
        h = shared_dot(state["features.%s.weight" % name], h)
        h = batch_norm(h, state[bn + "running_mean"], state[bn + "running_var"], state[bn + "weight"], state[bn + "bias"],
                       training, stats_out, "features.%s_bn" % name)
        h = torch.relu(h)
    return h


def encoder_max(state, x, training=False, stats_out=None):
    """models.py:84-85: the encoder followed by the max over the points -> (B,512)."""
    return torch.max(encoder_features(state, x, training, stats_out), dim=2)[0]
