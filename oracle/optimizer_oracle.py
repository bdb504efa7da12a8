"""CPU restatement of dpf-nets' own Adam (with the AMSGrad and coupled weight-decay variants its scripts use) and its
cosine learning-rate / beta2 schedule.

TEST INFRASTRUCTURE -- the checker, never the thing shipped.  Parity status: PINNED by tests/golden/optimizer.npz,
captured from the reference's `lib.networks.optimizers.Adam` / `LRUpdater` (oracle/gen_golden.py).

  Adam.step      lib/networks/optimizers.py:15-75   (per parameter, in this order)
  LRUpdater      lib/networks/optimizers.py:78-98
"""
import math

import numpy as np
import torch


def adam_step(p, grad, state, lr, betas, eps, weight_decay, amsgrad):
    """One step for one parameter; `state` is a dict created on first use (optimizers.py:30-40)."""
    if not state:
        state["step"] = 0
        state["exp_avg"] = torch.zeros_like(p)
        state["exp_avg_sq"] = torch.zeros_like(p)
        if amsgrad:
            state["max_exp_avg_sq"] = torch.zeros_like(p)
    beta1, beta2 = betas
    state["step"] += 1
    state["exp_avg"] = state["exp_avg"] * beta1 + (1 - beta1) * grad                   # :53
    state["exp_avg_sq"] = state["exp_avg_sq"] * beta2 + (1 - beta2) * grad * grad       # :54
    if amsgrad:
        state["max_exp_avg_sq"] = torch.max(state["max_exp_avg_sq"], state["exp_avg_sq"])   # :57
        denom = state["max_exp_avg_sq"].sqrt()
    else:
        denom = state["exp_avg_sq"].sqrt()
    bc1 = 1 - beta1 ** state["step"]                                                  # :63
    bc2 = math.sqrt(1 - beta2 ** state["step"])                                       # :64
    exp_avg_c = state["exp_avg"] / bc1
    denom_c = denom / bc2 + eps
    if weight_decay != 0:                                                             # :69-72
        return p - (p * weight_decay + lr * exp_avg_c / denom_c)
    return p - lr * exp_avg_c / denom_c                                               # :74


def lr_update(epoch_length, cycle_length, min_lr, max_lr, beta1, min_beta2, max_beta2, epoch, iteration):
    """optimizers.py:89-98 -> (lr, (beta1, beta2))"""
    rel_epoch = epoch % cycle_length
    cur = (rel_epoch * epoch_length + iteration) / (cycle_length * epoch_length)
    lr = min_lr + 0.5 * (max_lr - min_lr) * (1.0 + np.cos(np.pi * cur))
    b2 = min_beta2 + 0.5 * (max_beta2 - min_beta2) * (1.0 + np.cos(np.pi * cur))
    return lr, (beta1, b2)
