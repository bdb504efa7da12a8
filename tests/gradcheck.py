"""Bounds for the three scalars the goldens keep per parameter gradient g (oracle/gen_golden.py::_grad_projection): the sum,
the dot with a fixed N(0,1) vector r, and the 1-norm.

r04.  r03 bounded all three by 3e-3 * ||g||_1, which for an n-element gradient is ~sqrt(n) times the
magnitude of the random projection itself (a third of it for the 64 x 128 FiLM matrices).  An elementwise relative error
`tol` with independent signs moves a projection g.w by ~ tol * ||g||_2 * ||w||_inf-ish; the bounds here are
    |d(g.r)|  <= 4 tol ||g||_2            (r ~ N(0,1): g.r itself is ~ ||g||_2; 4 sigma)
    |d sum|   <= tol ||g||_2 sqrt(n)      (Cauchy-Schwarz against w = 1)
    |d ||g||_1| <= tol ||g||_2 sqrt(n)
plus an absolute slack of 1e-6 of the largest gradient norm of the model (a parameter whose gradient is pure rounding
noise -- a BatchNorm bias in front of another BatchNorm -- has nothing to be relative to).  ||g||_2 is taken from the
gradient under test (the golden does not store it; at these tolerances the two norms agree to tol)."""
import math


def check_projections(named_grads, projections, gold_of, tol, tag=""):
    """named_grads: {name: cpu tensor}; projections: {name: (sum, g.r, l1)} of the gradients under test; gold_of(name) ->
    the golden triple."""
    gmax = max(float(v.double().norm()) for v in named_grads.values())
    slack = 1e-6 * gmax
    worst = 0.0
    for k, v in projections.items():
        ref = gold_of(k)
        g = named_grads[k].double()
        n2, n = float(g.norm()), g.numel()
        for j, bound in ((1, 4.0 * tol * n2), (0, tol * n2 * math.sqrt(n)), (2, tol * n2 * math.sqrt(n))):
            err = abs(float(v[j]) - float(ref[j]))
            worst = max(worst, err / (bound + slack))
            assert err <= bound + slack, (tag, k, ("sum", "g.r", "l1")[j], float(v[j]), float(ref[j]), n2, n)
    return worst
