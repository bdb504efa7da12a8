"""Measured error of the EXPANDED squared distance |x|^2 + |y|^2 - 2 x.y in the approximate EMD (what an MFMA formulation of
csrc/emd.hip would compute), against the difference form (dx^2 + dy^2) + dz^2 the reference and the oracle use
(approxmatch.cu:54, oracle/structural_oracle.c sqd).  CPU only, numpy fp32; the auction itself is the oracle's, restated
vectorised (oracle_approxmatch, :136-190), so the two runs differ in d^2 alone.

    python tests/diag/emd_expanded_form_error.py [n] [clouds]

Variants of the expanded form:
  fp32   operands exact (a 3-way bf16 split over the K slots makes every product exact), fp32 accumulation in the MFMA's order
  f16x2  operands rounded to fp16 hi + fp16 lo (22 bits), products exact, fp32 accumulation
"""
import sys
import numpy as np

f32 = np.float32


def d2_diff(a, c):
    dx = c[None, :, 0] - a[:, None, 0]; dy = c[None, :, 1] - a[:, None, 1]; dz = c[None, :, 2] - a[:, None, 2]
    return (dx * dx + dy * dy) + dz * dz


def split16(v):
    hi = v.astype(np.float16).astype(f32)
    lo = (v - hi).astype(np.float16).astype(f32)
    return hi + lo          # what the two fragments together represent (22 bits)


def d2_expanded(a, c, operands):
    if operands == "f16x2":
        a, c = split16(a), split16(c)
    # the accumulator starts from the fp32 norms (row constant + column constant), then takes -2 x.y product by product
    na = ((a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) + a[:, 2] * a[:, 2]).astype(f32)
    nc = ((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2]).astype(f32)
    acc = (na[:, None] + nc[None, :]).astype(f32)
    for j in range(3):
        acc = (acc + (f32(-2.0) * a[:, None, j]).astype(np.float64) * c[None, :, j]).astype(f32)   # exact product, one rounding
    return np.maximum(acc, f32(0))


def approxmatch(a, c, d2):
    n, m = len(a), len(c)
    multiL, multiR = (f32(1), f32(n // m)) if n >= m else (f32(m // n), f32(1))
    match = np.zeros((m, n), f32)
    remainL = np.full(n, multiL, f32); remainR = np.full(m, multiR, f32)
    for j in range(7, -2, -1):
        level = f32(-(4.0 ** j))
        w = np.exp(level * d2).astype(f32)                  # (n, m)
        suml = (f32(1e-9) + (w * remainR[None, :]).sum(1, dtype=f32)).astype(f32)
        ratioL = remainL / suml
        sumr = (w * ratioL[:, None]).sum(0, dtype=f32) * remainR
        consumption = np.minimum(remainR / (sumr + f32(1e-9)), f32(1))
        ratioR = consumption * remainR
        remainR = np.maximum(f32(0), remainR - sumr)
        wk = w * ratioL[:, None] * ratioR[None, :]
        match += wk.T
        remainL = np.maximum(f32(0), remainL - wk.sum(1, dtype=f32))
    return match


def cost(a, c, match):
    d = np.sqrt(d2_diff(a, c).astype(np.float64))
    return float((match.T.astype(np.float64) * d).sum())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    clouds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rng = np.random.default_rng(5)
    print("n = m = %d, %d clouds, unit-cube-centred points (|x| <= 0.5) and a shifted copy (+0.3: what an uncentred cloud does)" % (n, clouds))
    for shift in (0.0, 0.3):
        for i in range(clouds):
            a = (rng.random((n, 3), dtype=f32) - f32(0.5) + f32(shift)).astype(f32)
            c = (a[rng.permutation(n)] + rng.normal(0, 0.02, (n, 3)).astype(f32)).astype(f32) if i % 2 else \
                (rng.random((n, 3), dtype=f32) - f32(0.5) + f32(shift)).astype(f32)
            ref = d2_diff(a, c)
            m0 = approxmatch(a, c, ref)
            c0 = cost(a, c, m0)
            row = []
            for ops in ("fp32", "f16x2"):
                d2 = d2_expanded(a, c, ops)
                m1 = approxmatch(a, c, d2)
                row.append("%s: max|d2 err| %.1e  cost rel %.1e  match max|err| %.1e (of %.2f)  row-sum err %.1e" % (
                    ops, float(np.abs(d2 - ref).max()), abs(cost(a, c, m1) - c0) / c0, float(np.abs(m1 - m0).max()), float(m0.max()),
                    float(np.abs(m1.sum(0) - m0.sum(0)).max())))
            print("shift %.1f cloud %d (%s)  cost %.4f\n    %s\n    %s" % (shift, i, "near pairs" if i % 2 else "uniform", c0, row[0], row[1]))


if __name__ == "__main__":
    main()
