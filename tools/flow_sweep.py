"""Timing sweep of the flow / film / chamfer kernels (HIP events) for tuning.  Not a test."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = ["bench.py"] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    for L in (14, 63):
        args.layers = L
        for prec in ("bf16", "bf16x3", "bf16x6"):
            args.precision = prec
            dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev)
            kt = bench.kernel_timings(dec, z, g, tgt_pm, L, prec)
            print("FW=%s L=%d %-7s film %.1f us flow %.1f us nn %.1f us" % (
                os.environ.get("DPF_FLOW_WAVES", "auto"), L, prec, kt["film_kernel"], kt["flow_kernel"], kt["nn_kernel"]), flush=True)


if __name__ == "__main__":
    main()
