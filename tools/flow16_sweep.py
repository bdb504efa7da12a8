"""Duration of the fused eval flow kernel alone (HIP events around a captured graph of back-to-back launches, as bench.py's
time_kernel) for small batches, 32-point tiles (csrc/flow.hip) against 16-point tiles (csrc/flow16.hip).
    python tools/flow16_sweep.py [--layers 14] [--batches 4,8,16,32]        (DPF_FLOW16_CW=2|4|8 picks the workgroup shape)"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=14)
    ap.add_argument("--batches", default="4,8,16,32")
    ap.add_argument("--latent", type=int, default=128)
    a = ap.parse_args()
    from dpf_nets_amd._lib import lib
    out = {"layers": a.layers, "cw_env": os.environ.get("DPF_FLOW16_CW", "auto"), "rows": []}
    dev = torch.device("cuda", 0)
    for B in [int(x) for x in a.batches.split(",")]:
        args = bench.parse(["--batch", str(B), "--layers", str(a.layers), "--latent", str(a.latent)])
        dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, B)
        row = {"B": B}
        for mode, name in ((0, "tile32"), (1, "tile16")):
            lib().dpf_flow_set_tile16(mode)
            ks = bench.make_kernels(dec, z, g, tgt_pm, a.layers, args.precision)
            for k in ks:
                k()
            torch.cuda.synchronize()
            row[name + "_us"] = bench.time_kernel(ks[1])
        lib().dpf_flow_set_tile16(-1)
        row["film_us"] = bench.time_kernel(ks[0])
        row["nn_us"] = bench.time_kernel(ks[2])
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
