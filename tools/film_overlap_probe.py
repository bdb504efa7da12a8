"""What does the flow kernel lose when the FiLM kernel of another batch runs beside it?  Times, per iteration:
flow alone | film then flow (one stream) | film on a second stream while flow runs (captured as one graph each)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dpf_nets_amd._lib import lib, check, PREC, MODE


def main():
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extra"]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, 32)
    stack = dec.stack()
    L, prec = args.layers, args.precision
    canon, meta, packed, G, prec = stack._ensure(prec, dev, L)
    B, _, N = z.shape
    Lb = lib()
    film = [torch.empty(Lb.dpf_flow_film_floats(L, B), dtype=torch.float32, device=dev) for _ in range(2)]
    p_out, sum_lv = torch.empty_like(z), torch.empty_like(z)
    eps = float(stack.layers[0].eps_value)
    side = torch.cuda.Stream()

    def do_film(buf, stream):
        check(Lb.dpf_flow_film(L, B, G, PREC[prec], packed.data_ptr(), g.data_ptr(), film[buf].data_ptr(), eps, stream), "film")

    def do_flow(buf, stream):
        check(Lb.dpf_flow_forward(L, B, N, MODE["direct"], PREC[prec], packed.data_ptr(), meta.data_ptr(), film[buf].data_ptr(),
                                  z.data_ptr(), p_out.data_ptr(), None, sum_lv.data_ptr(), None, None, None, eps, stream), "flow")

    def variant(kind, reps=20):
        cur = torch.cuda.current_stream()
        for it in range(reps):
            if kind == "flow":
                do_flow(0, cur.cuda_stream)
            elif kind == "seq":
                do_film(0, cur.cuda_stream); do_flow(0, cur.cuda_stream)
            else:                       # film of the next batch beside this batch's flow
                side.wait_stream(cur)
                do_film((it + 1) & 1, side.cuda_stream)
                do_flow(it & 1, cur.cuda_stream)
                cur.wait_stream(side)

    do_film(0, torch.cuda.current_stream().cuda_stream); do_film(1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for kind in ("flow", "seq", "overlap"):
        s0 = torch.cuda.Stream()
        with torch.cuda.stream(s0):
            variant(kind, 3)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            variant(kind)
        for _ in range(20):
            gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record(); e1.synchronize()
        print("%-8s %.2f us per iteration" % (kind, e0.elapsed_time(e1) / 200 * 1e3))


if __name__ == "__main__":
    main()
