"""bench.py's timed loop with several steps in flight (--streams S): every stream replays its own captured graph with its own
buffers, so steps that overlap on the chip must still produce exactly what a step produces alone."""
import os
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_steps_in_flight_produce_the_single_step_results():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, ROOT)
    import bench
    args = types.SimpleNamespace(batch=6, points=700, layers=6, latent=128, precision="f16x3", lists=False)
    dev = torch.device("cuda", 0)
    dec, state, n_flows, z, g, tgt, tgt_pm = bench.build_workload(args, dev, args.batch)
    step = bench.make_step(dec, z, g, tgt_pm, args.layers)
    ref = [t.clone() for t in step()]
    torch.cuda.synchronize()
    S = 3
    streams = [torch.cuda.Stream() for _ in range(S)]
    graphs, outs = [], []
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            step()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            outs.append(step())
        graphs.append(gr)
    for o in outs:                                   # poison the graphs' output buffers, then replay round-robin
        for t in o:
            t.zero_()
    for i in range(11):
        with torch.cuda.stream(streams[i % S]):
            graphs[i % S].replay()
    torch.cuda.synchronize()
    assert len({o[0].data_ptr() for o in outs}) == S                 # each stream has its own buffers
    for o in outs:
        for got, want in zip(o, ref):
            assert torch.equal(got, want)
    # the in-flight kernel timing helper runs and reports the three kernels
    kt = bench.kernel_timings_in_flight(dec, z, g, tgt_pm, args.layers, args.precision, 2, steps=12, warm=4)
    assert set(kt) == {"film_kernel", "flow_kernel", "nn_kernel"} and all(v > 0 for v in kt.values())
