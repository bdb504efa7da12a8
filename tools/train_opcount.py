"""Which ATen ops (and how many) one training step of the flattened decoder issues, per phase -- finds stray per-layer
tensor ops (r03: ~380 aten::copy_ per step -> hipMemcpyWithStream -> __amd_rocclr_copyBuffer)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd import networks as nets, synthetic as SY, distributed as D
from torch.profiler import profile, ProfilerActivity
B, N, n_flows, G = 32, 2048, 21, 128
dec = nets.LocalCondRNVPDecoder(n_flows, 64, G).cuda().train()
dec.flatten_parameters()
params = list(dec.parameters())
arena = D.GradArena(params)
opt = nets.Adam(params, lr=2.56e-4, weight_decay=1e-6, betas=(0.9, 0.999), amsgrad=True)
tgt, z, g = SY.synthetic_inputs(3, B, N, G)
tp = torch.from_numpy(tgt).cuda(); tg = torch.from_numpy(g).cuda()
pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
nll = nets.PointFlowNLL()


def phases():
    yield "zero_grad", lambda st: arena.zero_grad()
    yield "forward", lambda st: st.update(out=dec(tp, tg, mode="inverse"))
    yield "loss", lambda st: st.update(loss=nll(st["out"][0] + [tp], [pm] + st["out"][1], [pl] + st["out"][2]))
    yield "backward", lambda st: st["loss"].backward()
    yield "allreduce+step", lambda st: (arena.allreduce(), opt.step())


for it in range(6):
    st = {}
    for name, fn in phases():
        if it < 5:
            fn(st)
            continue
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=(name in sys.argv[1:])) as prof:
            fn(st); torch.cuda.synchronize()
        ev = prof.key_averages(group_by_stack_n=6 if name in sys.argv[1:] else 0)
        rows = sorted(ev, key=lambda e: -e.count)[:10]
        print("== %s" % name)
        for e in rows:
            print("  %-60s n=%5d cpu_us=%9.0f" % (e.key[:60], e.count, e.cpu_time_total))
            if name in sys.argv[1:] and e.key in ("aten::copy_", "hipMemcpyWithStream") and e.stack:
                for s in e.stack[:6]:
                    print("        ", s)
