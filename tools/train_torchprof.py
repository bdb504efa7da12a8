import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpf_nets_amd import networks as nets, synthetic as SY
B, N, n_flows, G = 32, 2048, 21, 128
dec = nets.LocalCondRNVPDecoder(n_flows, 64, G).cuda().train()
dec.flatten_parameters()
tgt, z, g = SY.synthetic_inputs(3, B, N, G)
tp = torch.from_numpy(tgt).cuda(); tg = torch.from_numpy(g).cuda()
pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
nll = nets.PointFlowNLL()
def fwd():
    return dec(tp, tg, mode="inverse")
for _ in range(2):
    ps, mus, lvs = fwd(); loss = nll(ps + [tp], [pm] + mus, [pl] + lvs); loss.backward()
from torch.profiler import profile, ProfilerActivity
for name, fn in (("forward", lambda: fwd()), ):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn(); torch.cuda.synchronize()
    print(name); print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
ps, mus, lvs = fwd()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    loss = nll(ps + [tp], [pm] + mus, [pl] + lvs); torch.cuda.synchronize()
print("loss"); print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8, max_name_column_width=60))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    loss.backward(); torch.cuda.synchronize()
print("backward"); print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
