"""s_memtime stamps of the training kernels' phases (workgroup (1,3), thread 0; needs libdpf_hip_prof.so: make -C dpf_nets_amd/csrc prof).
Ticks are shader-clock-ish (~1.9 GHz under this load: tbwd2's 57 K ticks are its 29.6 us)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpf_nets_amd import _lib
_lib.lib_path = lambda: os.path.join(ROOT, "dpf_nets_amd", "libdpf_hip_prof.so")
from dpf_nets_amd import networks as nets, synthetic as SY
B, N, G = 32, 2048, 128
h = _lib.lib()
h.dpf_debug_set_kprof.argtypes = [ctypes.c_void_p]
dec = nets.LocalCondRNVPDecoder(2, 64, G).cuda().train()
dec.flatten_parameters()
tgt, z, g = SY.synthetic_inputs(3, B, N, G)
x, tg = torch.from_numpy(tgt).cuda(), torch.from_numpy(g).cuda()
pm, pl = torch.zeros(B, 3, N).cuda(), torch.full((B, 3, N), -3.6).cuda()
nll = nets.PointFlowNLL()
h.dpf_train_graph_set_enabled(0)
prof = torch.zeros((8, 8), dtype=torch.int64, device="cuda")
rows = []
for it in range(6):
    if it == 2:
        h.dpf_debug_set_kprof(prof.data_ptr())
    dec.zero_grad()
    ps, mus, lvs = dec(x, tg, mode="inverse")
    nll(ps + [x], [pm] + mus, [pl] + lvs).backward()
    torch.cuda.synchronize()
    if it >= 2:
        rows.append(prof.cpu().numpy().copy())
h.dpf_debug_set_kprof(None)
t = np.median(np.stack(rows), axis=0)
names = {0: ("tstats_h1", ["entry", "prologue issued (bn0_fold, loads)", "first barrier passed", "both branches done", "exit"]),
         5: ("tstats_h1 prologue (stamps 0, 5, 6, 7, 1 of kernel 0)", None),
         1: ("tbwd1", ["entry", "prologue issued (coefs of pass 3, loads)", "first barrier passed", "main part done", "row published, ticket taken"]),
         7: ("tbwd1 main part (fused column sums)", ["first branch recomputed", "column sums' counter seen", "pass-3 totals loaded, coefficients, gradient finished", "coupling transform, stores", "first branch's sums", ]),
         2: ("tbwd2 (r05)", ["entry", "prologue done (staging, first recompute, means from the role workgroups, dh1 fragments)", "both tiles done", "exit (reduction)"]),
         6: ("tbwd2 role workgroup 1 (r05; same clock as the ordinary workgroup above: compare ABSOLUTE stamps below)", ["entry", "rows loaded and summed", "barrier", "means stored, left the CU", "counter raised"]),
         3: ("tbwd2 prologue (r05, large-batch form)", ["address setup done", "weight loads issued", "tables in LDS", "means: loads back, sums done", "partials in LDS", "barrier passed"])}
print("absolute stamps: tbwd2 ordinary", [int(v - t[2, 0]) for v in t[3, :6]], "exit", int(t[2, 3] - t[2, 0]), " role", [int(v - t[2, 0]) for v in t[6, :5]])
for kid, (name, labels) in names.items():
    print(name)
    if labels is None:
        seq = [("entry", 0), ("loads back, butterflies done", 5), ("first barrier", 6), ("fold (double) done", 7), ("fragments stored, DMA issued", 1)]
        for (la, ia), (lb, ib) in zip(seq[:-1], seq[1:]):
            print("   %-48s %8.0f ticks" % (la + " -> " + lb, t[0, ib] - t[0, ia]))
        continue
    for i in range(1, len(labels)):
        print("   %-48s %8.0f ticks" % (labels[i - 1] + " -> " + labels[i], t[kid, i] - t[kid, i - 1]))
    print("   %-48s %8.0f ticks" % ("total", t[kid, len(labels) - 1] - t[kid, 0]))
