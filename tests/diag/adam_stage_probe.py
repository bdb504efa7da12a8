"""Stage by stage: the reference's Adam update as PyTorch-ROCm executes it vs an IEEE numpy emulation (which stage rounds differently?)."""
import math, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
n = 1 << 20
gen = torch.Generator(device="cuda").manual_seed(n)
p = torch.randn(n, device="cuda", generator=gen)
m = torch.randn(n, device="cuda", generator=gen) * 1e-2
v = torch.rand(n, device="cuda", generator=gen) * 1e-3
vm = v * (1.0 + torch.rand(n, device="cuda", generator=gen))
g = torch.randn(n, device="cuda", generator=gen) * 0.3
f32 = np.float32
def cnt(a, b): return int((a.view(np.int32) != b.view(np.int32)).sum())
def fma(a, b, c): return (np.float64(a) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
lr, b1, b2, eps, wd, step = 2.56e-4, 0.9, 0.999, 1e-8, 1e-6, 1
P, M, V, VM, G = (t.cpu().numpy() for t in (p, m, v, vm, g))
# torch side, stage by stage (same calls as Adam._update)
ea, es, mx, ps = [m.clone()], [v.clone()], [vm.clone()], [p.clone()]
torch._foreach_mul_(ea, b1); torch._foreach_add_(ea, [g], alpha=1 - b1)
torch._foreach_mul_(es, b2); torch._foreach_addcmul_(es, [g], [g], value=1 - b2)
torch._foreach_maximum_(mx, es)
den = torch._foreach_sqrt(mx)
bc1, bc2 = 1 - b1 ** step, math.sqrt(1 - b2 ** step)
eac = torch._foreach_div(ea, bc1)
t_den0 = den[0].clone()
torch._foreach_div_(den, bc2); t_den1 = den[0].clone()
torch._foreach_add_(den, eps); t_den2 = den[0].clone()
upd = torch._foreach_mul(ps, wd); t_upd0 = upd[0].clone()
torch._foreach_addcdiv_(upd, eac, den, value=lr); t_upd1 = upd[0].clone()
torch._foreach_sub_(ps, upd)
# numpy side
m2 = fma(f32(1 - b1), G, M * f32(b1)); print("exp_avg", cnt(m2, ea[0].cpu().numpy()))
v2 = fma(f32(1 - b2), G * G, V * f32(b2)); print("exp_avg_sq", cnt(v2, es[0].cpu().numpy()))
vm2 = np.maximum(VM, v2); print("max", cnt(vm2, mx[0].cpu().numpy()))
d0 = np.sqrt(vm2); print("sqrt", cnt(d0, t_den0.cpu().numpy()))
d1 = d0 * f32(1.0 / bc2); print("div bc2 (x * f32(1/bc2))", cnt(d1, t_den1.cpu().numpy()), " IEEE division:", cnt(d0 / f32(bc2), t_den1.cpu().numpy()))
d2 = d1 + f32(eps); print("add eps", cnt(d2, t_den2.cpu().numpy()))
mc = m2 * f32(1.0 / bc1); print("div bc1 (x * f32(1/bc1))", cnt(mc, eac[0].cpu().numpy()), " IEEE division:", cnt(m2 / f32(bc1), eac[0].cpu().numpy()))
pw = P * f32(wd); print("p*wd", cnt(pw, t_upd0.cpu().numpy()))
q = mc / d2
u1 = fma(f32(lr), q, pw); print("addcdiv fma(lr, q, pw)", cnt(u1, t_upd1.cpu().numpy()), " unfused:", cnt(pw + f32(lr) * q, t_upd1.cpu().numpy()))
pn = P - u1; print("p - upd", cnt(pn, ps[0].cpu().numpy()))
