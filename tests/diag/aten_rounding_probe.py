"""How does PyTorch-ROCm round the individual ops of the reference's Adam step on this GPU?  Each op against numpy float32
candidates (IEEE division / reciprocal multiply, fused / unfused multiply-add)."""
import numpy as np, torch
torch.manual_seed(0)
n = 1 << 20
x = (torch.rand(n, device="cuda") * 1e-3 + 1e-9)
y = torch.randn(n, device="cuda") * 1e-2
z = torch.randn(n, device="cuda")
X, Y, Z = (t.cpu().numpy() for t in (x, y, z))
f32 = np.float32
def cnt(a, b): return int((a.view(np.int32) != b.view(np.int32)).sum())
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
s = 0.0316069770620507          # sqrt(1 - 0.999)
got = torch._foreach_sqrt([x])[0].cpu().numpy()
print("sqrt vs IEEE:", cnt(got, np.sqrt(X)))
got = torch._foreach_div([x], s)[0].cpu().numpy()
print("foreach_div(scalar): vs x / f32(s):", cnt(got, X / f32(s)), " vs x * f32(1/s):", cnt(got, X * f32(1.0 / s)), " vs x * (f32(1)/f32(s)):", cnt(got, X * (f32(1) / f32(s))),
      " vs (x / float64 s) rounded:", cnt(got, (X.astype(np.float64) / s).astype(f32)))
d = x.clone(); torch._foreach_div_([d], s); got = d.cpu().numpy()
print("foreach_div_(scalar): vs x / f32(s):", cnt(got, X / f32(s)), " vs x * f32(1/s):", cnt(got, X * f32(1.0 / s)), " vs double:", cnt(got, (X.astype(np.float64) / s).astype(f32)))
d = x.clone(); torch._foreach_add_([d], 1e-8); got = d.cpu().numpy()
print("foreach_add_(scalar): vs x + f32(1e-8):", cnt(got, X + f32(1e-8)), " vs double:", cnt(got, (X.astype(np.float64) + 1e-8).astype(f32)))
lr = 2.56e-4
u = z.clone(); torch._foreach_addcdiv_([u], [y], [x], value=lr); got = u.cpu().numpy()
q = Y / X
print("foreach_addcdiv_: vs fma(f32 lr, y/x, z):", cnt(got, fma(np.full(n, f32(lr)), q, Z)), " vs z + f32(lr)*q unfused:", cnt(got, Z + f32(lr) * q),
      " vs z + (lr*y)/x:", cnt(got, Z + (f32(lr) * Y) / X), " vs all in double:", cnt(got, (Z.astype(np.float64) + lr * (Y.astype(np.float64) / X.astype(np.float64))).astype(f32)),
      " vs double lr, f32 quotient:", cnt(got, (Z.astype(np.float64) + lr * q.astype(np.float64)).astype(f32)))
w = torch._foreach_mul([z], 1e-6)[0].cpu().numpy()
print("foreach_mul(scalar): vs z * f32(1e-6):", cnt(w, Z * f32(1e-6)), " vs double:", cnt(w, (Z.astype(np.float64) * 1e-6).astype(f32)))
a = y.clone(); torch._foreach_add_([a], [z], alpha=0.1); got = a.cpu().numpy()
print("foreach_add_(alpha): vs fma(f32 alpha):", cnt(got, fma(np.full(n, f32(0.1)), Z, Y)), " vs double alpha:", cnt(got, (Y.astype(np.float64) + 0.1 * Z.astype(np.float64)).astype(f32)))
