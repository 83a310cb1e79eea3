import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from al3d import token_ops as Tk
DEV = "cuda:0"
T_, C = 1000, 96
g = torch.Generator().manual_seed(T_ + C)
x = (torch.randn(T_, C, generator=g) * 1.7 + 0.3)
ln_w, ln_b = torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.1
w1, b1 = torch.randn(4 * C, C, generator=g) / C ** 0.5, torch.randn(4 * C, generator=g) * 0.1
w2, b2 = torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5, torch.randn(C, generator=g) * 0.1
xd = x.double()
xn = F.layer_norm(xd, (C,), ln_w.double(), ln_b.double(), 1e-5)
ref = xd + F.gelu(xn @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
dev = lambda t: t.to(DEV)
pk = Tk.PackedMlp(dev(ln_w), dev(ln_b), 1e-5, dev(w1), dev(b1), dev(w2), dev(b2))
got = Tk.mlp(dev(x).clone(), pk).cpu().double()
err = (got - ref).abs()
print("per-row max err (first 8):", err.max(1).values[:8])
print("rows with err > 1e-6:", (err.max(1).values > 1e-6).nonzero().flatten()[:40].tolist(), int((err.max(1).values > 1e-6).sum()))
print("per-col max err:", err.max(0).values[:40])
r = int(err.max(1).values.argmax()); print("worst row", r, err[r].max(), "x row absmax", x[r].abs().max(), "hidden absmax", F.gelu(xn[r] @ w1.double().t() + b1.double()).abs().max())

def run(tag, ln_w, ln_b, w1, b1, w2, b2, x):
    xd = x.double()
    xn = F.layer_norm(xd, (C,), ln_w.double(), ln_b.double(), 1e-5)
    ref = xd + F.gelu(xn @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    pk = Tk.PackedMlp(dev(ln_w), dev(ln_b), 1e-5, dev(w1), dev(b1), dev(w2), dev(b2))
    got = Tk.mlp(dev(x).clone(), pk).cpu().double()
    err = (got - ref).abs().max(1).values
    print(tag, "max", float(err.max()), "median", float(err.median()), "rows>1e-6:", int((err > 1e-6).sum()))
run("A: w1=0", ln_w, ln_b, w1 * 0, b1, w2, b2, x)
run("B: gamma=0", ln_w * 0, ln_b + 0.5, w1, b1, w2, b2, x)
run("C: x small spread", ln_w, ln_b, w1, b1, w2, b2, x * 0.01)
xs = x.clone(); xs[:, :] = x[0:1, :]
run("D: all rows equal", ln_w, ln_b, w1, b1, w2, b2, xs)
w1e = torch.zeros(4 * C, C); w1e[:C] = torch.eye(C)
w2e = torch.zeros(C, 4 * C); w2e[:, :C] = torch.eye(C)
b1e = torch.full((4 * C,), 30.0)
run("E: expose LN", ln_w, ln_b, w1e, b1e, w2e, b2 * 0, x)
run("F: expose LN, gamma=1 beta=0", ln_w * 0 + 1, ln_b * 0, w1e, b1e, w2e, b2 * 0, x)
xn32 = F.layer_norm(x, (C,), ln_w, ln_b, 1e-5).double()
xn64 = F.layer_norm(x.double(), (C,), ln_w.double(), ln_b.double(), 1e-5)
print("torch fp32 LN err:", float((xn32 - xn64).abs().max()))
xs = x.clone(); xs[:, :] = x[735:736, :]
run("G: all rows = row 735", ln_w, ln_b, w1, b1, w2, b2, xs)
xs = x.clone(); xs[735] = x[0]
run("H: row 735 := row 0", ln_w, ln_b, w1, b1, w2, b2, xs)
for rep in range(2):
    run("I: repeat plain", ln_w, ln_b, w1, b1, w2, b2, x)
xs = x.clone(); xs[:, :] = x[735:736, :]
run("J: row735, b1+10", ln_w, ln_b, w1, b1 + 10, w2, b2, xs)
xnr = F.layer_norm(xs, (C,), None, None, 1e-5)
run("K: row735 pre-normalised, ln identity", ln_w * 0 + 1, ln_b * 0, w1, b1, w2, b2, xnr)
run("L: row735, w2 small", ln_w, ln_b, w1, b1, w2 * 1e-3, b2, xs)
# which output channels / which hidden tiles matter: zero half of the hidden units
for lo, hi in ((0, 192), (192, 384), (0, 96), (96, 192), (0, 32), (32, 64), (64, 96)):
    w1m = w1.clone(); w1m[:lo] = 0; w1m[hi:] = 0
    b1m = b1.clone(); b1m[:lo] = -30; b1m[hi:] = -30
    run("M: hidden units [%d,%d)" % (lo, hi), ln_w, ln_b, w1m, b1m, w2, b2, xs)
