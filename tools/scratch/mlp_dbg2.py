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
dev = lambda t: t.to(DEV)
xs = x[735:736].repeat(128, 1)
def err_of(lo, hi):
    w1m = w1.clone(); w1m[:lo] = 0; w1m[hi:] = 0
    b1m = b1.clone(); b1m[:lo] = -30; b1m[hi:] = -30
    xd = xs.double()
    xn = F.layer_norm(xd, (C,), ln_w.double(), ln_b.double(), 1e-5)
    ref = xd + F.gelu(xn @ w1m.double().t() + b1m.double()) @ w2.double().t() + b2.double()
    pk = Tk.PackedMlp(dev(ln_w), dev(ln_b), 1e-5, dev(w1m), dev(b1m), dev(w2), dev(b2))
    got = Tk.mlp(dev(xs).clone(), pk).cpu().double()
    return float((got - ref).abs().max())
lo, hi = 192, 384
while hi - lo > 1:
    mid = (lo + hi) // 2
    if err_of(lo, mid) > 5e-6: hi = mid
    else: lo = mid
print("culprit hidden unit", lo, "err", err_of(lo, lo + 1))
xn = F.layer_norm(xs[0].double(), (C,), ln_w.double(), ln_b.double(), 1e-5)
pre = float(xn @ w1[lo].double() + b1[lo].double())
print("pre-activation", pre, "gelu", float(F.gelu(torch.tensor(pre, dtype=torch.float64))), "tile", lo // 32, "unit in tile", lo % 32)
print("w2 column absmax", float(w2[:, lo].abs().max()))
u = lo
w2p = torch.zeros(C, 4 * C); w2p[0, u] = 1.0; w2p[1, u] = 0.5; w2p[2, u + 1] = 1.0
pk = Tk.PackedMlp(dev(ln_w), dev(ln_b), 1e-5, dev(w1), dev(b1), dev(w2p), dev(b2 * 0))
got = Tk.mlp(dev(xs).clone(), pk).cpu().double()
xn = F.layer_norm(xs.double(), (C,), ln_w.double(), ln_b.double(), 1e-5)
hid = F.gelu(xn @ w1.double().t() + b1.double())
print("unit", u, "expected", float(hid[0, u]), "kernel", float(got[0, 0] - xs[0, 0].double()), "half-weight", float(got[0, 1] - xs[0, 1].double()) * 2,
      "| next unit expected", float(hid[0, u + 1]), "kernel", float(got[0, 2] - xs[0, 2].double()))
import numpy as np
v = np.float32(hid[0, u]); print("f16 neighbours", float(np.float16(v)), float(np.nextafter(np.float16(v), np.float16(1))), "diff to expected", float(got[0, 0] - xs[0, 0].double() - hid[0, u]))
