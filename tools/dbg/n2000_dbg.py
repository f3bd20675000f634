import numpy as np, torch
from nbed_amd.backend import HipBackend
from nbed_amd import synth
be = HipBackend()
N = 2000
pr = synth.problem(be, N, (512, 512), N // 12)
s_d = be.asarray(pr["S"])
hv = be.asarray(np.asarray(pr["hcore"])[None] + np.asarray(pr["V_emb"]))
x_d = be.sym_pow_fast(s_d, -0.5, pr["S"])
print("x_d", x_d.shape, float(x_d.abs().max()), "hv", float(hv.abs().max()))
chk = x_d @ s_d @ x_d
print("X S X - I", float((chk - torch.eye(N, device=chk.device, dtype=chk.dtype)).abs().max()))
x2 = torch.stack([x_d, x_d]).contiguous()
t1 = be.gemm(x2, hv)
print("t1", float(t1.abs().max()), "ref", float((x2 @ hv).abs().max()), "diff", float((t1 - x2 @ hv).abs().max()))
fo = be.gemm(t1, x2)
print("fo", float(fo.abs().max()), "diff", float((fo - (x2 @ hv) @ x2).abs().max()))
w, v = be.eigh(fo)
print("w", float(w.min()), float(w.max()))
