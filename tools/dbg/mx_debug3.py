import sys
import torch
from nbed_amd.backend import HipBackend

be = HipBackend()
n = int(sys.argv[1])
bounds = [int(a) for a in sys.argv[2].split(",")]
nb = n // 4
eri = be.synth_eri(n)
packed = be.eri_pack(eri, n)
tri = lambda k: k * (k + 1) // 2
bad = []
for T in range(nb):
    k = max(i for i in range(len(bounds) - 1) if bounds[i] <= T)
    for C in range(T + 1):
        r, s = 4 * T + 2, 4 * C + 1
        dm = torch.zeros(2, n, n, dtype=torch.float64, device=be.device)
        dm[0, r, s] += 1.0
        dm[0, s, r] += 1.0
        ref = be.jk_sym(eri, dm)[0].clone()
        got = be.jk_packed(packed, dm)[0].clone()
        e = (got - ref).abs().max().item()
        if e > 1e-12:
            off = (tri(T) - tri(bounds[k]) + C) * 16
            bad.append((k, T, C, off // 512, (off % 512) // 2))
print("bad blocks:", len(bad))
for b in bad:
    print("chunk %d T=%d C=%d slot %d ptid %d" % b)
