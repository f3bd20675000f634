import sys
import torch
from nbed_amd.backend import HipBackend

be = HipBackend()
n = int(sys.argv[1])
bounds = [int(a) for a in sys.argv[2].split(",")]
eri = be.synth_eri(n)
packed = be.eri_pack(eri, n)
for k in range(len(bounds) - 1):
    T = (bounds[k] + bounds[k + 1]) // 2
    for (r, s) in ((4 * T + 1, 2), (4 * T + 1, 4 * T + 1), (4 * T + 3, 4 * T)):
        dm = torch.zeros(2, n, n, dtype=torch.float64, device=be.device)
        dm[0, r, s] += 1.0
        dm[0, s, r] += 1.0
        ref = be.jk_sym(eri, dm).clone()
        got = be.jk_packed(packed, dm).clone()
        d = (got - ref).abs()
        print(f"chunk {k} rows [{bounds[k]},{bounds[k+1]}) T={T} (r,s)=({r},{s}): J err {d[0].max().item():.2e} (|J| {ref[0].abs().max().item():.2e}) K err {d[1].max().item():.2e}", flush=True)
