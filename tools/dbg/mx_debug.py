import sys
import torch
from nbed_amd.backend import HipBackend

be = HipBackend()
torch.manual_seed(7)
for n in [int(a) for a in sys.argv[1:]]:
    eri = be.synth_eri(n)
    packed = be.eri_pack(eri, n)
    dm = torch.randn(2, n, n, dtype=torch.float64, device=be.device)
    dm = dm + dm.transpose(1, 2)
    ref = be.jk_sym(eri, dm).clone()
    got = be.jk_packed(packed, dm).clone()
    d = (got - ref).abs()
    print(f"N={n}: J err {d[0].max().item():.3e}  Ka err {d[1].max().item():.3e}  Kb err {d[2].max().item():.3e}")
    for x in range(3):
        rows = (d[x].max(dim=1).values > 1e-9).nonzero().flatten().tolist()
        cols = (d[x].max(dim=0).values > 1e-9).nonzero().flatten().tolist()
        print(f"  mat {x}: bad rows {len(rows)} {rows[:12]}..{rows[-4:]}; bad cols {len(cols)} {cols[:12]}..{cols[-4:]}")
    bad = (d[0] > 1e-9).nonzero()
    print("  J bad entries:", bad.shape[0], bad[:10].tolist())
    del eri, packed
    torch.cuda.empty_cache()
