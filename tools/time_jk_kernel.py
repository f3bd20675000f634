"""Average duration of the packed J/K MAIN kernel alone (HIP events inside libnbx, NBX_PROF_JK_DENSE) at size N:
    python tools/time_jk_kernel.py 148"""
import sys

import torch

from nbed_amd import _nbx
from nbed_amd.backend import HipBackend

be = HipBackend()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
eri = be.synth_eri(n)
packed = be.eri_pack(eri, n)
dm = torch.randn(2, n, n, dtype=torch.float64, device=be.device)
dm = dm + dm.transpose(1, 2)
for _ in range(5):
    be.jk_packed(packed, dm)
be.synchronize()
be.profile(True, slots=[_nbx.PROF_JK_DENSE])
be.profile_reset()
for _ in range(50):
    be.jk_packed(packed, dm)
be.synchronize()
ms, cnt = be.profile_read(_nbx.PROF_JK_DENSE)
be.profile(False)
nbytes = be.lib.nbx_eri_packed_bytes(n, 0, n)
print(f"N={n}: main J/K kernel {ms / cnt * 1e3:.1f} us over {cnt} launches, {nbytes / (ms / cnt * 1e-3) / 1e12:.2f} TB/s on the {nbytes / 1e9:.3f} GB it reads")
