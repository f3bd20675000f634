"""A/B timing of the packed J/K kernel at N = 148: mean duration of the main kernel by the library's HIP events (JK slot),
three rounds of 200 back-to-back builds.  Run once per library under test, alternating, on ONE box (boxes of the pool differ
by 5-15 %):  NBX_LIB=scratch/libnbx_m4_NAME.so python tools/time_jk_m4_ab.py NAME   (tools/build_m4_variant.sh builds them)."""
import sys
import torch
from nbed_amd import _nbx
from nbed_amd.backend import HipBackend
be = HipBackend()
n = 148
eri = be.synth_eri(n)
packed = be.eri_pack(eri, n)
del eri
dm = torch.randn(2, n, n, dtype=torch.float64, device=be.device)
dm = dm + dm.transpose(1, 2)
for _ in range(20):
    be.jk_packed(packed, dm)
torch.cuda.synchronize()
out = []
for r in range(3):
    be.profile(True, slots=[_nbx.PROF_JK_DENSE])
    be.profile_reset()
    for _ in range(200):
        be.jk_packed(packed, dm)
    torch.cuda.synchronize()
    ms, cnt = be.profile_read(_nbx.PROF_JK_DENSE)
    be.profile(False)
    out.append(ms / cnt * 1e3)
print(sys.argv[1] if len(sys.argv) > 1 else "", " ".join(f"{x:.1f}" for x in out), "us", flush=True)
