"""A cold-start embedded SCF run at the bench size (conv 1e-6, DIIS), twice (the first pays the
process's one-off costs); run under `rocprofv3 --kernel-trace --output-format csv` and feed the
trace to tools/tts_summary.py to see where a REAL run's time goes (bench.py `time_to_solution`)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from nbed_amd import synth  # noqa: E402
from nbed_amd.backend import HipBackend  # noqa: E402
from nbed_amd.scf import GpuUHF, Mole, huzinaga_scf  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 148
be = HipBackend(0)
pr = synth.problem(be, N, (33, 33), 20)
eri = be.synth_eri(N)
mf0 = GpuUHF(Mole(N, pr["nelec"]), pr["S"], pr["hcore"], eri, backend=be)
packed = mf0.eri_packed_device()
for rep in range(3):
    mf = GpuUHF(Mole(N, pr["nelec"]), pr["S"], pr["hcore"], eri, backend=be, eri_packed=packed)
    mf.conv_tol, mf.max_cycle = 1e-6, 50
    hist = []
    torch.cuda.synchronize()
    be.zeros(3)  # marker: a fill kernel right before the run
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = huzinaga_scf(mf, pr["V_emb"], pr["D_env"], dm_conv_tol=1e-6, history=hist)
    torch.cuda.synchronize()
    print(f"run {rep}: {1e3 * (time.perf_counter() - t0):.2f} ms, {len(hist)} cycles, converged {out[4]}", flush=True)
    time.sleep(0.05)
