import logging, sys, time
sys.path.insert(0, ".")
import torch
from nbed_amd import synth
from nbed_amd.backend import HipBackend
from nbed_amd.scf import GpuUHF, Mole, huzinaga_scf
import importlib
hs = importlib.import_module("nbed_amd.scf.huzinaga_scf")
logging.basicConfig(level=logging.WARNING)
hs.logger.setLevel(logging.DEBUG)
h = logging.StreamHandler(sys.stdout); hs.logger.addHandler(h)
be = HipBackend()
N = 148
pr = synth.problem(be, N, (33, 33), 20)
mf = GpuUHF(Mole(N, pr["nelec"]), pr["S"], pr["hcore"], be.synth_eri(N), backend=be)
mf.eri_packed_device()
mf.conv_tol = -1.0
mf.max_cycle = 23
hist = []
ts = []
huzinaga_scf(mf, pr["V_emb"], pr["D_env"], history=hist, callback=lambda i: ts.append(time.perf_counter()))
for i, (e, d) in enumerate(hist):
    print(i, e, d, (ts[i + 1] - ts[i]) * 1e6 if i + 1 < len(ts) else None)
