import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from nbed_amd.backend import HipBackend
from nbed_amd import synth
be = HipBackend()
N, n = 148, 128
eri = be.synth_eri(N)
c = synth.sym_matrix(7, N)
ca = be.asarray(np.ascontiguousarray(c[:, :n])); cb = be.asarray(np.ascontiguousarray(c[:, ::-1][:, :n]))
for _ in range(2):
    be.ao2mo_pair(eri, ca, ca, ca, ca, cb, cb); be.ao2mo(eri, cb, cb, cb, cb)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3):
    be.ao2mo_pair(eri, ca, ca, ca, ca, cb, cb); be.ao2mo(eri, cb, cb, cb, cb)
torch.cuda.synchronize(); print("3-block build ms", (time.perf_counter()-t)/3*1e3)
