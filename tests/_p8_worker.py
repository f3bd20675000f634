"""Child process of test_gpu_bench_sizes.py::test_jk_eightfold_experimental_vs_c_oracle: the 8-fold packed
J/K kernel (csrc/jk_p8.hip, opt-in through NBX_JK_P8=1, read once per process) against the C oracle."""
import os
import sys

assert os.environ.get("NBX_JK_P8") == "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from nbed_amd.backend import HipBackend  # noqa: E402
from oracle import cref  # noqa: E402

be = HipBackend()
for n in (148, 104):
    eri = be.synth_eri(n)
    packed = be.eri_pack(eri, n)
    four_fold = n * (n + 1) // 2 * (n * (n + 1) // 2) * 8
    assert packed.numel() * 8 < 0.56 * four_fold, "not the 8-fold format"
    eri_h = be.to_host(eri)
    rng = np.random.default_rng(n)
    for ndm in (2, 1):
        dm = rng.normal(size=(ndm, n, n))
        dm = dm + dm.transpose(0, 2, 1)
        got = be.to_host(be.jk_packed(packed, be.asarray(dm)))
        ref = cref.jk(eri_h, dm)
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert err < 2e-14, (n, ndm, err)
        again = be.to_host(be.jk_packed(packed, be.asarray(dm)))
        assert np.array_equal(got, again), "not reproducible bit for bit"
    if n == 148:
        hv = rng.normal(size=(2, n, n))
        dm_d = be.asarray(dm2 := np.stack([dm[0], dm[0] * 0.5]))
        fock, vhf = be.jk_packed_fock(packed, dm_d, be.asarray(hv))
        jk = be.to_host(be.jk_packed(packed, dm_d))
        assert np.abs(be.to_host(vhf) - (jk[0] - jk[1:])).max() < 1e-12
        assert np.abs(be.to_host(fock) - (hv + jk[0] - jk[1:])).max() < 1e-12
print("P8 OK")
