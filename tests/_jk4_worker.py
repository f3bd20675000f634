"""Child process of tests/test_gpu_kernels.py::test_jk_4fold_walk_kernel_behind_the_switch: with NBX_JK_M8=0 in the
environment (read once per process) the sizes 97 .. 148 run on csrc/jk_m4.hip's 4-fold packed form again; J and K
against the C oracle on three row slabs, the packed size, the Dtot' table of the scalars kernel."""
import sys

import numpy as np

from nbed_amd.backend import HipBackend
from oracle import cref, synth


def main() -> int:
    be = HipBackend()
    for n in (104, 147, 148):
        assert be.lib.nbx_jk_packed_fold(n) == 4, n
        n4 = (n + 3) // 4 * 4
        assert be.lib.nbx_eri_packed_bytes(n, 0, n) == 8 * (n * (n + 1) // 2) * 16 * ((n4 // 4) * (n4 // 4 + 1) // 2) + 256
        eri = be.synth_eri(n)
        dm = np.stack([synth.sym_matrix(534, n), synth.sym_matrix(535, n)])
        packed = be.eri_pack(eri, n)
        del eri
        got = be.to_host(be.jk_packed(packed, be.asarray(dm)))
        for p0, p1 in [(0, 3), (n // 2 - 1, n // 2 + 2), (n - 3, n)]:
            ref = cref.jk(cref.synth_eri(n, p0, p1), dm, p0, p1)
            np.testing.assert_allclose(got[:, p0:p1], ref, rtol=0, atol=1e-11)
        np.testing.assert_array_equal(got[0], got[0].T)
        if n % 4 == 0:  # the table handed over by the scalars kernel: the same Fock matrices, bit for bit
            hv = be.asarray(np.stack([synth.sym_matrix(570, n), synth.sym_matrix(571, n)]))
            zeros = be.asarray(np.zeros((2, n, n)))
            dmd = be.asarray(dm)
            dts = be.jk_dts_new(n)
            f0, v0 = be.jk_packed_fock(packed, dmd, hv)
            be.huz_cycle_scalars_async(hv, None, zeros, zeros, dmd, dmd, dts=dts).get()
            f1, v1 = be.jk_packed_fock(packed, dmd, hv, dts=dts)
            np.testing.assert_array_equal(be.to_host(f1), be.to_host(f0))
            np.testing.assert_array_equal(be.to_host(v1), be.to_host(v0))
        del packed
    print("JK4 OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
