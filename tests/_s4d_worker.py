"""Child process of test_gpu_bench_sizes.py::test_jk_lds_dma_experimental_vs_c_oracle: the packed J/K kernel with
its tiles streamed straight into LDS (csrc/jk_s4d.hip, opt-in through NBX_JK_DMA=1|2|3, read once per process)
against the C oracle, and bit for bit against the numbers the production kernel gave the parent."""
import os
import sys

variant = os.environ.get("NBX_JK_DMA")
assert variant in ("0", "1", "2", "3")  # 0: the production kernel, for the bit-for-bit comparison
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from nbed_amd.backend import HipBackend  # noqa: E402
from oracle import cref  # noqa: E402

be = HipBackend()
n = 148
eri = be.synth_eri(n)
packed = be.eri_pack(eri, n)
eri_h = be.to_host(eri)
rng = np.random.default_rng(n)
sums = []
for ndm in (2, 1):
    dm = rng.normal(size=(ndm, n, n))
    dm = dm + dm.transpose(0, 2, 1)
    got = be.to_host(be.jk_packed(packed, be.asarray(dm)))
    ref = cref.jk(eri_h, dm)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-14, (ndm, err)
    assert np.array_equal(got, be.to_host(be.jk_packed(packed, be.asarray(dm)))), "not reproducible bit for bit"
    # a row slab (the multi-GPU shape): additive
    cut = 61
    parts = (be.to_host(be.jk_packed(be.eri_pack(eri[:cut], n, 0, cut), be.asarray(dm), 0, cut))
             + be.to_host(be.jk_packed(be.eri_pack(eri[cut:], n, cut, n), be.asarray(dm), cut, n)))
    assert np.abs(parts - ref).max() / np.abs(ref).max() < 2e-14
    sums.append(got.tobytes().hex()[:64] + "%.17g" % float(np.abs(got).sum()))
print("S4D OK", variant, " ".join(sums))
