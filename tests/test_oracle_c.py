"""The C oracle (oracle/c/*.c) held to the numpy oracle and to the defining sums: the generator of
the synthetic (pq|rs) bit for bit, the streamed contractions (additive symmetric J/K slab, half
transform of sampled (i, j) pairs) and the dense J/K / four-index transform against einsum."""

import numpy as np
import pytest

from oracle import cref, hamiltonian, synth
from oracle.pyscf_like import get_jk


@pytest.mark.parametrize("n", [1, 2, 7, 13, 24])
def test_c_generator_is_bit_identical_to_numpy(n):
    np.testing.assert_array_equal(cref.synth_eri(n), synth.eri_dense(n))
    if n >= 7:
        np.testing.assert_array_equal(cref.synth_eri(n, 2, n - 1), synth.eri_block(n, 2, n - 1))


@pytest.mark.parametrize("ndm", [1, 2])
def test_c_dense_jk_and_transform_match_einsum(ndm):
    n = 18
    eri = synth.eri_dense(n)
    dm = np.stack([synth.sym_matrix(20 + x, n) for x in range(ndm)])
    vj, vk = get_jk(eri, dm)
    out = cref.jk(eri, dm)
    np.testing.assert_allclose(out[0], vj.sum(0), rtol=0, atol=1e-13)
    np.testing.assert_allclose(out[1:], vk, rtol=0, atol=1e-13)
    slab = cref.jk(eri[3:9], dm, 3, 9)
    np.testing.assert_allclose(slab, out[:, 3:9], rtol=0, atol=1e-14)
    cs = [synth.general_matrix(30 + i, n, d) for i, d in enumerate((4, 5, 3, 6))]
    np.testing.assert_allclose(cref.ao2mo(eri, *cs), hamiltonian.ao2mo_full(eri, *cs), rtol=0, atol=1e-13)


def test_c_streamed_symmetric_jk_slabs_add_up_to_the_definition():
    n = 20
    eri = synth.eri_dense(n)
    dm = np.stack([synth.sym_matrix(8, n), synth.sym_matrix(9, n)])
    vj, vk = get_jk(eri, dm)
    full = cref.jk_synth_sym(n, dm)
    np.testing.assert_allclose(full[0], vj.sum(0), rtol=0, atol=1e-13)
    np.testing.assert_allclose(full[1:], vk, rtol=0, atol=1e-13)
    parts = cref.jk_synth_sym(n, dm, 0, 7) + cref.jk_synth_sym(n, dm, 7, n)
    np.testing.assert_allclose(parts, full, rtol=0, atol=1e-13)
    # a slab holds exactly the pairs (p, q <= p) and their mirror images: J of a one-row slab
    one = cref.jk_synth_sym(n, dm, 5, 6)
    jrow = np.einsum("qrs,rs->q", eri[5, :6], dm.sum(0))
    np.testing.assert_allclose(one[0][5, :6], jrow, rtol=0, atol=1e-13)
    np.testing.assert_allclose(one[0][:6, 5], jrow, rtol=0, atol=1e-13)
    assert np.count_nonzero(one[0]) <= 11


def test_c_half_transform_of_sampled_pairs():
    n, r = 20, 13
    eri = synth.eri_dense(n)
    a, b = synth.general_matrix(5, 3, n), synth.general_matrix(6, 3, n)
    y = cref.half_transform_rs(n, r, a, b)
    np.testing.assert_allclose(y, np.einsum("kp,kq,pqs->ks", a, b, eri[:, :, r, : r + 1]), rtol=0, atol=1e-13)
