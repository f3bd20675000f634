"""GPU parity tests, kernel level: every libnbx entry point (through the C ABI via
ctypes) against the CPU oracle / numpy on the same seeded inputs.

Tolerances (fp64): element-wise ops and GEMMs 1e-12 relative to the operand scale;
eigenvalues / singular values 1e-12 * ||A||; J/K 1e-12 (sums of N^2 products of O(1/N)).
"""

import numpy as np
import pytest

from conftest import load_golden
from oracle import hamiltonian, synth
from oracle.pyscf_like import get_jk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from nbed_amd.backend import HipBackend

    return HipBackend()


def rnd(stream, *shape):
    return synth.val(stream, np.arange(int(np.prod(shape)))).reshape(shape)


def symm(stream, n):
    return synth.sym_matrix(stream, n)


# ---------------------------------------------------------------- synthetic ERI
@pytest.mark.parametrize("n", [1, 2, 7, 12, 13])
def test_synth_eri_bit_exact(be, n):
    got = be.to_host(be.synth_eri(n))
    np.testing.assert_array_equal(got, synth.eri_dense(n))


def test_synth_eri_slab(be):
    n = 11
    got = be.to_host(be.synth_eri(n, 3, 8))
    np.testing.assert_array_equal(got, synth.eri_block(n, 3, 8))


# ---------------------------------------------------------------- J/K
@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 12, 24, 37, 64])
@pytest.mark.parametrize("ndm", [1, 2])
def test_jk_dense_vs_oracle(be, n, ndm):
    eri_h = synth.eri_dense(n)
    dm = np.stack([symm(20 + x, n) for x in range(ndm)])
    vj, vk = get_jk(eri_h, dm)
    out = be.to_host(be.jk(be.asarray(eri_h), be.asarray(dm)))
    np.testing.assert_allclose(out[0], vj.sum(axis=0), rtol=0, atol=1e-12 * n)
    for x in range(ndm):
        np.testing.assert_allclose(out[1 + x], vk[x], rtol=0, atol=1e-12 * n)


def test_jk_dense_slabs_equal_full(be):
    n = 24
    eri = be.synth_eri(n)
    dm = be.asarray(np.stack([symm(31, n), symm(32, n)]))
    full = be.to_host(be.jk(eri, dm))
    for p0, p1 in [(0, 5), (5, 24), (7, 8)]:
        slab = be.to_host(be.jk(eri[p0:p1].contiguous(), dm, p0, p1))
        np.testing.assert_array_equal(slab, full[:, p0:p1])


def test_jk_dense_full_size_properties(be):
    """N = 148 (the bench shape): linearity and symmetry, no CPU reference needed."""
    n = 148
    eri = be.synth_eri(n)
    d1 = np.stack([symm(41, n), symm(42, n)])
    d2 = np.stack([symm(43, n), symm(44, n)])
    j1 = be.to_host(be.jk(eri, be.asarray(d1)))
    j2 = be.to_host(be.jk(eri, be.asarray(d2)))
    j12 = be.to_host(be.jk(eri, be.asarray(d1 + 0.5 * d2)))
    np.testing.assert_allclose(j12, j1 + 0.5 * j2, rtol=0, atol=1e-10)
    for k in range(3):
        np.testing.assert_allclose(j1[k], j1[k].T, rtol=0, atol=1e-10)
    # a handful of elements against direct sums on the host
    eri_rows = be.to_host(eri[5])  # (q, r, s) for p = 5
    np.testing.assert_allclose(j1[0][5, 9], np.sum(eri_rows[9] * (d1[0] + d1[1])), rtol=0, atol=1e-10)
    np.testing.assert_allclose(j1[1][5, 17], np.einsum("qs,qs->", eri_rows[:, 17, :], d1[0]), rtol=0, atol=1e-10)


# ---------------------------------------------------------------- GEMM
@pytest.mark.parametrize("ta", ["N", "T"])
@pytest.mark.parametrize("tb", ["N", "T"])
@pytest.mark.parametrize("mnk", [(1, 1, 1), (7, 5, 3), (16, 16, 4), (33, 65, 17), (148, 148, 148), (128, 300, 148),
                                 (200, 40, 9)])
def test_gemm(be, ta, tb, mnk):
    m, n, k = mnk
    a = rnd(50, *((k, m) if ta == "T" else (m, k)))
    b = rnd(51, *((n, k) if tb == "T" else (k, n)))
    ref = (a.T if ta == "T" else a) @ (b.T if tb == "T" else b)
    got = be.to_host(be.gemm(be.asarray(a), be.asarray(b), ta, tb))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13 * max(k, 1))


def test_gemm_identity_asymmetric(be):
    # A = I with an asymmetric B catches a transposed C write (MFMA C/D lane map)
    n = 48
    b = np.arange(n * n, dtype=np.float64).reshape(n, n)
    got = be.to_host(be.gemm(be.asarray(np.eye(n)), be.asarray(b)))
    np.testing.assert_array_equal(got, b)
    got = be.to_host(be.gemm(be.asarray(b), be.asarray(np.eye(n))))
    np.testing.assert_array_equal(got, b)


def test_gemm_batched_alpha_beta(be):
    a = rnd(52, 3, 20, 30)
    b = rnd(53, 30, 25)  # shared across the batch
    c0 = rnd(54, 3, 20, 25)
    out = be.asarray(c0)
    be.gemm(be.asarray(a), be.asarray(b), alpha=0.5, beta=-2.0, out=out)
    np.testing.assert_allclose(be.to_host(out), 0.5 * (a @ b) - 2.0 * c0, rtol=0, atol=1e-12)


def test_gemm_large_tile_path(be):
    m, n, k = 256, 4096, 148  # 128x128 tile configuration
    a, b = rnd(55, k, m), rnd(56, k, n)
    got = be.to_host(be.gemm(be.asarray(a), be.asarray(b), "T", "N"))
    np.testing.assert_allclose(got, a.T @ b, rtol=0, atol=1e-11)


@pytest.mark.parametrize("mnk", [(256, 4096, 4), (130, 8190, 12), (200, 3000, 148), (128, 70000, 20), (66, 66 * 128, 2000)])
def test_gemm_tn_lds_dma_kernel_edges(be, mnk):
    """gemm_m4_tn_kernel ('T','N' operands, k-tiles by LDS-DMA, v_mfma_f64_4x4x4): a single 4-row step, a tile and a
    half, edge tiles whose clamped loads bring columns nobody stores, a long k -- against numpy; batched with alpha and
    beta; and the shapes it must hand back to the register-staged kernel (odd extents, k not a multiple of 4, k = 0)."""
    m, n, k = mnk
    a, b = rnd(57, k, m), rnd(58, k, n)
    got = be.to_host(be.gemm(be.asarray(a), be.asarray(b), "T", "N"))
    np.testing.assert_allclose(got, a.T @ b, rtol=0, atol=2e-13 * k)
    if k == 148:
        a3, b3, c0 = rnd(59, 5, k, m), rnd(60, 5, k, n), rnd(61, 5, m, n)
        out = be.asarray(c0)
        be.gemm(be.asarray(a3), be.asarray(b3), "T", "N", alpha=0.5, beta=-2.0, out=out)
        np.testing.assert_allclose(be.to_host(out), 0.5 * np.einsum("bkm,bkn->bmn", a3, b3) - 2.0 * c0, rtol=0, atol=1e-11)
        for mm, nn, kk in [(201, 3000, 148), (200, 3001, 148), (200, 3000, 150)]:
            a2, b2 = rnd(62, kk, mm), rnd(63, kk, nn)
            got2 = be.to_host(be.gemm(be.asarray(a2), be.asarray(b2), "T", "N"))
            np.testing.assert_allclose(got2, a2.T @ b2, rtol=0, atol=1e-11)
        c1 = rnd(64, m, n)
        out = be.asarray(c1)
        be.gemm_raw("T", "N", m, n, 0, 1.0, out, m, 0, out, n, 0, 0.5, out, n, 0, 1)  # k = 0: C <- beta C
        np.testing.assert_allclose(be.to_host(out), 0.5 * c1, rtol=0, atol=0)


# ---------------------------------------------------------------- element-wise / reductions
def test_fock_and_huzinaga_and_scalars(be):
    n = 37
    h = symm(60, n)
    v = np.stack([symm(61, n), symm(62, n)]) * 0.1
    jk = rnd(63, 3, n, n)
    fock, vhf = be.fock_uhf(be.asarray(h), be.asarray(v), be.asarray(jk))
    vhf_ref = jk[0] - jk[1:]
    np.testing.assert_allclose(be.to_host(vhf), vhf_ref, rtol=0, atol=1e-15)
    np.testing.assert_allclose(be.to_host(fock), h + v + vhf_ref, rtol=0, atol=1e-14)
    h3 = np.stack([h, h + 0.01])
    fock3, _ = be.fock_uhf(be.asarray(h3), None, be.asarray(jk))
    np.testing.assert_allclose(be.to_host(fock3), h3 + vhf_ref, rtol=0, atol=1e-14)

    fds = rnd(64, 2, n, n)
    f_io = be.asarray(h3)
    hz = be.huzinaga_sym(be.asarray(fds), 1.0, f_io)
    hz_ref = -(fds + fds.transpose(0, 2, 1))
    np.testing.assert_allclose(be.to_host(hz), hz_ref, rtol=0, atol=1e-15)
    np.testing.assert_allclose(be.to_host(f_io), h3 + hz_ref, rtol=0, atol=1e-14)
    hz2 = be.huzinaga_sym(be.asarray(fds[0]), 0.5)
    np.testing.assert_allclose(be.to_host(hz2), -0.5 * (fds[0] + fds[0].T), rtol=0, atol=1e-15)

    a, b = rnd(65, 2, n, n), rnd(66, 2, n, n)
    np.testing.assert_allclose(be.trace_prod(be.asarray(a), be.asarray(b)), np.einsum("xij,xji->x", a, b),
                               rtol=0, atol=1e-12)
    np.testing.assert_allclose(be.trace_prod(be.asarray(a[0]), be.asarray(b[0])), np.einsum("ij,ji->", a[0], b[0]),
                               rtol=0, atol=1e-12)

    dm, dm_old = rnd(67, 2, n, n), rnd(68, 2, n, n)
    sc = be.huz_cycle_scalars(be.asarray(h), be.asarray(v), be.asarray(vhf_ref), be.asarray(hz_ref),
                              be.asarray(dm), be.asarray(dm_old))
    ham = h + v + 0.5 * vhf_ref + hz_ref
    np.testing.assert_allclose(sc[:2], np.einsum("xij,xji->x", ham, dm), rtol=0, atol=1e-11)
    np.testing.assert_allclose(sc[2:], np.linalg.norm(dm - dm_old, axis=(-2, -1)), rtol=0, atol=1e-12)
    # the no-synchronisation variant (one launch, finer tiles, results stored to pinned memory by
    # the kernel): same numbers up to the summation order
    pend = be.huz_cycle_scalars_async(be.asarray(h), be.asarray(v), be.asarray(vhf_ref), be.asarray(hz_ref),
                                      be.asarray(dm), be.asarray(dm_old))
    np.testing.assert_allclose(pend.get(), sc, rtol=1e-13, atol=0)
    again = be.huz_cycle_scalars_async(be.asarray(h), be.asarray(v), be.asarray(vhf_ref), be.asarray(hz_ref),
                                       be.asarray(dm), be.asarray(dm_old))
    np.testing.assert_array_equal(again.get(), pend.get())  # fixed order: reproducible bit for bit


def test_vector_algebra(be):
    n = 1000
    vecs = rnd(70, 5, n)
    x = rnd(71, n)
    np.testing.assert_allclose(be.dots(be.asarray(x), be.asarray(vecs)), vecs @ x, rtol=0, atol=1e-12)
    coef = [0.3, -1.2, 2.0, 0.0, 1e-3]
    np.testing.assert_allclose(be.to_host(be.lincomb(coef, be.asarray(vecs))), np.asarray(coef) @ vecs, rtol=0,
                               atol=1e-14)
    y = be.asarray(x)
    be.axpby(2.0, be.asarray(vecs[0]), -0.5, y)
    np.testing.assert_allclose(be.to_host(y), 2.0 * vecs[0] - 0.5 * x, rtol=0, atol=1e-15)


@pytest.mark.parametrize("case", ["generic", "converging", "wraparound", "settled"])
def test_diis_device_matches_pyscf_semantics(be, case):
    """nbx_diis_update against the oracle's pyscf.lib.diis.DIIS restatement, update by update:
    first call only stores x; slots wrap after `space`; near convergence the Pulay matrix has
    |eigenvalues| < 1e-14 and PySCF's mode-dropping branch decides the coefficients."""
    from nbed_amd.scf.diis import DIIS as DeviceDIIS
    from oracle.pyscf_like import DIIS as OracleDIIS

    n = 2 * 37 * 37
    dev, ref = DeviceDIIS(be), OracleDIIS()
    fixed = rnd(80, n)
    nsteps = {"generic": 5, "converging": 9, "wraparound": 11, "settled": 30}[case]
    for it in range(nsteps):
        if case == "converging":  # geometric approach to a fixed point: errors ~ 1e-1 ... 1e-9
            x = fixed + rnd(81 + it, n) * 10.0 ** (-1 - it)
        elif case == "settled":  # a long run: errors halve down to rounding noise and stay there (the Pulay
            # solve starts every update from the basis the one before ended in: nbx_diis_coef_doubles)
            x = fixed + rnd(81 + it, n) * max(1e-3 * 0.5 ** it, 1e-13)
        else:
            x = fixed + rnd(81 + it, n) * 0.1
        got = be.to_host(dev.update(be.asarray(x)))
        want = ref.update(x)
        scale = np.max(np.abs(want))
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-9 * scale, err_msg=f"update {it}")
    assert dev.get_num_vec() == ref.get_num_vec()


@pytest.mark.parametrize("n,naux,nocc", [(24, 40, (5, 4)), (24, 7, (6,)), (148, 70, (70, 66)), (150, 130, (75, 0))])
def test_density_fitted_jk_vs_numpy_and_dense_contraction(be, n, naux, nocc):
    """nbx_jk_df (an extra: the GEMM-shaped J/K of SURVEY section 7 step 5) against its definition in numpy, the
    generator against the host one, slabs of the auxiliary index against the whole, and -- at the small size -- against
    the dense contraction of the integrals the factor stands for, (pq|rs) = sum_L B_L[pq] B_L[rs], through the oracle."""
    b_h = synth.df_factor(n, 0, naux)
    b_d = be.df_synth(n, 0, naux)
    np.testing.assert_array_equal(be.to_host(b_d), b_h)
    np.testing.assert_array_equal(b_h, b_h.transpose(0, 2, 1))
    ndm = len(nocc)
    c = np.stack([np.linalg.qr(rnd(95 + x, n, n))[0] for x in range(ndm)])
    dms = np.stack([c[x][:, :nocc[x]] @ c[x][:, :nocc[x]].T for x in range(ndm)])
    dtot = dms.sum(0) if ndm == 2 else 2.0 * dms[0]
    j_ref = np.einsum("lpq,l->pq", b_h, np.einsum("lpq,pq->l", b_h, dtot))
    k_ref = np.stack([np.einsum("lpr,rs,lsq->pq", b_h, dms[x], b_h, optimize=True) for x in range(ndm)])
    got = be.to_host(be.jk_df(b_d, be.asarray(c), nocc))
    scale = max(np.abs(j_ref).max(), np.abs(k_ref).max())
    np.testing.assert_allclose(got[0], j_ref, rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(got[1:], k_ref, rtol=0, atol=1e-12 * scale)
    # additive over slabs of the auxiliary basis (the multi-GPU split)
    cut = naux // 3
    parts = be.to_host(be.jk_df(b_d[:cut], be.asarray(c), nocc)) + be.to_host(be.jk_df(b_d[cut:], be.asarray(c), nocc))
    np.testing.assert_allclose(parts, got, rtol=0, atol=1e-12 * scale)
    if n == 24 and ndm == 2:
        eri = np.einsum("lpq,lrs->pqrs", b_h, b_h)
        j_d = np.einsum("pqrs,rs->pq", eri, dtot)
        k_d = np.stack([np.einsum("prqs,rs->pq", eri, dms[x]) for x in range(2)])
        jk_dev = be.to_host(be.jk(be.asarray(eri), be.asarray(dms)))
        np.testing.assert_allclose(jk_dev[0], j_d, rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(got[0], jk_dev[0], rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(got[1:], jk_dev[1:], rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(k_d, jk_dev[1:], rtol=0, atol=1e-12 * scale)


def test_results_gathered_to_host_in_one_launch(be):
    """nbx_gather_to_host behind ``to_host_many`` (the four arrays an SCF run returns): values, shapes, empty and
    non-contiguous members (those take the concatenation path)."""
    import torch

    arrs = [rnd(90, 2, 37, 37), rnd(91, 2, 37), rnd(92, 2, 37, 37), rnd(93, 5)]
    dev = [be.asarray(a) for a in arrs]
    for got, want in zip(be.to_host_many(dev), arrs):
        assert got.shape == want.shape
        np.testing.assert_array_equal(got, want)
    mixed = [dev[0].transpose(1, 2), dev[1], torch.empty(0, dtype=torch.float64, device=dev[0].device)]
    out = be.to_host_many(mixed)
    np.testing.assert_array_equal(out[0], arrs[0].transpose(0, 2, 1))
    np.testing.assert_array_equal(out[1], arrs[1])
    assert out[2].size == 0
    later = be.to_host_many(dev, wait=False)  # returns at once; get() polls the word the kernel stores last
    for got, want in zip(later.get(), arrs):
        np.testing.assert_array_equal(got, want)
    big = be.asarray(rnd(94, 300000))  # larger than the landing buffer the backend starts with
    np.testing.assert_array_equal(be.to_host_many([big, dev[3]])[0], be.to_host(big))


def test_transpose_scale_and_chem_to_phys(be):
    a = rnd(72, 3, 45, 70)
    np.testing.assert_array_equal(be.to_host(be.transpose(be.asarray(a))), a.transpose(0, 2, 1))
    s = rnd(73, 70)
    got = be.to_host(be.scale_cols(be.asarray(a[0]), be.asarray(s)))
    np.testing.assert_allclose(got, a[0] * s, rtol=0, atol=1e-16)
    x = rnd(74, 3, 4, 5, 6)
    np.testing.assert_array_equal(be.to_host(be.chem_to_phys(be.asarray(x))), x.transpose(0, 2, 3, 1))


# ---------------------------------------------------------------- eigh
def check_eigh(be, a, tol=1e-12):
    w, v = be.eigh(be.asarray(a), check=True)
    w, v = be.to_host(w), be.to_host(v)
    scale = max(np.linalg.norm(a, axis=(-2, -1)).max(), 1e-300)
    w_ref = np.linalg.eigvalsh(a)
    np.testing.assert_allclose(w, w_ref, rtol=0, atol=tol * scale)
    av = a @ v
    np.testing.assert_allclose(av, v * w[..., None, :], rtol=0, atol=tol * scale)
    eye = np.eye(a.shape[-1])
    np.testing.assert_allclose(np.swapaxes(v, -1, -2) @ v, np.broadcast_to(eye, a.shape), rtol=0, atol=1e-12)
    assert np.all(np.diff(w, axis=-1) >= 0)
    assert all(s > 0 for s in be.last_eigh_sweeps)
    return w, v


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 24, 37, 100, 148])
def test_eigh_random(be, n):
    check_eigh(be, symm(80, n))


def test_eigh_batched_spin_pair(be):
    check_eigh(be, np.stack([symm(81, 30), symm(82, 30)]))


def test_eigh_degenerate_and_graded(be):
    n = 20
    u = rnd(83, n)
    a = np.eye(n) + np.outer(u, u)  # (n-1)-fold degenerate eigenvalue 1
    check_eigh(be, a)
    # mu-shift-like grading: a 1e6 block next to O(1)
    g = symm(84, n)
    p = np.zeros((n, n))
    p[:3, :3] = 1e6 * (np.eye(3) + 0.1 * symm(85, 3))
    w, v = check_eigh(be, g + p)
    # the O(1) eigenvalues must be accurate in ABSOLUTE terms, not relative to 1e6
    w_ref = np.linalg.eigvalsh(g + p)
    np.testing.assert_allclose(w[: n - 3], w_ref[: n - 3], rtol=0, atol=1e-9)
    check_eigh(be, np.zeros((5, 5)))
    check_eigh(be, np.diag(np.arange(6.0)))


def test_eigh_warm_start(be):
    """Seeding with the eigenvectors of a nearby matrix gives the same decomposition in
    fewer sweeps (what huzinaga_scf does from one SCF cycle to the next)."""
    n = 148
    a = np.stack([symm(86, n), symm(87, n)])
    w0, v0 = be.eigh(be.asarray(a), check=True)
    cold = list(be.last_eigh_sweeps)
    a2 = a + 1e-3 * np.stack([symm(88, n), symm(89, n)])
    w, v = be.eigh(be.asarray(a2), check=True, v0=v0)
    warm = list(be.last_eigh_sweeps)
    w, v = be.to_host(w), be.to_host(v)
    np.testing.assert_allclose(w, np.linalg.eigvalsh(a2), rtol=0, atol=1e-11)
    np.testing.assert_allclose(a2 @ v, v * w[:, None, :], rtol=0, atol=1e-10)
    np.testing.assert_allclose(np.swapaxes(v, -1, -2) @ v, np.broadcast_to(np.eye(n), a.shape), rtol=0, atol=1e-12)
    # cold: Jacobi sweep count, or 1 when the tridiagonal pipeline delivered (cold starts, N >= 64);
    # warm: accepted by refinement (>= 1000) or a few sweeps on the nearly diagonal V0^T A V0
    assert all(c_ >= 1 for c_ in cold) and all(w_ >= 1000 or w_ <= 4 for w_ in warm), (warm, cold)


def eig_quality(a, w, v):
    n = a.shape[-1]
    res = np.max(np.abs(a @ v - v * w[..., None, :]))
    orth = np.max(np.abs(np.swapaxes(v, -1, -2) @ v - np.eye(n)))
    return res, orth


@pytest.mark.parametrize("n", [12, 37, 148, 196, 230, 300])
@pytest.mark.parametrize("eps", [1e-9, 1e-6, 1e-4])
def test_eigh_warm_refinement_accepts_small_perturbations(be, n, eps):
    """Warm start from the eigenvectors of a nearby matrix: the GEMM refinement (status >= 1000)
    must deliver the same quality as the Jacobi path, sorted ascending."""
    a = np.stack([symm(300 + n, n), symm(301 + n, n)])
    _, v0 = be.eigh(be.asarray(a))
    a2 = a + eps * np.stack([symm(302 + n, n), symm(303 + n, n)])
    w, v = be.eigh(be.asarray(a2), check=True, v0=v0)
    st = list(be.last_eigh_sweeps)
    w, v = be.to_host(w), be.to_host(v)
    assert np.all(np.diff(w, axis=-1) >= 0)
    np.testing.assert_allclose(w, np.linalg.eigvalsh(a2), rtol=0, atol=1e-12 * n)
    res, orth = eig_quality(a2, w, v)
    assert res < 2e-13 * n and orth < 1e-13 * n, (res, orth, st)
    if eps <= 1e-6:
        assert all(x >= 1000 for x in st), st  # refinement, not sweeps


def test_eigh_warm_refinement_falls_back(be):
    """Cases the refinement must hand to Jacobi (decided on the device): a perturbation outside
    the contracting regime, a garbage (non-orthogonal) start, and coupled near-degenerate
    eigenvalues; plus eigenvalue crossings, which it must re-sort."""
    n = 64
    a = symm(320, n)
    w0, v0 = be.eigh(be.asarray(a))
    v0_h = be.to_host(v0)
    # (1) large perturbation
    a_big = a + 0.5 * symm(321, n)
    w, v = be.eigh(be.asarray(a_big), check=True, v0=v0)
    assert be.last_eigh_sweeps[0] < 1000
    res, orth = eig_quality(a_big, be.to_host(w), be.to_host(v))
    assert res < 1e-11 and orth < 1e-12
    np.testing.assert_allclose(be.to_host(w), np.linalg.eigvalsh(a_big), rtol=0, atol=1e-11)
    # (2) degenerate pair split and mixed by the perturbation: d = diag(.., 1, 1, ..), coupling 1e-7
    d = np.linspace(-3.0, 3.0, n)
    d[10] = d[11] = 0.123
    q, _ = np.linalg.qr(rnd(322, n, n))
    a_deg = (q * d) @ q.T
    a_deg = 0.5 * (a_deg + a_deg.T)
    pert = np.zeros((n, n))
    pert[10, 11] = pert[11, 10] = 1e-7
    a_deg2 = a_deg + q @ pert @ q.T
    a_deg2 = 0.5 * (a_deg2 + a_deg2.T)
    w, v = be.eigh(be.asarray(a_deg2), check=True, v0=be.asarray(q))
    wh, vh = be.to_host(w), be.to_host(v)
    res, orth = eig_quality(a_deg2, wh, vh)
    assert res < 1e-12 and orth < 1e-12, (res, orth, be.last_eigh_sweeps)
    np.testing.assert_allclose(wh, np.linalg.eigvalsh(a_deg2), rtol=0, atol=1e-12)
    # (3) crossing: start vectors ordered for `a`, matrix has two levels swapped
    d2 = np.linspace(-1.0, 1.0, n)
    a_x = (q * d2) @ q.T
    d3 = d2.copy()
    d3[[20, 21]] = d3[[21, 20]] + np.array([1e-3, -1e-3])
    a_x2 = (q * d3) @ q.T
    a_x2 = 0.5 * (a_x2 + a_x2.T)
    w, v = be.eigh(be.asarray(a_x2), check=True, v0=be.asarray(q))
    wh, vh = be.to_host(w), be.to_host(v)
    assert np.all(np.diff(wh) >= 0)
    res, orth = eig_quality(a_x2, wh, vh)
    assert res < 1e-12 and orth < 1e-12
    np.testing.assert_allclose(wh, np.sort(d3), rtol=0, atol=1e-12)
    # (4) garbage start: not orthonormal at all
    w, v = be.eigh(be.asarray(a), check=True, v0=be.asarray(rnd(323, n, n)))
    assert be.last_eigh_sweeps[0] < 1000
    # (5) N > 196 (fallback = the tridiagonal pipeline, chosen after reading the status back)
    n2 = 230
    a2 = symm(324, n2)
    _, v2 = be.eigh(be.asarray(a2))
    a2b = a2 + 0.5 * symm(325, n2)
    w, v = be.eigh(be.asarray(a2b), check=True, v0=v2)
    assert be.last_eigh_sweeps[0] < 1000
    res, orth = eig_quality(a2b, be.to_host(w), be.to_host(v))
    assert res < 1e-10 and orth < 1e-11, (res, orth)
    np.testing.assert_allclose(be.to_host(w), np.linalg.eigvalsh(a2b), rtol=0, atol=1e-10)
    # queued-iteration knob: results do not depend on it
    for iters in (0, 1, 2):
        w_i, v_i = be.eigh(be.asarray(a_big), v0=v0, refine_iters=iters)
        np.testing.assert_allclose(be.to_host(w_i), np.linalg.eigvalsh(a_big), rtol=0, atol=1e-11)


@pytest.mark.parametrize("n", [196, 197, 230, 431])
def test_eigh_lds_boundary_and_tridiagonal_path(be, n):
    check_eigh(be, symm(90, n))


def test_eigh_tridiagonal_path_hard_cases(be):
    """N > 196 (Householder + multisection + inverse iteration, Jacobi polish when clustered)."""
    n = 230
    u = rnd(83, n)
    check_eigh(be, np.eye(n) + np.outer(u, u))          # (n-1)-fold degenerate eigenvalue
    g = symm(84, n)
    p = np.zeros((n, n))
    p[:5, :5] = 1e6 * (np.eye(5) + 0.1 * symm(85, 5))    # mu-shift-like grading
    w, v = check_eigh(be, g + p)
    np.testing.assert_allclose(w[: n - 5], np.linalg.eigvalsh(g + p)[: n - 5], rtol=0, atol=1e-8)
    check_eigh(be, np.diag(np.arange(float(n))))          # already diagonal: every reflector is trivial
    check_eigh(be, np.stack([symm(86, n), symm(87, n) + np.eye(n)]))  # batched


@pytest.mark.parametrize("n,batch", [(199, 2), (257, 2), (300, 1), (384, 2), (520, 2), (700, 1), (1000, 2), (1100, 2), (2000, 1)])
def test_eigh_whole_chip_reduction_every_instance(be, n, batch):
    """csrc/eigh_grid.hip (N > 198): the Householder reduction with the matrix in the LDS of up to 256 workgroups, one
    grid-wide hand-over per step, and the compact-WY back-transformation -- every instance of the kernel (2, 4 or 8
    registers per lane and vector; two matrices side by side up to N = 1024, one after the other above; sizes that
    are not multiples of 64; workgroups with a different number of rows) against numpy's eigenvalues, with residual
    and orthonormality at rounding level.  NBX_TRIDIAG_GRID=0 would select the one-workgroup kernel instead."""
    a = np.stack([symm(90 + x, n) for x in range(batch)])
    a = a[0] if batch == 1 else a
    w, v = be.eigh(be.asarray(a), check=True)
    w, v = be.to_host(w), be.to_host(v)
    scale = np.linalg.norm(a, axis=(-2, -1)).max()
    np.testing.assert_allclose(w, np.linalg.eigvalsh(a), rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(a @ v, v * w[..., None, :], rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(np.swapaxes(v, -1, -2) @ v, np.broadcast_to(np.eye(n), a.shape), rtol=0, atol=1e-12)


def test_eigh_whole_chip_reduction_is_reproducible(be):
    """The hand-over words of csrc/eigh_grid.hip carry a (launch, step) tag mixed with their value and are polled by
    their readers -- no grid-wide barrier orders the workgroups.  Six runs of the same batched problem (N = 1000: two
    matrices side by side, 256 workgroups, 999 steps) must give the same bits every time and the right spectrum."""
    n = 1000
    a_h = np.stack([symm(90, n), symm(91, n)])
    a = be.asarray(a_h)
    w_ref = np.stack([np.linalg.eigvalsh(a_h[x]) for x in range(2)])
    first = None
    for _ in range(6):
        w, v = be.eigh(a, check=True)
        w, v = be.to_host(w), be.to_host(v)
        np.testing.assert_allclose(w, w_ref, rtol=0, atol=1e-11)
        if first is None:
            first = (w, v)
        else:
            np.testing.assert_array_equal(w, first[0])
            np.testing.assert_array_equal(v, first[1])


def test_eigh_whole_chip_reduction_hard_cases(be):
    """The same path on spectra that break naive reductions: an (n-1)-fold degenerate eigenvalue (reflectors with
    tau = 0 almost everywhere after the first), a matrix that is already diagonal, a 1e6 block beside O(1) entries
    (the mu-shifted Fock matrix of nbed/driver.py:518), and zero rows."""
    n = 520
    u = rnd(83, n)
    check_eigh(be, np.eye(n) + np.outer(u, u))
    check_eigh(be, np.diag(np.arange(float(n))))
    g = symm(84, n)
    p = np.zeros((n, n))
    p[:7, :7] = 1e6 * (np.eye(7) + 0.1 * symm(85, 7))
    w, v = check_eigh(be, g + p)
    np.testing.assert_allclose(w[: n - 7], np.linalg.eigvalsh(g + p)[: n - 7], rtol=0, atol=1e-8)
    z = symm(86, n)
    z[100:140] = 0.0
    z[:, 100:140] = 0.0
    check_eigh(be, z)


@pytest.mark.parametrize("p", [-0.5, 0.5, -1.0])
def test_sym_pow(be, p):
    n = 37
    s = synth.overlap(n)
    w, u = np.linalg.eigh(s)
    ref = (u * w**p) @ u.T
    got = be.to_host(be.sym_pow(be.asarray(s), p))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


# ---------------------------------------------------------------- SVD
@pytest.mark.parametrize("shape", [(1, 1), (6, 5), (5, 8), (33, 33), (20, 64), (64, 20), (7, 3), (148, 33), (12, 41)])
def test_svd_right(be, shape):
    m, n = shape
    a = rnd(90, m, n)
    s, vt = be.svd_right(be.asarray(a))
    s, vt = be.to_host(s), be.to_host(vt)
    s_ref = np.linalg.svd(a, compute_uv=False)
    np.testing.assert_allclose(s, s_ref, rtol=0, atol=1e-13 * s_ref[0] * max(m, n))
    np.testing.assert_allclose(vt @ vt.T, np.eye(n), rtol=0, atol=1e-12)
    g = a @ vt.T  # columns orthogonal, norms = singular values (then zeros)
    norms = np.linalg.norm(g, axis=0)
    np.testing.assert_allclose(norms[: len(s)], s, rtol=0, atol=1e-12 * s_ref[0])
    np.testing.assert_allclose(norms[len(s):], 0, atol=1e-12 * s_ref[0])
    gram = g.T @ g
    np.testing.assert_allclose(gram - np.diag(np.diag(gram)), 0, atol=1e-12 * s_ref[0] ** 2)
    assert be.last_svd_sweeps > 0


def test_svd_rank_deficient_tiny_values(be):
    # exactly rank-3 matrix: the remaining singular values must come out ~1e-16, not ~1e-8
    m, n = 20, 12
    a = rnd(91, m, 3) @ rnd(92, 3, n)
    s, vt = be.svd_right(be.asarray(a))
    s = be.to_host(s)
    assert np.all(s[3:] < 1e-13 * s[0])
    np.testing.assert_allclose(s[:3], np.linalg.svd(a, compute_uv=False)[:3], rtol=1e-12)


# ---------------------------------------------------------------- four-index transform
@pytest.mark.parametrize("n,dims", [(12, (5, 6, 7, 4)), (13, (3, 3, 3, 3)), (24, (20, 20, 20, 20))])
def test_ao2mo_vs_oracle(be, n, dims):
    eri_h = synth.eri_dense(n)
    cs = [rnd(100 + i, n, d) for i, d in enumerate(dims)]
    ref = hamiltonian.ao2mo_full(eri_h, *cs)
    got = be.to_host(be.ao2mo(be.asarray(eri_h), *[be.asarray(c) for c in cs]))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    # outer-index slabs (the multi-GPU shard axis) reproduce the full tensor bit for bit
    lo = be.to_host(be.ao2mo(be.asarray(eri_h), *[be.asarray(c) for c in cs], i0=0, i1=2))
    hi = be.to_host(be.ao2mo(be.asarray(eri_h), *[be.asarray(c) for c in cs], i0=2, i1=dims[0]))
    np.testing.assert_array_equal(np.concatenate([lo, hi]), got)


# ---------------------------------------------------------------- streamed (generated) ERI path
@pytest.mark.parametrize("n", [7, 12, 37, 64])
@pytest.mark.parametrize("ndm", [1, 2])
def test_jk_synth_equals_dense(be, n, ndm):
    """J/K with the integrals generated in registers == J/K on the materialised tensor, bit for bit
    in the values read (same hash), so only summation order could differ: it does not."""
    dm = be.asarray(np.stack([symm(20 + x, n) for x in range(ndm)]))
    dense = be.to_host(be.jk(be.synth_eri(n), dm))
    streamed = be.to_host(be.jk_synth(n, dm))
    np.testing.assert_array_equal(streamed, dense)
    slab = be.to_host(be.jk_synth(n, dm, 2, n - 1))
    np.testing.assert_array_equal(slab, dense[:, 2 : n - 1])


@pytest.mark.parametrize("n,dims", [(12, (5, 6, 7, 4)), (24, (9, 9, 8, 8))])
def test_ao2mo_synth_vs_oracle(be, n, dims):
    cs = [rnd(100 + i, n, d) for i, d in enumerate(dims)]
    ref = hamiltonian.ao2mo_full(synth.eri_dense(n), *cs)
    dcs = [be.asarray(c) for c in cs]
    got = be.to_host(be.ao2mo_synth(n, *dcs))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    # r shards are partial sums (the multi-GPU axis of the streamed path)
    part = be.to_host(be.ao2mo_synth(n, *dcs, r0=0, r1=5)) + be.to_host(be.ao2mo_synth(n, *dcs, r0=5, r1=n))
    np.testing.assert_allclose(part, ref, rtol=0, atol=1e-12)


def test_spinorb_scatter_golden(be):
    g = load_golden("spinorb_from_spatial")
    h1, h2 = be.spinorb_scatter(be.asarray(g["one_body"]), be.asarray(g["two_body"]), 1e-8, 1.0)
    np.testing.assert_array_equal(be.to_host(h1), g["h1"])
    np.testing.assert_array_equal(be.to_host(h2), g["h2"])
    h1, h2 = be.spinorb_scatter(be.asarray(g["one_body"]), be.asarray(g["two_body"]), 1e-8, 0.5)
    np.testing.assert_array_equal(be.to_host(h2), 0.5 * g["h2"])


def test_ao2mo_pair_equals_two_transforms_bitwise(be):
    """(aa|aa) and (aa|bb) in one pass (quarters 1-2 shared) are bit for bit the two separate
    transforms, full and per outer-index slab (the multi-GPU shard axis)."""
    n, na = 24, 14
    eri = be.synth_eri(n)
    ca, cb = be.asarray(rnd(410, n, na)), be.asarray(rnd(411, n, na))
    aa = be.to_host(be.ao2mo(eri, ca, ca, ca, ca))
    ab = be.to_host(be.ao2mo(eri, ca, ca, cb, cb))
    p1, p2 = be.ao2mo_pair(eri, ca, ca, ca, ca, cb, cb)
    np.testing.assert_array_equal(be.to_host(p1), aa)
    np.testing.assert_array_equal(be.to_host(p2), ab)
    s1, s2 = be.ao2mo_pair(eri, ca, ca, ca, ca, cb, cb, i0=3, i1=9)
    np.testing.assert_array_equal(be.to_host(s1), aa[3:9])
    np.testing.assert_array_equal(be.to_host(s2), ab[3:9])
    ref = hamiltonian.ao2mo_full(synth.eri_dense(n), be.to_host(ca), be.to_host(ca), be.to_host(cb), be.to_host(cb))
    np.testing.assert_allclose(ab, ref, rtol=0, atol=1e-12)


@pytest.mark.parametrize("n,na,nb_", [(12, 5, 4), (24, 14, 14), (37, 20, 9)])
def test_ao2mo_pair_sym_vs_oracle(be, n, na, nb_):
    """(ij|kl) = (ji|kl) transform (quarters 3-4 on the pairs j <= i): against the einsum definition
    and the unsymmetrised kernels, single tensor and pair; outputs exactly symmetric in (i, j)."""
    eri_h = synth.eri_dense(n)
    ca_h, cb_h = rnd(420, n, na), rnd(421, n, nb_)
    eri, ca, cb = be.asarray(eri_h), be.asarray(ca_h), be.asarray(cb_h)
    ref_aa = hamiltonian.ao2mo_full(eri_h, ca_h, ca_h, ca_h, ca_h)
    ref_ab = hamiltonian.ao2mo_full(eri_h, ca_h, ca_h, cb_h, cb_h)
    one = be.to_host(be.ao2mo_pair_sym(eri, ca, ca, ca))
    np.testing.assert_allclose(one, ref_aa, rtol=0, atol=1e-12)
    p1, p2 = be.ao2mo_pair_sym(eri, ca, ca, ca, cb, cb)
    p1, p2 = be.to_host(p1), be.to_host(p2)
    np.testing.assert_array_equal(p1, one)
    np.testing.assert_allclose(p2, ref_ab, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(p1, p1.transpose(1, 0, 2, 3))
    np.testing.assert_array_equal(p2, p2.transpose(1, 0, 2, 3))
    np.testing.assert_allclose(p2, be.to_host(be.ao2mo(eri, ca, ca, cb, cb)), rtol=0, atol=1e-13)
    # ... and with (pq|rs) = (pq|sr) too: quarters 1-2 on the packed (r, s <= r) columns
    eri_rs = be.eri_pack_rs(eri, n)
    ir, js = np.tril_indices(n)
    np.testing.assert_array_equal(be.to_host(eri_rs), eri_h[:, :, ir, js])
    q1, q2 = be.ao2mo_pair_sym(eri_rs, ca, ca, ca, cb, cb, rs_packed=True)
    np.testing.assert_allclose(be.to_host(q1), ref_aa, rtol=0, atol=1e-12)
    np.testing.assert_allclose(be.to_host(q2), ref_ab, rtol=0, atol=1e-12)
    np.testing.assert_allclose(be.to_host(be.ao2mo_pair_sym(eri_rs, cb, cb, cb, rs_packed=True)),
                               hamiltonian.ao2mo_full(eri_h, cb_h, cb_h, cb_h, cb_h), rtol=0, atol=1e-12)


def test_ao2mo_synth_pair_equals_two_streamed_transforms(be):
    n, na = 20, 9
    ca, cb = be.asarray(rnd(412, n, na)), be.asarray(rnd(413, n, na))
    aa = be.to_host(be.ao2mo_synth(n, ca, ca, ca, ca, r0=2, r1=17))
    ab = be.to_host(be.ao2mo_synth(n, ca, ca, cb, cb, r0=2, r1=17))
    p1, p2 = be.ao2mo_synth_pair(n, ca, ca, ca, ca, cb, cb, r0=2, r1=17)
    np.testing.assert_array_equal(be.to_host(p1), aa)
    np.testing.assert_array_equal(be.to_host(p2), ab)


@pytest.mark.parametrize("n", [2, 4, 8, 12, 24, 38, 64])
@pytest.mark.parametrize("ndm", [1, 2])
def test_jk_sym_vs_oracle(be, n, ndm):
    """Symmetric J/K (only the tiles q <= p are read) against the einsum definition; slabs add up."""
    eri_h = synth.eri_dense(n)
    dm = np.stack([symm(420 + x, n) for x in range(ndm)])
    vj, vk = get_jk(eri_h, dm)
    eri = be.asarray(eri_h)
    got = be.to_host(be.jk_sym(eri, be.asarray(dm)))
    np.testing.assert_allclose(got[0], vj.sum(0) if vj.ndim == 3 else vj, rtol=0, atol=1e-12)
    np.testing.assert_allclose(got[1:], vk.reshape(ndm, n, n), rtol=0, atol=1e-12)
    if n >= 4:
        cut = n // 3 + 1
        parts = (be.to_host(be.jk_sym(eri[:cut], be.asarray(dm), 0, cut))
                 + be.to_host(be.jk_sym(eri[cut:], be.asarray(dm), cut, n)))
        np.testing.assert_allclose(parts, got, rtol=0, atol=1e-12)


def test_jk_sym_fallback_sizes_and_full_size(be):
    """Odd N goes through the plain kernel inside; N = 148 (the bench size) against the plain kernel."""
    n = 13
    eri_h = synth.eri_dense(n)
    dm = np.stack([symm(430, n), symm(431, n)])
    vj, vk = get_jk(eri_h, dm)
    got = be.to_host(be.jk_sym(be.asarray(eri_h), be.asarray(dm)))
    np.testing.assert_allclose(got[0], vj.sum(0), rtol=0, atol=1e-12)
    np.testing.assert_allclose(got[1:], vk, rtol=0, atol=1e-12)
    n = 148
    eri = be.synth_eri(n)
    dmd = be.asarray(np.stack([symm(432, n), symm(433, n)]))
    a, b = be.to_host(be.jk_sym(eri, dmd)), be.to_host(be.jk(eri, dmd))
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-11)
    np.testing.assert_array_equal(a[0], a[0].T)  # J is written once per pair: exactly symmetric


@pytest.mark.parametrize("n", [24, 148, 200])
def test_geig_refine_tracks_perturbed_pencil(be, n):
    """nbx_geig_refine: eigenpairs of (F, S) refined from the solution of a nearby pencil (what the
    SCF loop hands it) against scipy.linalg.eigh; S-orthonormal output; statuses."""
    import scipy.linalg

    s_h = synth.overlap(n)
    f0 = np.stack([symm(540, n) - np.diag(np.arange(n) * 0.7), symm(541, n) - np.diag(np.arange(n) * 0.9)])
    c0 = np.stack([scipy.linalg.eigh(f0[x], s_h)[1] for x in range(2)])
    # (the perturbation has to stay below the level spacing: a cluster threshold omega of the size
    # of the gaps would -- correctly -- refuse the matrix)
    f1 = f0 + (1e-3 if n < 100 else 2e-5) * np.stack([symm(542, n), symm(543, n)])
    s_b = be.asarray(np.stack([s_h, s_h]))
    w, c = be.geig_refine(be.asarray(f1), s_b, be.asarray(c0), refine_iters=3)
    st = be.to_host(be.last_eigh_status_d)
    assert np.all(st > 1000), st
    w, c = be.to_host(w), be.to_host(c)
    for x in range(2):
        we, _ = scipy.linalg.eigh(f1[x], s_h)
        np.testing.assert_allclose(w[x], we, rtol=0, atol=1e-11 * n)
        np.testing.assert_allclose(c[x].T @ s_h @ c[x], np.eye(n), rtol=0, atol=1e-12)
        np.testing.assert_allclose(f1[x] @ c[x], s_h @ c[x] * w[x], rtol=0, atol=1e-10 * n)
    # one iteration queued: no sorting pass (the update GEMM writes the result, the E kernel the
    # eigenvalues and the verdict); a tiny perturbation is accepted ...
    f2 = f0 + 1e-9 * np.stack([symm(545, n), symm(546, n)])
    w1, c1 = be.geig_refine(be.asarray(f2), s_b, be.asarray(c0), refine_iters=1)
    assert np.all(be.to_host(be.last_eigh_status_d) == 1001)
    w1, c1 = be.to_host(w1), be.to_host(c1)
    for x in range(2):
        np.testing.assert_allclose(w1[x], scipy.linalg.eigh(f2[x], s_h)[0], rtol=0, atol=1e-11 * n)
        np.testing.assert_allclose(c1[x].T @ s_h @ c1[x], np.eye(n), rtol=0, atol=1e-12)
        np.testing.assert_allclose(f2[x] @ c1[x], s_h @ c1[x] * w1[x], rtol=0, atol=1e-10 * n)
    # ... a start whose columns are not in ascending order of eigenvalue is refused (-3)
    swapped = c0.copy()
    swapped[:, :, [2, 3]] = swapped[:, :, [3, 2]]
    be.geig_refine(be.asarray(f2), s_b, be.asarray(swapped), refine_iters=1)
    assert np.all(be.to_host(be.last_eigh_status_d) == -3)
    # a start that is nowhere near: refused (status <= 0), never a silent wrong answer
    q = np.linalg.qr(rnd(544, n, n))[0]
    x_h = np.linalg.inv(scipy.linalg.sqrtm(s_h).real)
    bad = np.stack([x_h @ q, x_h @ q])
    be.geig_refine(be.asarray(f1), s_b, be.asarray(bad), refine_iters=3)
    assert np.all(be.to_host(be.last_eigh_status_d) <= 0)


@pytest.mark.parametrize("n", [7, 16, 24, 148])
def test_huzinaga_fused_matches_two_kernel_path(be, n):
    """nbx_huzinaga_fused (product + symmetrisation + F += Hz in one launch) against numpy and the
    GEMM + nbx_huzinaga_sym pair it replaces; Hz comes out exactly symmetric."""
    f = np.stack([symm(560, n), symm(561, n)])
    ds = rnd(562, 2, n, n)
    for kappa in (1.0, 0.5):
        hz, fo = be.huzinaga_fused(be.asarray(f), be.asarray(ds), kappa)
        fds = np.einsum("xij,xjk->xik", f, ds)
        ref = -kappa * (fds + fds.transpose(0, 2, 1))
        np.testing.assert_allclose(be.to_host(hz), ref, rtol=0, atol=1e-12 * n)
        np.testing.assert_allclose(be.to_host(fo), f + ref, rtol=0, atol=1e-12 * n)
        hz_h = be.to_host(hz)
        np.testing.assert_array_equal(hz_h, hz_h.transpose(0, 2, 1))
        old = be.to_host(be.huzinaga_sym(be.gemm(be.asarray(f), be.asarray(ds)), kappa))
        np.testing.assert_allclose(hz_h, old, rtol=0, atol=1e-12 * n)
    hz1, fo1 = be.huzinaga_fused(be.asarray(f[0]), be.asarray(ds[0]), 1.0)  # restricted (2-D) form
    np.testing.assert_allclose(be.to_host(hz1), -(f[0] @ ds[0] + (f[0] @ ds[0]).T), rtol=0, atol=1e-12 * n)


def s4_layout(n):
    """The packed tile format of nbx_eri_pack (include/nbx.h), restated in numpy: offsets of the
    entries (a, b <= a) of one tile and the tile length."""
    def supported(nb):
        s = n // nb
        ls = s | 1
        e0 = (nb * s * (s + 1) // 2 + 1) & ~1
        er = ((nb // 2) * s * ls + 1) & ~1
        return n % nb == 0 and e0 >= 128 and er >= 128, s, ls, e0, er
    ok4, *_ = supported(4)
    nb = 4 if n % 4 == 0 and ok4 else 2
    _, s, ls, e0, er = supported(nb)
    tri = s * (s + 1) // 2
    off = np.full((n, n), -1, dtype=np.int64)
    for a in range(n):
        for b in range(a + 1):
            i, j, ai, bi = a // s, b // s, a % s, b % s
            if i == j:
                off[a, b] = i * tri + ai * (ai + 1) // 2 + bi
            else:
                r = i ^ j
                hb = r.bit_length() - 1
                slot = ((j >> (hb + 1)) << hb) | (j & ((1 << hb) - 1))
                if r == 3:  # ordered by row block: (2, 1) before (3, 0)
                    slot = 1 - j
                off[a, b] = e0 + (r - 1) * er + slot * s * ls + ai * ls + bi
    return off, e0 + (nb - 1) * er


@pytest.mark.parametrize("n", [24, 32, 50, 72])
def test_eri_pack_layout(be, n):
    """nbx_eri_pack against the numpy restatement of the documented format: bit-identical values at
    the documented offsets, zero pads, tiles q <= p in sequence, slabs start at T(p0, 0)."""
    assert be.jk_packed_supported(n)
    eri_h = synth.eri_dense(n)
    off, m = s4_layout(n)
    want = np.zeros((n * (n + 1) // 2, m))
    ia, ib = np.tril_indices(n)
    t = 0
    for p in range(n):
        for q in range(p + 1):
            want[t, off[ia, ib]] = eri_h[p, q][ia, ib]
            t += 1
    got = be.to_host(be.eri_pack(be.asarray(eri_h), n))
    np.testing.assert_array_equal(got.reshape(-1, m), want)
    p0, p1 = n // 3, n // 3 + 5
    slab = be.to_host(be.eri_pack(be.asarray(eri_h[p0:p1]), n, p0, p1))
    np.testing.assert_array_equal(slab.reshape(-1, m), want[p0 * (p0 + 1) // 2 : p1 * (p1 + 1) // 2])


@pytest.mark.parametrize("n,ndm", [(24, 2), (24, 1), (32, 2), (50, 2), (72, 2), (72, 1), (98, 2)])
def test_jk_packed_vs_oracle(be, n, ndm):
    """Packed J/K (q <= p and s <= r read: a quarter of the tensor) against the einsum definition;
    slabs add up; the result is reproducible bit for bit."""
    eri_h = synth.eri_dense(n)
    dm = np.stack([symm(520 + x, n) for x in range(ndm)])
    vj, vk = get_jk(eri_h, dm)
    eri = be.asarray(eri_h)
    packed = be.eri_pack(eri, n)
    got = be.to_host(be.jk_packed(packed, be.asarray(dm)))
    np.testing.assert_allclose(got[0], vj.sum(0) if vj.ndim == 3 else vj, rtol=0, atol=2e-12)
    np.testing.assert_allclose(got[1:], vk.reshape(ndm, n, n), rtol=0, atol=2e-12)
    np.testing.assert_array_equal(got[0], got[0].T)
    np.testing.assert_array_equal(be.to_host(be.jk_packed(packed, be.asarray(dm))), got)
    cut = n // 3 + 1
    parts = (be.to_host(be.jk_packed(be.eri_pack(eri[:cut], n, 0, cut), be.asarray(dm), 0, cut))
             + be.to_host(be.jk_packed(be.eri_pack(eri[cut:], n, cut, n), be.asarray(dm), cut, n)))
    np.testing.assert_allclose(parts, got, rtol=0, atol=2e-12)
    # empty slab: contributes zeros
    assert float(be.jk_packed(be.eri_pack(eri[:0], n, 3, 3), be.asarray(dm), 3, 3).abs().max()) == 0.0


@pytest.mark.parametrize("n", [100, 104, 116, 118, 124, 128, 130, 132, 140, 144, 148, 152, 192, 256])
def test_jk_packed_full_size(be, n):
    """The kernel instances of the larger sizes against the plain streaming kernel on the generated tensor, both
    spins: N = 100 .. 148 in steps of four are jk_m8.hip's (the 8-fold packed form, one instance per size; 148 is the bench
    size; the sizes here cover its chunk sizes LP = 3 .. 6 and the three-chunk instance N = 104; 118 and 130 run as 120 and
    132 with zero rows and columns), the others jk_s4.hip's."""
    eri = be.synth_eri(n)
    dmd = be.asarray(np.stack([symm(532, n), symm(533, n)]))
    a = be.to_host(be.jk_packed(be.eri_pack(eri, n), dmd))
    b = be.to_host(be.jk(eri, dmd))
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-11 * (n / 148) ** 2)
    np.testing.assert_array_equal(a[0], a[0].T)


@pytest.mark.parametrize("n", [24, 72, 104, 128, 136, 148])
def test_jk_packed_fock_and_prepared_dtot_table(be, n):
    """nbx_jk_packed_fock (Fock assembly in the reduction) against J/K + nbx_fock_uhf, and the Dtot'
    table left by the scalars kernel (nbx_huz_cycle_scalars_dts) against the one the build makes
    itself: bit-identical Fock matrices, also when the table is reused for a second density.
    (N = 100 .. 148 in steps of four are served by jk_m8.hip -- jk_m4.hip behind NBX_JK_M8=0 --, whose table has jk_m4.hip's order with
    other chunk boundaries: csrc/jk_m4_layout.h, nbx_jk_m8_weight_layout.)"""
    eri = be.synth_eri(n)
    packed = be.eri_pack(eri, n)
    hv = be.asarray(np.stack([symm(570, n), symm(571, n)]))
    zeros = be.asarray(np.zeros((2, n, n)))
    dts = be.jk_dts_new(n)
    for seed in (572, 574):
        dm = be.asarray(np.stack([symm(seed, n), symm(seed + 1, n)]))
        fock0, vhf0 = be.jk_packed_fock(packed, dm, hv)
        jk = be.jk_packed(packed, dm)
        f_ref, v_ref = be.fock_uhf(hv, None, jk)
        np.testing.assert_array_equal(be.to_host(fock0), be.to_host(f_ref))
        np.testing.assert_array_equal(be.to_host(vhf0), be.to_host(v_ref))
        be.huz_cycle_scalars_async(hv, None, zeros, zeros, dm, dm, dts=dts).get()
        fock1, vhf1 = be.jk_packed_fock(packed, dm, hv, dts=dts)
        np.testing.assert_array_equal(be.to_host(fock1), be.to_host(fock0))
        np.testing.assert_array_equal(be.to_host(vhf1), be.to_host(vhf0))


@pytest.mark.parametrize("n", [152, 192, 250, 256])
def test_jk_packed_fock_of_the_sizes_that_make_their_own_table(be, n):
    """N > 148 (jk_mx.hip: the weights of a tile no longer fit the loading waves' registers and are streamed from a
    table the build writes itself): no Dtot' table is handed over -- nbx_jk_dts_bytes is 0 -- and the fused Fock
    assembly equals J/K + nbx_fock_uhf bit for bit."""
    eri = be.synth_eri(n)
    packed = be.eri_pack(eri, n)
    hv = be.asarray(np.stack([symm(570, n), symm(571, n)]))
    with pytest.raises(ValueError):
        be.jk_dts_new(n)
    for seed in (572, 574):
        dm = be.asarray(np.stack([symm(seed, n), symm(seed + 1, n)]))
        fock0, vhf0 = be.jk_packed_fock(packed, dm, hv)
        f_ref, v_ref = be.fock_uhf(hv, None, be.jk_packed(packed, dm))
        np.testing.assert_array_equal(be.to_host(fock0), be.to_host(f_ref))
        np.testing.assert_array_equal(be.to_host(vhf0), be.to_host(v_ref))


@pytest.mark.parametrize("n", [104, 132, 148, 160, 250, 304])
def test_jk_mx_slabs_add_up_and_single_density(be, n):
    """csrc/jk_m8.hip (N = 104 -- three chunks per tile --, 132, 148: the 8-fold packed form, workgroup ranges cut at equal
    cost inside every slab) and
    csrc/jk_mx.hip through the slab interface that the multi-GPU split uses (three row slabs cut at equal triangular
    work, packed separately, partial J/K added: the whole-tensor result to rounding) and with ONE density (the NDM = 1
    instances), at a size with whole-row chunks, a zero-padded one and one with band-segment chunks; against the
    symmetric kernel on the dense tensor."""
    import torch

    torch.cuda.empty_cache()
    eri = be.synth_eri(n)
    dm = be.asarray(np.stack([symm(560, n), symm(561, n)]))
    ref = be.to_host(be.jk_sym(eri, dm))
    whole = be.to_host(be.jk_packed(be.eri_pack(eri, n), dm))
    np.testing.assert_allclose(whole, ref, rtol=0, atol=1e-11 * (n / 148) ** 2)
    cuts = [0] + [int(round(n * np.sqrt(g / 3.0))) for g in (1, 2)] + [n]
    acc = np.zeros_like(ref)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        acc += be.to_host(be.jk_packed(be.eri_pack(eri[lo:hi], n, lo, hi), dm, lo, hi))
    np.testing.assert_allclose(acc, ref, rtol=0, atol=1e-11 * (n / 148) ** 2)
    one = be.to_host(be.jk_packed(be.eri_pack(eri, n), dm[:1]))
    ref1 = be.to_host(be.jk_sym(eri, dm[:1]))
    np.testing.assert_allclose(one, ref1, rtol=0, atol=1e-11 * (n / 148) ** 2)


def test_jk_4fold_walk_kernel_behind_the_switch(be):
    """csrc/jk_m4.hip (the 4-fold packed walk kernel that served N = 97 .. 148 until csrc/jk_m8.hip took them) stays in
    the library behind NBX_JK_M8=0, which is read once per process: a child process runs N = 104, 147 (zero-padded)
    and 148 on it against the C oracle (tests/_jk4_worker.py)."""
    import gc
    import os
    import subprocess
    import sys
    from pathlib import Path

    import torch

    assert be.lib.nbx_jk_packed_fold(148) == 8 and be.lib.nbx_jk_packed_fold(145) == 8 and be.lib.nbx_jk_packed_fold(152) == 4
    gc.collect()
    torch.cuda.empty_cache()
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, NBX_JK_M8="0", PYTHONPATH=str(root))
    r = subprocess.run([sys.executable, str(root / "tests" / "_jk4_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "JK4 OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_jk_packed_unsupported_sizes(be):
    """Sizes outside the packed kernel's reach are refused loudly (the host then keeps the
    symmetric / plain kernel): too small, between instances, above 400."""
    from nbed_amd._nbx import NbxError

    for n in (3, 7, 13, 258, 322, 402, 500):
        assert not be.jk_packed_supported(n)
    eri = be.synth_eri(7)
    with pytest.raises((ValueError, NbxError)):
        be.eri_pack(eri, 7)


@pytest.mark.parametrize("n", [17, 22, 49, 75, 97, 102, 117, 147, 149, 150])
def test_jk_packed_zero_padded_sizes_vs_oracle(be, n):
    """Sizes the packed kernel has no instance for (odd N; N = 22, 102, 150: no fitting block
    geometry) run as the next covered size with zero rows/columns (include/nbx.h): J and K against
    the einsum / C oracle, slabs additive, Fock assembly, exact symmetry of J."""
    from oracle import cref

    assert be.lib.nbx_jk_packed_supported(n) == 2
    eri_h = cref.synth_eri(n)
    dm = np.stack([symm(536, n), symm(537, n)])
    ref = cref.jk(eri_h, dm)
    eri = be.asarray(eri_h)
    packed = be.eri_pack(eri, n)
    got = be.to_host(be.jk_packed(packed, be.asarray(dm)))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11)
    np.testing.assert_array_equal(got[0], got[0].T)
    cut = n // 3 + 1
    parts = (be.to_host(be.jk_packed(be.eri_pack(eri[:cut], n, 0, cut), be.asarray(dm), 0, cut))
             + be.to_host(be.jk_packed(be.eri_pack(eri[cut:], n, cut, n), be.asarray(dm), cut, n)))
    np.testing.assert_allclose(parts, got, rtol=0, atol=1e-11)
    hv = np.stack([symm(570, n), symm(571, n)])
    fock, vhf = be.jk_packed_fock(packed, be.asarray(dm), be.asarray(hv))
    np.testing.assert_allclose(be.to_host(vhf), ref[0] - ref[1:], rtol=0, atol=2e-11)
    np.testing.assert_allclose(be.to_host(fock), hv + ref[0] - ref[1:], rtol=0, atol=2e-11)
    one = be.to_host(be.jk_packed(packed, be.asarray(dm[:1])))
    np.testing.assert_allclose(one, cref.jk(eri_h, dm[:1]), rtol=0, atol=1e-11)


def test_fused_scf_on_a_zero_padded_size_vs_oracle(be):
    """N = 149 (odd: padded to 152 inside libnbx) through the fused Huzinaga loop against the oracle."""
    from oracle import cref
    from oracle import huzinaga as oracle_huz
    from oracle.pyscf_like import ToyMol, ToyUHF

    from nbed_amd.scf import GpuUHF, Mole, huzinaga_scf

    n, nocc, n_env = 149, (30, 29), 12
    pr = synth.problem(n, nocc, n_env)
    eri_h = cref.synth_eri(n)

    class CUHF(ToyUHF):
        def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
            jk = cref.jk(self._eri, np.asarray(dm))
            return jk[0] - jk[1:]

    ref = CUHF(ToyMol(n, pr["nelec"]), pr["S"], pr["hcore"], eri_h)
    mf = GpuUHF(Mole(n, pr["nelec"]), pr["S"], pr["hcore"], be.asarray(eri_h), backend=be)
    assert mf.eri_packed_device() is not None
    ref.max_cycle = mf.max_cycle = 60
    ref.conv_tol = mf.conv_tol = 1e-10
    rc, re, rd, rhz, rconv = oracle_huz.huzinaga_scf(ref, pr["V_emb"], pr["D_env"], dm_conv_tol=1e-8)
    c, e, d, hz, conv = huzinaga_scf(mf, pr["V_emb"], pr["D_env"], dm_conv_tol=1e-8)
    assert conv and rconv
    np.testing.assert_allclose(e, re, rtol=0, atol=1e-8)
    np.testing.assert_allclose(d, rd, rtol=0, atol=1e-8)
    np.testing.assert_allclose(hz, rhz, rtol=0, atol=1e-8)


def test_cdiis_device_and_orbital_gradient_norm(be):
    """nbx_diis_update_err (CDIIS ring on the device) against the oracle's pyscf.scf.diis.CDIIS
    restatement over 11 updates (space 8: the ring wraps), and nbx_vo_sumsq against numpy."""
    from nbed_amd.scf.diis import CDIIS as DeviceCDIIS
    from oracle.pyscf_like import CDIIS as OracleCDIIS

    n = 21
    s_h = synth.overlap(n)
    dev, ref = DeviceCDIIS(be, be.asarray(s_h)), OracleCDIIS()
    for it in range(11):
        d = np.stack([symm(440 + 2 * it, n), symm(441 + 2 * it, n)]) * 0.3
        f = np.stack([symm(470 + 2 * it, n), symm(471 + 2 * it, n)])
        got = be.to_host(dev.update(be.asarray(d), be.asarray(f)))
        want = ref.update(s_h, d, f)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * np.max(np.abs(want)), err_msg=f"update {it}")
    fmo = rnd(499, 2, n, n)
    nocc = (7, 5)
    g2 = be.to_host(be.vo_sumsq(be.asarray(fmo), nocc))
    np.testing.assert_allclose(g2, [np.sum(fmo[x][nocc[x]:, : nocc[x]] ** 2) for x in range(2)], rtol=1e-13)


def test_degenerate_shapes(be):
    """Empty slabs, single-element problems and zero inner dimensions through the C ABI."""
    n = 6
    eri = be.synth_eri(n)
    dm = be.asarray(np.stack([symm(500, n), symm(501, n)]))
    # empty row slab: the additive form contributes zeros, the row-slab form has no rows
    assert float(be.jk_sym(eri[:0], dm, 3, 3).abs().max()) == 0.0
    assert be.jk(eri[:0], dm, 3, 3).shape == (3, 0, n)
    # empty outer-index slab of the transform
    c = be.asarray(rnd(502, n, 4))
    assert be.ao2mo(eri, c, c, c, c, i0=2, i1=2).shape == (0, 4, 4, 4)
    a1, a2 = be.ao2mo_pair(eri, c, c, c, c, c, c, i0=1, i1=1)
    assert a1.shape == (0, 4, 4, 4) and a2.shape == (0, 4, 4, 4)
    # 1 x 1 eigenproblem, cold and warm
    one = be.asarray(np.array([[2.5]]))
    w, v = be.eigh(one)
    assert float(w[0]) == 2.5 and abs(abs(float(v[0, 0])) - 1.0) < 1e-15
    w, v = be.eigh(one, v0=v)
    assert float(w[0]) == 2.5
    # k = 0 product: beta * C
    cm = be.asarray(rnd(503, 5, 7))
    out = be.gemm(be.asarray(np.zeros((5, 0))), be.asarray(np.zeros((0, 7))), beta=2.0, out=be.copy(cm))
    np.testing.assert_allclose(be.to_host(out), 2.0 * be.to_host(cm), rtol=0, atol=0)
    # N = 2 symmetric J/K (one column pair per row)
    e2 = synth.eri_dense(2)
    d2 = np.stack([symm(504, 2), symm(505, 2)])
    vj, vk = get_jk(e2, d2)
    got = be.to_host(be.jk_sym(be.asarray(e2), be.asarray(d2)))
    np.testing.assert_allclose(got[0], vj.sum(0), atol=1e-14)
    np.testing.assert_allclose(got[1:], vk, atol=1e-14)


@pytest.mark.parametrize("n", [2, 6, 24, 38])
@pytest.mark.parametrize("ndm", [1, 2])
def test_jk_synth_sym_vs_oracle(be, n, ndm):
    """Generated-integral J/K with the tiles q <= p only: against einsum on the dense synthetic
    tensor; slabs add up."""
    eri_h = synth.eri_dense(n)
    dm = np.stack([symm(520 + x, n) for x in range(ndm)])
    vj, vk = get_jk(eri_h, dm)
    got = be.to_host(be.jk_synth_sym(n, be.asarray(dm)))
    np.testing.assert_allclose(got[0], vj.sum(0) if vj.ndim == 3 else vj, rtol=0, atol=1e-12)
    np.testing.assert_allclose(got[1:], vk.reshape(ndm, n, n), rtol=0, atol=1e-12)
    if n >= 6:
        cut = n // 2 + 1
        parts = be.to_host(be.jk_synth_sym(n, be.asarray(dm), 0, cut)) + be.to_host(be.jk_synth_sym(n, be.asarray(dm), cut, n))
        np.testing.assert_allclose(parts, got, rtol=0, atol=1e-12)


def test_jk_synth_sym_wide_rows_match_plain_kernel(be):
    """N = 600 and 1100: two and four column segments per thread; a few slab rows against nbx_jk_synth."""
    for n, rows in ((600, (297, 300)), (1100, (548, 550))):
        dm = be.asarray(np.stack([symm(530, n), symm(531, n)]))
        p0, p1 = rows
        sym = be.to_host(be.jk_synth_sym(n, dm, p0, p1))
        plain = be.to_host(be.jk_synth(n, dm, p0, p1))  # rows p0..p1 of J and K
        # the additive form holds the pairs (p, q <= p): compare what both determine completely,
        # J[p, q <= p] and the K rows of the LAST slab row restricted to ... use J only + symmetry
        for i, p in enumerate(range(p0, p1)):
            np.testing.assert_allclose(sym[0, p, : p + 1], plain[0, i, : p + 1], rtol=0, atol=1e-11)
            np.testing.assert_allclose(sym[0, : p + 1, p], plain[0, i, : p + 1], rtol=0, atol=1e-11)


def test_large_device_to_host_copy_is_exact(be):
    """Tensors of 512 MB and more come back through the pinned-buffer pipeline: bit-exact, any shape,
    including a ragged last chunk."""
    n = (1 << 26) + 12345
    a = be.torch.arange(n, dtype=be.torch.float64, device=be.device) * 0.5
    h = be.to_host(a)
    assert h.shape == (n,) and h.dtype == np.float64
    np.testing.assert_array_equal(h[:5], [0.0, 0.5, 1.0, 1.5, 2.0])
    assert h[-1] == (n - 1) * 0.5 and h[33554432] == 33554432 * 0.5 and h[33554431] == 33554431 * 0.5
    assert float(np.sum(h[::4097])) == float(a[::4097].sum())
    b = a[: 1 << 26].reshape(64, 1024, 1024)
    np.testing.assert_array_equal(be.to_host(b)[63, 1023, 1020:], be.to_host(b[63, 1023, 1020:]))
    # and the way up: a large host array through the same kind of pipeline
    up = be.asarray(h)
    assert up.shape == (n,) and bool((up == a).all())


def test_spinorb_scatter_streamed_equals_dense(be):
    """The piecewise (streamed-to-host) scatter against the one-shot kernel, at a size that takes the
    streaming path (n = 46: 92^4 = 71.6 M elements, three pieces with a ragged last one)."""
    n = 46
    one = be.asarray(rnd(540, 2, n, n))
    two = be.asarray(rnd(541, 4, n, n, n, n) * 1e-3)
    h1_d, h2_d = be.spinorb_scatter(one, two, 1e-8, 0.5)
    h1_h, h2_h = be.spinorb_scatter_to_host(one, two, 1e-8, 0.5)
    np.testing.assert_array_equal(h1_h, be.to_host(h1_d))
    ref = h2_d.cpu().numpy()
    assert h2_h.shape == ref.shape
    np.testing.assert_array_equal(h2_h, ref)


def test_c_abi_error_behaviour(be):
    """No exception crosses the C boundary: bad arguments, short workspaces and unsupported sizes
    come back as negative NBX_E_* codes with a message (surfaced as NbxError by the binding); a
    failed call leaves the context usable."""
    import ctypes

    from nbed_amd import _nbx

    lib, ctx = be.lib, be.ctx
    n = 8
    a = be.asarray(symm(400, n))
    dm = be.asarray(np.stack([symm(401, n), symm(402, n)]))
    eri = be.synth_eri(n)
    out = be.empty((3, n, n))
    null = ctypes.c_void_p(0)

    def code(name, *args):
        return getattr(lib, name)(ctx, *args)

    # null pointers / bad shapes -> NBX_E_INVALID
    assert code("nbx_jk_dense", n, 0, n, null, be._p(dm), 2, be._p(out), null, 0) == -1
    assert code("nbx_jk_dense", n, 0, n + 1, be._p(eri), be._p(dm), 2, be._p(out), null, 0) == -1
    assert code("nbx_jk_dense", n, 0, n, be._p(eri), be._p(dm), 3, be._p(out), null, 0) == -1
    assert code("nbx_gemm", b"X", b"N", n, n, n, 1.0, be._p(a), n, 0, be._p(a), n, 0, 0.0, be._p(out), n, 0, 1) == -1
    assert code("nbx_gemm", b"N", b"N", n, n, n, 1.0, be._p(a), n - 1, 0, be._p(a), n, 0, 0.0, be._p(out), n, 0, 1) == -1
    assert code("nbx_diis_update", 16, 6, 7, 1, be._p(a), be._p(a), be._p(a), be._p(a), be._p(a), be._p(a)) == -1
    assert code("nbx_diis_update", 16, 17, 0, 1, be._p(a), be._p(a), be._p(a), be._p(a), be._p(a), be._p(a)) == -1
    assert code("nbx_eigh_warm_ex", n, 1, be._p(a), null, be._p(out), be._p(out), be._p(out), 1 << 30, -1) == -1
    # workspace too small -> NBX_E_NOMEM, with the required size in the message
    need = lib.nbx_jk_dense_worksize(n, n, 2)
    work = be.torch.empty(need, dtype=be.torch.uint8, device=be.device)
    assert code("nbx_jk_dense", n, 0, n, be._p(eri), be._p(dm), 2, be._p(out), be._p(work), need - 1) == -3
    assert str(need).encode() in lib.nbx_last_error()
    assert code("nbx_eigh", n, 1, be._p(a), be._p(out), be._p(out), be._p(work), 8) == -3
    # sizes the kernels do not cover -> NBX_E_UNSUPPORTED (never a silent wrong answer)
    assert code("nbx_jk_synth", 2050, 0, 1, 1, be._p(dm), 2, be._p(out), be._p(work), need) in (-5, -3)
    # the binding raises, and the context still works afterwards
    with pytest.raises(_nbx.NbxError) as err:
        be._call("nbx_axpby", 4, 1.0, null, 0.0, null)
    assert err.value.code == -1
    jk = be.jk(eri, dm)
    eri_h = synth.eri_dense(n)
    np.testing.assert_allclose(be.to_host(jk)[0], np.einsum("pqrs,rs->pq", eri_h, be.to_host(dm).sum(0)), atol=1e-13)


@pytest.mark.parametrize("n", [37, 148, 200])
@pytest.mark.parametrize("p", [-0.5, 0.5, -1.0])
def test_sym_pow_newton_schulz(be, n, p):
    """S^p by coupled Newton-Schulz GEMM iterations (the SCF's S^-1/2, SPADE's S^1/2) against the
    spectral definition and the eigensolver route; refusal (None) outside its domain."""
    s = synth.overlap(n)
    w, u = np.linalg.eigh(s)
    ref = (u * w**p) @ u.T
    got = be.sym_pow_newton_schulz(be.asarray(s), p, s)
    assert got is not None
    got = be.to_host(got)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13)
    np.testing.assert_allclose(got, got.T, rtol=0, atol=1e-15)
    np.testing.assert_allclose(be.to_host(be.sym_pow_fast(be.asarray(s), p, s)), be.to_host(be.sym_pow(be.asarray(s), p)),
                               rtol=0, atol=1e-12)
    if n == 37:
        # moderately ill-conditioned (1e4): still accurate to cond * eps
        q, _ = np.linalg.qr(rnd(95, n, n))
        wv = np.logspace(0, -4, n)
        s2 = (q * wv) @ q.T
        s2 = 0.5 * (s2 + s2.T)
        got2 = be.sym_pow_newton_schulz(be.asarray(s2), p, s2)
        ref2 = (q * wv**p) @ q.T
        assert got2 is not None
        np.testing.assert_allclose(be.to_host(got2), ref2, rtol=0, atol=1e-10 * np.max(np.abs(ref2)))
        # indefinite / hopelessly conditioned input: refused, and sym_pow_fast falls back to the eigen route
        assert be.sym_pow_newton_schulz(be.asarray(symm(96, n)), p) is None
        s3 = (q * np.logspace(0, -9, n)) @ q.T
        s3 = 0.5 * (s3 + s3.T)
        assert be.sym_pow_newton_schulz(be.asarray(s3), p, s3) is None


@pytest.mark.parametrize("n,nocc,gap", [(148, (5, 5), 0.5), (148, (33, 30), 0.05), (100, (10, 9), 0.02), (37, (3, 3), 0.3),
                                        (20, (1, 19), 0.4)])
def test_purify_matches_the_eigenvector_projector(be, n, nocc, gap):
    """nbx_purify (SP2 purification, one MFMA product per step, verdict on the device) against C_occ C_occ^T of
    numpy's eigh: spectra 22 Ha wide with the gap at the Fermi level set by hand, low and high filling."""
    rng = np.random.default_rng(n + nocc[0])
    fs, ps = [], []
    for x in range(2):
        q, _ = np.linalg.qr(rng.normal(size=(n, n)))
        lam = np.sort(rng.uniform(-12, 10, size=n))
        k = nocc[x]
        lam[k:] = lam[k:] - lam[k] + lam[k - 1] + gap
        f = (q * lam) @ q.T
        fs.append(0.5 * (f + f.T))
        ps.append(q[:, :k] @ q[:, :k].T)
    p_d, st = be.purify(be.asarray(np.stack(fs)), nocc)
    st = st.cpu().numpy()
    assert np.all(st > 0) and np.all(st <= 72)
    p = be.to_host(p_d)
    np.testing.assert_allclose(p, np.stack(ps), rtol=0, atol=5e-11)
    np.testing.assert_allclose(p, p.transpose(0, 2, 1), rtol=0, atol=1e-13)
    assert abs(np.trace(p[0]) - nocc[0]) < 1e-9 and abs(np.trace(p[1]) - nocc[1]) < 1e-9
    # reproducible bit for bit; a step limit that is too small is reported, not papered over
    np.testing.assert_array_equal(be.to_host(be.purify(be.asarray(np.stack(fs)), nocc)[0]), p)
    assert np.all(be.purify(be.asarray(np.stack(fs)), nocc, max_iter=5)[1].cpu().numpy() == -1)


def test_purify_reports_a_missing_gap(be):
    """Levels nocc and nocc + 1 degenerate: no projector exists, the status words say so."""
    n, k = 64, 7
    rng = np.random.default_rng(1)
    q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.sort(rng.uniform(-5, 5, size=n))
    lam[k] = lam[k - 1]
    f = (q * lam) @ q.T
    _, st = be.purify(be.asarray(np.stack([0.5 * (f + f.T)] * 2)), (k, k))
    assert np.all(st.cpu().numpy() < 0)


@pytest.mark.parametrize("n", [37, 104, 148, 260])
def test_eigh_approx_seeds_the_warm_solver_from_another_stream(be, n):
    """nbx_eigh_approx (the tridiagonal route with nothing read back) on a side stream: eigenvalues as numpy's,
    vectors orthonormal with small residuals, and nbx_eigh_warm_ex started from them gives nbx_eigh's result."""
    import torch

    rng = np.random.default_rng(n)
    mats = []
    for x in range(2):
        q, _ = np.linalg.qr(rng.normal(size=(n, n)))
        lam = np.sort(rng.uniform(-11, 4, size=n))
        f = (q * lam) @ q.T
        mats.append(0.5 * (f + f.T))
    a_h = np.stack(mats)
    a = be.asarray(a_h)
    side = be.side()
    assert side is be.side() and side.ctx.value != be.ctx.value
    be.fork_to(side)
    with torch.cuda.stream(side.stream):
        w, v, st = side.eigh_approx(a)
        st_h = side.async_to_host(st)
    assert np.all(st_h.get() == 1)
    be.join_from(side)
    w_h, v_h = be.to_host(w), be.to_host(v)
    for x in range(2):
        np.testing.assert_allclose(w_h[x], np.linalg.eigvalsh(a_h[x]), rtol=0, atol=1e-11)
        np.testing.assert_allclose(v_h[x].T @ v_h[x], np.eye(n), rtol=0, atol=1e-11)
        assert np.abs(a_h[x] @ v_h[x] - v_h[x] * w_h[x]).max() < 1e-7
    w_cold, v_cold = be.eigh(a)
    w_warm, v_warm = be.eigh(a, v0=v, refine_iters=6)
    np.testing.assert_allclose(be.to_host(w_warm), be.to_host(w_cold), rtol=0, atol=1e-12)
    vw = be.to_host(v_warm)
    for x in range(2):
        assert np.abs(a_h[x] @ vw[x] - vw[x] * be.to_host(w_warm)[x]).max() < 1e-11
    # same bits again (the SCF loop relies on the side solve being reproducible)
    with torch.cuda.stream(side.stream):
        w2, v2, _ = side.eigh_approx(a)
    be.join_from(side)
    np.testing.assert_array_equal(be.to_host(v2), v_h)


def test_eigh_approx_flags_vectors_it_could_not_separate(be):
    """A spectrum with an exactly degenerate cluster: inverse iteration cannot separate it, status -1 (or, if the
    vectors happen to come out independent, they are orthonormal): never a silent non-orthonormal start."""
    n = 96
    rng = np.random.default_rng(5)
    q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.sort(rng.uniform(-3, 3, size=n))
    lam[40:52] = lam[40]
    f = (q * lam) @ q.T
    a = be.asarray(np.stack([0.5 * (f + f.T)] * 2))
    w, v, st = be.eigh_approx(a)
    st, v_h = st.cpu().numpy(), be.to_host(v)
    for x in range(2):
        if st[x] > 0:
            np.testing.assert_allclose(v_h[x].T @ v_h[x], np.eye(n), rtol=0, atol=1e-10)
        else:
            assert st[x] == -1


@pytest.mark.parametrize("g,n", [(1000, 148), (257, 7), (64, 200), (3001, 24), (130, 72), (70, 165)])
def test_xc_density_and_potential_passes_match_the_tensor_expressions(be, g, n):
    """nbx_xc_rho (c = ao D on the matrix cores, rho / grad rho reduced in the epilogue, both spins) and nbx_xc_vmat
    (ao^T half + its transpose with ``half`` built on the fly, split over chunks of grid points) against the tensor
    expressions of nbed_amd.xc.XCProvider.__call__ they replace -- every tile-count / block-size instance."""
    import torch

    gen = torch.Generator(device="cpu").manual_seed(g + n)
    ao = torch.randn(g, n, dtype=torch.float64, generator=gen)
    dao = torch.randn(3, g, n, dtype=torch.float64, generator=gen)
    dm = torch.randn(2, n, n, dtype=torch.float64, generator=gen)
    dm = 0.5 * (dm + dm.transpose(1, 2))
    ao_d, dao_d = ao.to(be.device), dao.to(be.device)
    rho, grad = be.xc_rho(ao_d, dao_d, dm.to(be.device))
    c = torch.einsum("gm,xmn->xgn", ao, dm)
    want_rho = (c * ao[None]).sum(dim=2)
    want_grad = 2.0 * torch.einsum("xgn,agn->xag", c, dao)
    scale = float(want_rho.abs().max())
    torch.testing.assert_close(rho.cpu(), want_rho, rtol=0, atol=1e-12 * max(scale, 1.0))
    torch.testing.assert_close(grad.cpu(), want_grad, rtol=0, atol=1e-12 * max(float(want_grad.abs().max()), 1.0))
    vr = torch.randn(2, g, dtype=torch.float64, generator=gen)
    vec = torch.randn(2, 3, g, dtype=torch.float64, generator=gen)
    got = be.xc_vmat(ao_d, dao_d, vr.to(be.device), vec.to(be.device)).cpu()
    half = 0.5 * vr[:, :, None] * ao[None] + torch.einsum("xag,agn->xgn", vec, dao)
    v = torch.einsum("gm,xgn->xmn", ao, half)
    want = v + v.transpose(1, 2)
    torch.testing.assert_close(got, want, rtol=0, atol=2e-12 * max(float(want.abs().max()), 1.0))
    np.testing.assert_array_equal(got.numpy(), be.xc_vmat(ao_d, dao_d, vr.to(be.device), vec.to(be.device)).cpu().numpy())


@pytest.mark.parametrize("name", ["b3lyp", "lda", "lda,vwn", "slater"])
def test_xc_functional_kernel_matches_autograd(be, name):
    """nbx_xc_functional -- energy density and first derivatives written out analytically (Slater, B88, VWN-RPA / VWN5,
    LYP) -- against the torch expression it replaces: ``nbed_amd.xc.energy_density`` differentiated by autograd, with
    XCProvider's conventions (clamped densities, weights folded in, empty points dropped).  Densities over eleven
    orders of magnitude, spin polarisations up to 1e3, reduced gradients from 0.01 to 100."""
    import torch

    from nbed_amd import _nbx, xc

    rng = np.random.default_rng(7)
    g = 5000
    ra = 10 ** rng.uniform(-9, 2, g)
    rb = ra * 10 ** rng.uniform(-3, 3, g)
    ga = rng.normal(size=(3, g)) * ra ** (4 / 3) * 10 ** rng.uniform(-2, 2, g)
    gb = rng.normal(size=(3, g)) * rb ** (4 / 3) * 10 ** rng.uniform(-2, 2, g)
    ra[:50] = 10 ** rng.uniform(-18, -14, 50)  # below the floor: clamped / dropped points
    rb[:25] = 10 ** rng.uniform(-18, -14, 25)
    w = rng.uniform(0.1, 2.0, g)
    floor = xc.XCProvider.RHO_FLOOR
    rho = torch.tensor(np.stack([ra, rb]))
    grad = torch.tensor(np.stack([ga, gb]))
    wt = torch.tensor(w)
    keep = ((rho[0] + rho[1]) > floor).to(torch.float64)
    tra = torch.clamp(rho[0], min=0.5 * floor).requires_grad_(True)
    trb = torch.clamp(rho[1], min=0.5 * floor).requires_grad_(True)
    saa = ((grad[0] * grad[0]).sum(dim=0) + 1e-40).requires_grad_(True)
    sab = (grad[0] * grad[1]).sum(dim=0).requires_grad_(True)
    sbb = ((grad[1] * grad[1]).sum(dim=0) + 1e-40).requires_grad_(True)
    exc = (wt * keep * xc.energy_density(name, tra, trb, saa, sab, sbb)).sum()
    vra, vrb, vsaa, vsab, vsbb = (x if x is not None else torch.zeros_like(wt) for x in torch.autograd.grad(
        exc, (tra, trb, saa, sab, sbb), allow_unused=True))
    want_vr = torch.stack([vra, vrb])
    want_vec = torch.stack([2.0 * vsaa * grad[0] + vsab * grad[1], 2.0 * vsbb * grad[1] + vsab * grad[0]])
    vr, vec, sums = be.xc_functional(_nbx.XC_CODES[name], rho.to(be.device), grad.to(be.device), wt.to(be.device), floor)
    sums = sums.cpu().numpy()
    assert abs(sums[0] - float(exc)) < 1e-12 * abs(float(exc))
    assert abs(sums[1] - float((wt * (rho[0] + rho[1])).sum())) < 1e-12 * float((wt * (rho[0] + rho[1])).sum())
    # 1e-11 relative to the size of each entry (the LYP derivative is a difference of terms up to 1e4 times larger)
    err_r = ((vr.cpu() - want_vr).abs() / (want_vr.abs() + 1e-300)).max()
    err_v = ((vec.cpu() - want_vec).abs() / (want_vec.abs() + 1e-300)).max()
    assert float(err_r) < 5e-11 and float(err_v) < 5e-11, (float(err_r), float(err_v))


@pytest.mark.parametrize("n", [64, 65, 128, 129, 150, 151, 176, 177, 198, 199, 208, 209])
def test_cold_eigh_across_the_register_and_lds_kernel_instances(be, n):
    """Cold nbx_eigh (tridiagonal route) on either side of every size at which a different instance of
    tridiag_reg_kernel / invit_lds_kernel / backtransform_reg_kernel -- or the in-memory kernel -- takes over."""
    rng = np.random.default_rng(n)
    mats = []
    for x in range(3):
        q, _ = np.linalg.qr(rng.normal(size=(n, n)))
        lam = np.sort(rng.uniform(-15, 6, size=n))
        f = (q * lam) @ q.T
        mats.append(0.5 * (f + f.T))
    a_h = np.stack(mats)
    w, v = be.eigh(be.asarray(a_h))
    w_h, v_h = be.to_host(w), be.to_host(v)
    for x in range(3):
        np.testing.assert_allclose(w_h[x], np.linalg.eigvalsh(a_h[x]), rtol=0, atol=1e-12)
        np.testing.assert_allclose(v_h[x].T @ v_h[x], np.eye(n), rtol=0, atol=1e-13)
        assert np.abs(a_h[x] @ v_h[x] - v_h[x] * w_h[x]).max() < 1e-12


def test_kernel_timing_brackets_and_sampling(be):
    """nbx_profile_enable / nbx_profile_sample / nbx_profile_read (include/nbx.h "in-library kernel timing"): HIP
    events around the J/K kernel of every launch, or of one launch in `every`; what bench.py's roofline uses."""
    from nbed_amd import _nbx

    n = 24
    eri = be.synth_eri(n)
    dm = be.asarray(np.stack([symm(880, n), symm(881, n)]))
    be.profile(True, slots=[_nbx.PROF_JK_DENSE])
    be.profile_reset()
    for _ in range(8):
        be.jk_sym(eri, dm)
    ms_all, cnt_all = be.profile_read(_nbx.PROF_JK_DENSE)
    assert cnt_all == 8 and 0.0 < ms_all < 50.0
    be.profile(True, slots=[_nbx.PROF_JK_DENSE], every=4)
    be.profile_reset()
    for _ in range(8):
        be.jk_sym(eri, dm)
    ms_s, cnt_s = be.profile_read(_nbx.PROF_JK_DENSE)
    assert cnt_s == 2 and 0.0 < ms_s < 50.0  # the first launch after the reset, then every fourth
    be.profile(True, slots=[_nbx.PROF_EIGH])  # a slot that is not enabled records nothing
    be.profile_reset()
    be.jk_sym(eri, dm)
    assert be.profile_read(_nbx.PROF_JK_DENSE)[1] == 0
    be.profile(False)
    be.profile_reset()
    be.jk_sym(eri, dm)
    assert be.profile_read(_nbx.PROF_JK_DENSE)[1] == 0


def test_kernels_do_not_depend_on_what_lds_held(be):
    """Every CU's LDS filled with NaN (nbx_debug_fill_lds) before each launch: the J/K kernels, the transform and the
    eigensolver give the bits they gave before.  (A kernel that multiplies something it masked to zero by LDS it has not
    written passes every parity test until an earlier kernel happens to have left a NaN there.)"""
    nan = float("nan")

    def same(fn):
        ref = be.to_host(fn())
        assert np.isfinite(ref).all()
        for _ in range(2):
            be.debug_fill_lds(nan)
            np.testing.assert_array_equal(be.to_host(fn()), ref)

    for n in (100, 116, 128, 148):
        eri = be.synth_eri(n)
        packed = be.eri_pack(eri, n)
        for ndm in (2, 1):
            dm = be.asarray(np.stack([symm(900 + x, n) for x in range(ndm)]))
            same(lambda: be.jk_packed(packed, dm))
            same(lambda: be.jk_sym(eri, dm))
        h = n // 3
        slab = be.eri_pack(eri[h:], n, h, n)
        dm = be.asarray(np.stack([symm(902, n), symm(903, n)]))
        same(lambda: be.jk_packed(slab, dm, h, n))
        del eri, packed, slab
    n = 24
    eri = be.synth_eri(n)
    dm = be.asarray(np.stack([symm(904, n), symm(905, n)]))
    same(lambda: be.jk(eri, dm))
    same(lambda: be.jk_sym(eri, dm))
    a = be.asarray(np.stack([symm(906, 72), symm(907, 72)]))
    same(lambda: be.eigh(a)[0])
    same(lambda: be.eigh(a)[1])
    same(lambda: be.purify(be.asarray(np.stack([symm(906, 72), symm(907, 72)]) / 40.0), (20, 19))[0])
    same(lambda: be.huzinaga_fused(a, be.asarray(np.stack([symm(908, 72), symm(909, 72)])), 1.0)[0])
    # the transform's products at the benchmark shape (the 'T','N' kernel with its k-tiles written into LDS by the load
    # unit, the tiled and the small GEMM) and a streamed one with the integrals generated in registers
    n, na = 148, 128
    eri = be.synth_eri(n)
    c = be.asarray(np.linalg.qr(symm(910, n))[0][:, :na].copy())
    same(lambda: be.ao2mo(eri, c, c, c, c))
    del eri
    c6 = be.asarray(np.linalg.qr(symm(911, 96))[0][:, :16].copy())
    same(lambda: be.ao2mo_synth(96, c6, c6, c6, c6))
    for m, k, nn in ((148, 148, 148), (300, 20, 70), (33, 257, 129)):
        x, y = be.asarray(symm(912, max(m, k))[:m, :k].copy()), be.asarray(symm(913, max(k, nn))[:k, :nn].copy())
        same(lambda: be.gemm(x, y))
        same(lambda: be.gemm(y, x, "T", "T"))
