"""GPU parity tests AT THE SIZES THE BENCHMARK QUOTES, each against the CPU oracle (never against
another HIP kernel):

* BASELINE configs[2] shape (N_AO = 148, n = 128): packed J/K, the rs-packed pair transform, and a
  whole fused Huzinaga SCF run of 25 cycles with DIIS;
* BASELINE configs[3] shape (N_AO = 2000, n = 128, integrals generated in registers): one r-slab of the
  streamed transform -- the <128,128,B_GEN> instance of the GEMM -- in full at low r and on sampled
  (i,j,k,l) at r = 1000, plus the streamed transform at n_act >= 72 against the dense einsum oracle;
* N = 128 / 192 / 256 instances of the packed J/K kernel, N = 600 / 1100 of the generated one (J and K).

Reference contract: nbed/scf/huzinaga_scf.py:154-201 (the loop), :156 (get_veff -> J/K),
nbed/ham_builder.py:119-133 (the four spin-block transforms).  The oracle's integrals at these sizes
come from oracle/c/synth_ref.c (bit-identical to oracle/synth.py, tests/test_oracle_golden.py).
"""

import numpy as np
import pytest

from oracle import cref, hamiltonian, synth
from oracle import huzinaga as oracle_huz
from oracle.pyscf_like import ToyMol, ToyUHF

pytestmark = pytest.mark.gpu

N_BENCH, N_ACT = 148, 128


@pytest.fixture(scope="module")
def be():
    from nbed_amd.backend import HipBackend

    return HipBackend()


@pytest.fixture(scope="module")
def eri148():
    """The dense synthetic (pq|rs) at the bench size from the C oracle (3.84 GB on the host)."""
    return cref.synth_eri(N_BENCH)


def symm(stream, n):
    return synth.sym_matrix(stream, n)


def rnd(stream, *shape):
    return synth.val(stream, np.arange(int(np.prod(shape)))).reshape(shape)


# ---------------------------------------------------------------- J/K at the bench size
def test_jk_packed_n148_vs_c_oracle(be, eri148):
    """jk_m4_kernel's N = 148 instance (the bench's roofline kernel: the packed J/K walk on v_mfma_f64_4x4x4,
    csrc/jk_m4.hip) against oracle/c/jk_ref.c on the same tensor: J, K_alpha, K_beta, all rows.  Also the device
    generator against the C generator."""
    n = N_BENCH
    eri = be.synth_eri(n)
    np.testing.assert_array_equal(be.to_host(eri[7:9]), eri148[7:9])  # same inputs on both sides
    dm = np.stack([symm(532, n), symm(533, n)])
    ref = cref.jk(eri148, dm)
    got = be.to_host(be.jk_packed(be.eri_pack(eri, n), be.asarray(dm)))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11)
    np.testing.assert_array_equal(got[0], got[0].T)
    # the fused Fock assembly of the SCF loop on top of it
    hv = np.stack([symm(570, n), symm(571, n)])
    fock, vhf = be.jk_packed_fock(be.eri_pack(eri, n), be.asarray(dm), be.asarray(hv))
    np.testing.assert_allclose(be.to_host(vhf), ref[0] - ref[1:], rtol=0, atol=2e-11)
    np.testing.assert_allclose(be.to_host(fock), hv + ref[0] - ref[1:], rtol=0, atol=2e-11)


MX_SIZES = list(range(152, 257, 8)) + [272, 288, 304, 320, 336, 352, 368, 384]


@pytest.mark.parametrize("n", list(range(100, 148, 4)) + MX_SIZES + [156, 188, 252, 300])
def test_jk_packed_every_instance_vs_c_oracle(be, n):
    """EVERY instance of the packed J/K kernels -- each size of the MFMA walk is separately generated straight-line code
    (csrc/jk_m4.hip: one instance per multiple of four, N = 100 .. 148; csrc/jk_mx.hip: every multiple of eight up to
    256, whole block rows per chunk, and 272 .. 384 in steps of sixteen, from N = 304 with the lower bands cut into
    column segments; 156, 188, 252, 300 run zero-padded as the next instance) -- on three row slabs (first, middle,
    last rows) of J and K against the C oracle (oracle/c/jk_ref.c) on slabs of the generated tensor.  (N = 148 itself:
    all rows, above.)"""
    import gc

    import torch

    # (the dense tensor of N = 384 is 174 GB: nothing of the size before may stay reserved -- a workspace of the backend
    #  that the allocator carved out of the freed 146 GB segment of N = 368 pins that whole segment)
    be.release_workspaces()
    gc.collect()
    torch.cuda.empty_cache()
    eri = be.synth_eri(n)
    dm = np.stack([symm(534, n), symm(535, n)])
    packed = be.eri_pack(eri, n)
    del eri
    got = be.to_host(be.jk_packed(packed, be.asarray(dm)))
    del packed
    for p0, p1 in [(0, 3), (n // 2 - 1, n // 2 + 2), (n - 3, n)]:
        ref = cref.jk(cref.synth_eri(n, p0, p1), dm, p0, p1)  # (3, rows, N): J rows, K rows
        np.testing.assert_allclose(got[:, p0:p1], ref, rtol=0, atol=1e-11 * (n / 148) ** 2)
    np.testing.assert_array_equal(got[0], got[0].T)


def test_jk_synth_sym_wide_rows_j_and_k_vs_c_oracle(be):
    """N = 600 and 1100 (two and four column segments per thread): the additive slab form -- J AND K
    contributions of the pairs (p, q <= p) and their mirror images -- against the C oracle's."""
    for n, (p0, p1) in ((600, (297, 300)), (1100, (548, 550))):
        dm = np.stack([symm(530, n), symm(531, n)])
        got = be.to_host(be.jk_synth_sym(n, be.asarray(dm), p0, p1))
        ref = cref.jk_synth_sym(n, dm, p0, p1)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-11)


# ---------------------------------------------------------------- transform at the bench size
def test_ao2mo_pair_sym_rs_bench_shape_vs_oracle(be, eri148):
    """nbx_ao2mo_pair_sym_rs at (N, n) = (148, 128) -- triangular-batch GEMM, symmetric-packed A
    operand, pair-scatter epilogue -- against the einsum oracle on sampled outer-index slabs of all
    three spin blocks; outputs exactly symmetric in (i, j)."""
    n, na = N_BENCH, N_ACT
    pr = synth.problem(n, (33, 33), 20)
    _, c = synth.lowdin_orthonormal(pr["S"], pr["hcore"])
    ca_h = np.ascontiguousarray(c[:, :na])
    cb_h = np.ascontiguousarray(c[:, ::-1][:, :na])
    eri = be.synth_eri(n)
    eri_rs = be.eri_pack_rs(eri, n)
    del eri
    ca, cb = be.asarray(ca_h), be.asarray(cb_h)
    aa, ab = be.ao2mo_pair_sym(eri_rs, ca, ca, ca, cb, cb, rs_packed=True)
    bb = be.ao2mo_pair_sym(eri_rs, cb, cb, cb, rs_packed=True)
    idx = [0, 1, 63, 100, 127]
    for got_d, (c12, c3) in ((aa, (ca_h, ca_h)), (ab, (ca_h, cb_h)), (bb, (cb_h, cb_h))):
        ref = hamiltonian.ao2mo_full(eri148, np.ascontiguousarray(c12[:, idx]), c12, c3, c3)
        got = be.to_host(got_d[idx])
        scale = np.max(np.abs(ref))
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * max(scale, 1.0) * n)
        assert bool((got_d == got_d.transpose(0, 1)).all())  # (ij|kl) = (ji|kl) exactly
    # the i-sharded form of the same transform (multi-GPU path) on one slab
    s_aa, s_ab = be.ao2mo_pair(be.synth_eri(n), ca, ca, ca, ca, cb, cb, i0=60, i1=64)
    np.testing.assert_allclose(be.to_host(s_aa), be.to_host(aa[60:64]), rtol=0, atol=1e-11)
    ref = hamiltonian.ao2mo_full(eri148, np.ascontiguousarray(ca_h[:, 60:64]), ca_h, cb_h, cb_h)
    np.testing.assert_allclose(be.to_host(s_ab), ref, rtol=0, atol=1e-11)


# ---------------------------------------------------------------- streamed transform, large tile instance
def test_ao2mo_synth_large_tile_instance_vs_oracle(be):
    """n_act = 72 > 64 columns and N = 80: the streamed transform runs the <128,128,...,B_GEN> GEMM
    instance (the N = 2000 kernel); compared with the dense einsum oracle in full, and as r-shards."""
    n, na = 80, 72
    eri_h = cref.synth_eri(n)
    cs = [rnd(600 + i, n, na) for i in range(4)]
    ref = hamiltonian.ao2mo_full(eri_h, *cs)
    dcs = [be.asarray(c) for c in cs]
    got = be.to_host(be.ao2mo_synth(n, *dcs))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11)
    part = be.to_host(be.ao2mo_synth(n, *dcs, r0=0, r1=51)) + be.to_host(be.ao2mo_synth(n, *dcs, r0=51, r1=n))
    np.testing.assert_allclose(part, ref, rtol=0, atol=1e-11)
    p1, p2 = be.ao2mo_synth_pair(n, dcs[0], dcs[1], dcs[2], dcs[3], dcs[3], dcs[2])
    np.testing.assert_allclose(be.to_host(p1), ref, rtol=0, atol=1e-11)
    np.testing.assert_allclose(be.to_host(p2), hamiltonian.ao2mo_full(eri_h, cs[0], cs[1], cs[3], cs[2]), rtol=0,
                               atol=1e-11)


def _n2000_coeffs():
    nb, na = 2000, N_ACT
    return nb, na, np.ascontiguousarray(synth.sym_matrix(7, nb)[:, :na]), np.ascontiguousarray(
        synth.sym_matrix(6, nb)[:, :na])


def test_ao2mo_synth_n2000_low_r_slab_full_tensor(be):
    """BASELINE configs[3] shape: the r-slab [3, 5) of the N_AO = 2000, n = 128 streamed transform,
    ALL n^4 elements, against a host direct sum over oracle/synth.py integrals:
    out = sum_{r in slab} sum_{s<=r} (C1^T E_rs C2)[i,j] (C3[r,k] C4[s,l] + [s<r] C3[s,k] C4[r,l])."""
    nb, na, c_a, c_b = _n2000_coeffs()
    r0, r1 = 3, 5
    p = np.arange(nb, dtype=np.uint64)[:, None]
    q = np.arange(nb, dtype=np.uint64)[None, :]
    ref = np.zeros((na, na, na, na))
    for r in range(r0, r1):
        for s in range(r + 1):
            e_rs = synth.val(synth.STREAM_ERI, synth.eri_canon(p, q, np.uint64(r), np.uint64(s))) * synth.eri_scale(nb)
            x = c_a.T @ e_rs @ c_a  # (i, j)
            kl = np.outer(c_b[r], c_b[s])
            if s < r:
                kl = kl + np.outer(c_b[s], c_b[r])
            ref += x[:, :, None, None] * kl[None, None]
    ca, cb = be.asarray(c_a), be.asarray(c_b)
    got = be.to_host(be.ao2mo_synth(nb, ca, ca, cb, cb, r0=r0, r1=r1))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * nb)


def test_ao2mo_synth_n2000_mid_r_slab_sampled(be):
    """The slab r = 1000 of the same transform (an average-cost slab: 1001 pairs s <= r, 4e9 integrals):
    64 sampled (i,j,k,l) against the C oracle's half transform of the sampled (i,j) pairs."""
    nb, na, c_a, c_b = _n2000_coeffs()
    r = 1000
    ij = [(0, 0), (1, 0), (5, 77), (77, 5), (127, 127), (64, 31), (100, 101), (13, 126)]
    kl = [(0, 0), (0, 1), (127, 3), (3, 127), (64, 64), (90, 17), (31, 126), (126, 125)]
    a = np.stack([c_a[:, i] for i, _ in ij])
    b = np.stack([c_a[:, j] for _, j in ij])
    y = cref.half_transform_rs(nb, r, a, b)  # (8, r+1): sum_pq C1[p,i] C2[q,j] (pq|rs), s <= r
    ca, cb = be.asarray(c_a), be.asarray(c_b)
    got = be.to_host(be.ao2mo_synth(nb, ca, ca, cb, cb, r0=r, r1=r + 1))
    for m, (i, j) in enumerate(ij):
        for k, l in kl:
            w = c_b[r, k] * c_b[: r + 1, l]
            w[:r] += c_b[:r, k] * c_b[r, l]
            want = float(y[m] @ w)
            assert abs(got[i, j, k, l] - want) < 1e-12 * nb, (i, j, k, l, got[i, j, k, l], want)
    # (ij|kl) = (ji|kl) holds for the partial sums too when C1 = C2 (up to rounding)
    np.testing.assert_allclose(got[5, 77], got[77, 5], rtol=0, atol=1e-12 * nb)


# ---------------------------------------------------------------- the fused SCF loop at the bench size
def test_fused_huzinaga_scf_bench_inputs_vs_oracle(be, eri148):
    """bench.py's workload (N_AO = 148, (33,33) occupied, 20 environment MOs, DIIS on, stopping rule
    off) for 25 cycles through the fused device loop -- packed J/K + Fock, fused Huzinaga operator,
    device DIIS, guarded then tracked eigensolve, scalars kernel -- against oracle.huzinaga.huzinaga_scf
    (nbed/scf/huzinaga_scf.py:93-206) with the C oracle's J/K: per-cycle energies and the final
    (eps, D, Hz) to north_star's 1e-8."""
    from nbed_amd.scf import GpuUHF, Mole, huzinaga_scf

    n, nocc, n_env, ncyc = N_BENCH, (33, 33), 20, 25
    pr = synth.problem(n, nocc, n_env)

    class CUHF(ToyUHF):
        def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
            jk = cref.jk(self._eri, np.asarray(dm))
            return jk[0] - jk[1:]

    ref = CUHF(ToyMol(n, pr["nelec"]), pr["S"], pr["hcore"], eri148)
    mf = GpuUHF(Mole(n, pr["nelec"]), pr["S"], pr["hcore"], be.synth_eri(n), backend=be)
    assert mf.eri_packed_device() is not None  # the packed J/K kernel is the one that runs
    ref.max_cycle = mf.max_cycle = ncyc
    ref.conv_tol = mf.conv_tol = -1.0  # never satisfied: exactly ncyc cycles on both sides
    rh, gh = [], []
    rc, re, rd, rhz, rconv = oracle_huz.huzinaga_scf(ref, pr["V_emb"], pr["D_env"], history=rh)
    c, e, d, hz, conv = huzinaga_scf(mf, pr["V_emb"], pr["D_env"], history=gh)
    assert not conv and not rconv and len(gh) == len(rh) == ncyc
    for i, ((ge, gd), (oe, od)) in enumerate(zip(gh, rh)):
        np.testing.assert_allclose(ge, oe, rtol=0, atol=1e-8, err_msg=f"energy of cycle {i}")
        assert abs(gd - od) < 1e-8, (i, gd, od)
    assert rh[-1][1] < 1e-6  # the run did converge the density (the stopping rule was only disabled)
    np.testing.assert_allclose(e, re, rtol=0, atol=1e-8)
    np.testing.assert_allclose(d, rd, rtol=0, atol=1e-8)
    np.testing.assert_allclose(hz, rhz, rtol=0, atol=1e-8)
    s = pr["S"]
    for x in range(2):
        assert abs(np.trace(d[x] @ s) - pr["nelec"][x]) < 1e-9
        assert abs(np.trace(d[x] @ s @ pr["D_env"][x] @ s)) < 1e-9


def _experimental_build() -> bool:
    """libnbx built with `make -C nbed_amd/csrc EXPERIMENTAL=1` (the J/K experiments of DESIGN.md section 9 linked in)."""
    from nbed_amd import _nbx

    return bool(_nbx.load_library().nbx_experimental())


needs_experimental = pytest.mark.skipif("not _experimental_build()",
                                        reason="the shipped libnbx.so holds no experiments: make EXPERIMENTAL=1")


@needs_experimental
def test_jk_eightfold_experimental_vs_c_oracle():
    """csrc/jk_p8.hip (8-fold tiles; opt-in, NBX_JK_P8=1 is read once per process: a child process) at
    N = 148 and 104 against oracle/c/jk_ref.c, the Fock epilogue included."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    env = dict(os.environ, NBX_JK_P8="1")
    out = subprocess.run([sys.executable, str(Path(__file__).with_name("_p8_worker.py"))], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "P8 OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@needs_experimental
@pytest.mark.parametrize("variant", ["1", "2", "3"])
def test_jk_lds_dma_experimental_vs_c_oracle(variant):
    """csrc/jk_s4d.hip (tiles streamed straight into LDS; opt-in) in a child process with
    NBX_JK_DMA set: parity with the C oracle at the bench size; variants 1 and 2 keep the production kernel's
    work distribution and give its numbers bit for bit (variant 3 runs three workgroups per CU: other partial
    sums, other rounding)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    outs = {}
    for v in (variant, "0") if variant != "3" else (variant,):
        env = dict(os.environ, NBX_JK_DMA=v)
        out = subprocess.run([sys.executable, str(Path(__file__).with_name("_s4d_worker.py"))], env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "S4D OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
        outs[v] = out.stdout.strip().splitlines()[-1].split(" ", 3)[3]
    if variant != "3":
        assert outs[variant] == outs["0"]


@pytest.mark.parametrize("m4", ["1", "0"])
def test_jk_packed_n148_both_kernels_vs_symmetric_kernel(m4):
    """N = 148 is served by csrc/jk_m4.hip (the walk on v_mfma_f64_4x4x4_4b_f64 over block-major tiles streamed into LDS by
    the load unit) unless NBX_JK_M4=0 hands it back to csrc/jk_s4.hip -- the switch is read once per process: a child
    process each -- both density counts and row slabs, against the symmetric kernel (itself held to the C oracle above)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    env = dict(os.environ, NBX_JK_M4=m4, PYTHONPATH=str(Path(__file__).resolve().parent.parent))
    out = subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "tools" / "time_jk_packed.py"), "148"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    errs = [float(ln.split("=")[3].split()[0]) for ln in out.stdout.splitlines() if "max|packed - sym|" in ln]
    assert len(errs) == 2 and max(errs) < 1e-13, out.stdout
