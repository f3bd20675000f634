"""Huzinaga-projected SCF on the GPU.

Drop-in for nbed/scf/huzinaga_scf.py: same function names, arguments, return values and
quirks:

* Loewdin orthogonalisation with X = S^-1/2 computed once (:128);
* plain DIIS on F starting at loop index 2 (:162-164) with pyscf.lib.diis semantics;
* the energy uses ``vhf`` of the PREVIOUS density with the NEW density (:181-185);
* ``E_prev`` starts at 0 and convergence takes the max over the spin components (:191);
* the returned Huzinaga operator is the last cycle's, built from the pre-DIIS Fock (:159).

Two execution paths, same arithmetic:

* fused: ``scf_method`` is a ``GpuUHF`` whose ``get_veff``/``get_hcore`` are not monkey
  patched -> J/K, Fock assembly, projector products, DIIS, eigensolve, density and the
  per-cycle scalars all stay in HBM and nothing in a cycle waits for the host.  The 4 doubles
  of the convergence test are copied back asynchronously and read one cycle late: cycle i+1
  is already queued when cycle i is judged, and is simply dropped if cycle i had converged
  (results are those of cycle i, exactly as in the sequential loop);
* generic: any object implementing the reference's SCF protocol (numpy ``get_veff`` ...):
  its ``get_veff``/``get_occ`` are called as the reference calls them, everything else
  (projector GEMMs, eigensolve, density, traces) still runs on the GPU.
"""

from __future__ import annotations

import logging
import os
from typing import Optional

import numpy as np

from ..backend import get_backend
from .diis import DIIS
from .gpu_scf import GpuUHF
from .pyscf_compat import is_hf, is_ks

logger = logging.getLogger(__name__)


def _as3(be, a):
    """Device array with a leading batch (spin) axis."""
    d = be.asarray(np.asarray(a))
    return d if d.dim() == 3 else d.reshape(1, *d.shape)


def get_huzinaga_operator(fock, dm_occ_S, dm_virt_S, backend=None):
    """Huzinaga operator -(kappa)[(F DS) + (F DS)^T] (+ virtual term), numpy in/out.

    nbed/scf/huzinaga_scf.py:65-90; kappa = 1/2 for 2-D input, 1 for 3-D (:80,88).
    """
    be = backend if backend is not None else get_backend()
    fock = np.asarray(fock)
    out = _huzinaga_device(be, _as3(be, fock), _as3(be, dm_occ_S),
                           None if dm_virt_S is None or not np.any(dm_virt_S) else _as3(be, dm_virt_S),
                           0.5 if fock.ndim == 2 else 1.0)
    out = be.to_host(out)
    return out[0] if fock.ndim == 2 else out


def _huzinaga_device(be, fock3, ds_occ3, ds_virt3, kappa, fock_io=None):
    """Hz on device; if ``fock_io`` is given it is updated in place (F += Hz)."""
    fds = be.gemm(fock3, ds_occ3)
    hz = be.huzinaga_sym(fds, kappa, fock_io)
    if ds_virt3 is not None:
        # -(kappa)[FDvS + (FDvS)^T - 2 (DvS)^T (F DvS)]   (huzinaga_scf.py:82-88)
        fdv = be.gemm(fock3, ds_virt3)
        hzv = be.huzinaga_sym(fdv, kappa, None)
        corr = be.gemm(ds_virt3, fdv, "T", "N")
        be.axpby(2.0 * kappa, corr, 1.0, hzv)
        be.axpby(1.0, hzv, 1.0, hz)
        if fock_io is not None:
            be.axpby(1.0, hzv, 1.0, fock_io)
    return hz


def calculate_hf_energy(scf_method, embedding_potential, density_matrix, vhf, huzinaga_op_occ, backend=None):
    """tr[(hcore + V_emb + vhf/2 + Hz) D] per spin (nbed/scf/huzinaga_scf.py:14-33)."""
    be = backend if backend is not None else get_backend()
    ham = np.asarray(scf_method.get_hcore()) + embedding_potential + 0.5 * np.asarray(vhf) + huzinaga_op_occ
    return be.trace_prod(be.asarray(ham), be.asarray(np.asarray(density_matrix)))


def calculate_ks_energy(scf_method, embedding_potential, density_matrix, huzinaga_op_occ, backend=None):
    """E_coul + E_xc + tr[D (hcore + Hz + V_emb)] (nbed/scf/huzinaga_scf.py:36-62).

    The XC part comes from the SCF object's own ``get_veff`` (``.ecoul``/``.exc``), exactly as
    in the reference; this package has no XC quadrature (out of scope).
    """
    be = backend if backend is not None else get_backend()
    vhf_updated = scf_method.get_veff(dm=density_matrix)
    energy = vhf_updated.ecoul + vhf_updated.exc
    ham = np.asarray(scf_method.get_hcore()) + huzinaga_op_occ + embedding_potential
    energy += be.trace_prod(be.asarray(np.asarray(density_matrix)), be.asarray(ham))
    return energy


def _is_fused(scf_method) -> bool:
    """GpuUHF with the class's own get_veff/get_hcore/get_occ (no instance monkey patches)."""
    if not isinstance(scf_method, GpuUHF) or is_ks(scf_method):
        return False
    d = vars(scf_method)
    return not any(k in d for k in ("get_veff", "get_hcore", "get_occ", "make_rdm1"))


MAX_REFINE_ITERS = 6  # the library's limit (nbx_eigh_warm_ex / nbx_geig_refine)


class History(list):
    """``history=`` argument of ``huzinaga_scf`` that also reports HOW the run went: ``info`` holds
    ``cycle_call`` (one C call per cycle: nbx_huz_cycle), ``split`` (J/K slabs + all-reduce) and ``restarts``
    (one reason per repetition of the whole SCF after a rejected purified / tracked cycle; normally empty)."""

    def __init__(self, *args):
        super().__init__(*args)
        self.info = {"restarts": []}


class _PurificationFailed(Exception):
    """A density taken from purification (initial guess or cycle) did not resolve a gap: the run is repeated
    with an eigensolver in every cycle."""


# purification (nbx_purify; nbx_huz_cycle mode 2): the density of the initial guess and of the cycles up to the
# first one whose density moved by less than PURIFY_DM_CHANGE (judged one cycle late, so the cold eigensolve that
# follows sees a change ~10x smaller and the cycles after it are settled by refinement); entered again when a
# guarded cycle needed at least PURIFY_MIN_SWEEPS Jacobi sweeps while the density still moved by more than that
PURIFY_MIN_SWEEPS = 6
PURIFY_DM_CHANGE = 0.2


class _TrackedEigensolveFailed(Exception):
    """A cycle solved by unguarded refinement (nbx_geig_refine) was not accepted."""


def huzinaga_scf(
    scf_method,
    embedding_potential: np.ndarray,
    dm_environment_occupied: np.ndarray,
    dm_environment_virtual: np.ndarray | None = None,
    dm_conv_tol: float = 1e-6,
    dm_initial_guess: Optional[np.ndarray] = None,
    use_DIIS: Optional[bool] = True,
    backend=None,
    history: list | None = None,
    callback=None,
):
    """Manual SCF with the Huzinaga projector; see the module docstring.

    Returns (mo_coeff, mo_energy, density_matrix, huzinaga_op, conv_flag) as numpy arrays
    (nbed/scf/huzinaga_scf.py:206).  Not in the reference's signature: ``backend``; ``history``
    collects (energy, dm_diff) per cycle; ``callback(i)`` is called on the host before cycle
    ``i`` is queued (bench.py uses it to separate warm-up cycles from timed ones).

    The fused device loop solves the eigenproblem of a cycle by refining the previous cycle's
    vectors with no fallback solver queued behind it once a cycle has shown that refinement is
    accepted at once ("tracked" cycles: 5 launches instead of 12).  Acceptance is checked on the
    host one cycle late like everything else; in the (not yet observed) case that a tracked cycle
    is rejected the whole run is repeated with the guarded solver, which gives the same numbers.
    """
    args = (scf_method, embedding_potential, dm_environment_occupied, dm_environment_virtual, dm_conv_tol,
            dm_initial_guess, use_DIIS, backend, history, callback)
    tracked = os.environ.get("NBED_TRACKED_EIG", "1") != "0"
    purify = os.environ.get("NBED_PURIFY", "1")
    # NBED_PURIFY: "1" (default) densities by purification from the first cycle on, until the density has settled
    # (and again whenever a guarded cycle shows that warm starts do not help); "0" never
    for _ in range(3):
        try:
            return _huzinaga_scf(*args, allow_tracked=tracked, allow_purify=purify)
        except _TrackedEigensolveFailed as exc:
            logger.warning("tracked eigensolve rejected a cycle: repeating the SCF with the guarded solver")
            tracked = False
            reason = f"tracked eigensolve rejected ({exc})"
        except _PurificationFailed as exc:
            logger.warning("purification cycle unusable (%s): repeating the SCF with an eigensolver in every cycle", exc)
            purify = "0"
            reason = f"purification unusable ({exc})"
        if history is not None:
            del history[:]
            if hasattr(history, "info"):
                history.info.setdefault("restarts", []).append(reason)
    return _huzinaga_scf(*args, allow_tracked=False, allow_purify="0")


def _huzinaga_scf(scf_method, embedding_potential, dm_environment_occupied, dm_environment_virtual, dm_conv_tol,
                  dm_initial_guess, use_DIIS, backend, history, callback, allow_tracked, allow_purify="0"):
    if not (is_ks(scf_method) or is_hf(scf_method)):
        raise TypeError("Cannot run Huzinaga SCF with type %s" % type(scf_method))
    be = backend if backend is not None else (getattr(scf_method, "be", None) or get_backend())
    fused = _is_fused(scf_method)
    ks = is_ks(scf_method)

    embedding_potential = np.asarray(embedding_potential)
    restricted = np.asarray(dm_environment_occupied).ndim == 2
    kappa = 0.5 if restricted else 1.0
    nb = 1 if restricted else 2

    s_h = np.asarray(scf_method.get_ovlp())
    s_d = scf_method._s_d if fused else be.asarray(s_h)
    if fused:
        x_d = scf_method.x_device()
    else:
        x_d = be.sym_pow_fast(s_d, -0.5, s_h) if hasattr(be, "sym_pow_fast") else be.sym_pow(s_d, -0.5)
    adiis = DIIS(be) if use_DIIS else None

    hcore_h = np.asarray(scf_method.get_hcore())
    if hasattr(be, "asarray_many"):  # one upload for the call's inputs
        up = [np.asarray(dm_environment_occupied), hcore_h, embedding_potential]
        if dm_environment_virtual is not None:
            up.append(np.asarray(dm_environment_virtual))
        up = be.asarray_many(up)
        dm_occ_d, hcore_d, vemb_d = up[:3]
        dm_virt_d = up[3] if dm_environment_virtual is not None else None
    else:
        dm_occ_d, hcore_d, vemb_d = (be.asarray(np.asarray(a)) for a in (dm_environment_occupied, hcore_h,
                                                                         embedding_potential))
        dm_virt_d = be.asarray(np.asarray(dm_environment_virtual)) if dm_environment_virtual is not None else None

    def batch3(d):
        return d if d.dim() == 3 else d.reshape(1, *d.shape)

    def overlap_per_spin():
        """S once per spin on the device (for the tracked solver): from the resident copy, not a second upload."""
        if hasattr(be, "torch") and isinstance(s_d, be.torch.Tensor):
            return be.torch.stack([s_d] * nb).contiguous()
        return be.asarray(np.stack([s_h] * nb))

    ds_occ = be.gemm(batch3(dm_occ_d), s_d)
    ds_virt = None
    if dm_virt_d is not None:
        ds_virt = be.gemm(batch3(dm_virt_d), s_d)

    # h + V_emb broadcast to the batch shape (:139,157)
    hv = be.zeros((nb,) + s_h.shape)
    for x in range(nb):
        be.axpby(1.0, hcore_d if hcore_d.dim() == 2 else hcore_d[x], 1.0, hv[x])
        be.axpby(1.0, vemb_d if vemb_d.dim() == 2 else vemb_d[x], 1.0, hv[x])

    warm = {"v": None, "iters": MAX_REFINE_ITERS, "tracked": False, "c": None,
            "purify": allow_purify != "0", "pur_iters": 0}
    s_b = None  # the overlap once per spin, for the tracked solver

    def diagonalise(fock3):
        nonlocal s_b
        if warm["tracked"] and warm["c"] is not None:
            # refine (eps, C) of the previous cycle on the pencil (F, S) directly
            if s_b is None:
                s_b = overlap_per_spin()
            e_d, c_new = be.geig_refine(fock3, s_b, warm["c"], refine_iters=warm["iters"])
            warm["c"] = c_new
            return e_d, c_new
        # Loewdin step (:166-169).  The previous cycle's orthonormal eigenvectors seed the
        # solver: X F X is nearly diagonal in that basis once the SCF is under way.
        fo = be.gemm(be.gemm(x_d, fock3), x_d)
        if lookahead:
            e_d, c_ortho = be.eigh(fo, v0=warm["v"], refine_iters=warm["iters"])
        else:
            e_d, c_ortho = be.eigh(fo, v0=warm["v"])
        warm["v"] = c_ortho
        warm["c"] = be.gemm(x_d, c_ortho)
        return e_d, warm["c"]

    # fused path: the eigensolver returns ascending eigenvalues, so aufbau occupation
    # (get_occ: the n_alpha / n_beta lowest, as the reference's UHF object does) is a fixed
    # vector and the MO energies need not leave the device inside the loop
    lookahead = (fused and (not ks) and nb == 2 and hasattr(be, "huz_cycle_scalars_async")
                 and hasattr(be, "density_occ"))

    can_track = lookahead and hasattr(be, "geig_refine")
    # everything the loop will need from the host goes up now: an upload inside the loop waits for
    # the cycles queued ahead of it and leaves the GPU idle while the host catches up
    if can_track and allow_tracked:
        s_b = overlap_per_spin()
    if adiis is not None and hasattr(adiis, "reserve"):
        adiis.reserve(nb * s_h.shape[0] * s_h.shape[1])

    def occupations(e_d, c_d):
        if lookahead:
            return e_d, None  # density() takes the leading n_alpha / n_beta columns
        e_h = be.to_host(e_d)
        if fused:
            return e_h, scf_method.get_occ(e_h)
        c_h = be.to_host(c_d)
        if restricted:
            return e_h, scf_method.get_occ(e_h[0], c_h[0])[None]
        return e_h, scf_method.get_occ(e_h, c_h)

    def density(c_d, occ_h):
        if lookahead:  # occupied columns are the leading ones: D = C_occ C_occ^T, no copy/scale
            return be.density_occ(c_d, scf_method.mol.nelec)
        scaled = be.scale_cols(be.copy(c_d), be.asarray(np.asarray(occ_h, dtype=np.float64)))
        return be.gemm(scaled, c_d, "N", "T")

    def unbatch(a_h):
        return a_h[0] if restricted else a_h

    dts_d, dts_ready = None, False
    sh = getattr(scf_method, "shards", None)
    split = sh is not None and (sh.world > 1 or sh.force_collective)  # J/K slabs summed by an all-reduce
    if lookahead and hasattr(scf_method, "dts_device") and scf_method.fused_fock_available(hv):
        dts_d = scf_method.dts_device()
    # One C call per cycle (nbx_huz_cycle) instead of 12-40 marshalled launches: the first cycles of an
    # SCF are host bound otherwise (the GPU idles 100-300 us per cycle until the eigensolver settles).
    # Same kernels, order and operands as the step-by-step path below: bit-identical results.  Every size
    # (packed J/K kernel, or the symmetric one on the dense tensor) and every world size: over several ranks
    # the cycle is two calls around the all-reduce of the J/K partials (nbx_huz_cycle_jk / _post).
    use_cycle_call = (lookahead and ds_virt is None and hasattr(be, "huz_cycle") and sh is not None
                      and os.environ.get("NBED_CYCLE_CALL", "1") != "0")
    if not use_cycle_call:
        warm["purify"] = False  # (nbx_huz_cycle mode 2 only: the step-by-step path always solves the eigenproblem)
    if history is not None and hasattr(history, "info"):
        history.info.update(cycle_call=bool(use_cycle_call), split=bool(split))

    # ---- initial guess from the projected core Hamiltonian (:139-148)
    guess_status = None
    if dm_initial_guess is None:
        fock = be.copy(hv)
        _huzinaga_device(be, fock, ds_occ, ds_virt, kappa, fock_io=fock)
        if warm["purify"]:
            # only the density of the guess is used (:148): the projector on its occupied levels by purification
            # (0.4 ms) instead of a cold Jacobi eigensolve (3 ms at N = 148); verdict read with cycle 0's
            p_d, pst = be.purify(be.gemm(be.gemm(x_d, fock), x_d), scf_method.mol.nelec)
            dm_d = be.gemm(be.gemm(x_d, p_d), x_d)
            guess_status = be.async_to_host(pst)
        else:
            e_d, c_d = diagonalise(fock)
            _, occ_h = occupations(e_d, c_d)
            dm_d = density(c_d, occ_h)
    else:
        dm_d = _as3(be, dm_initial_guess)

    conv_flag = False
    scf_energy_prev = 0
    mo_energy_h = None
    hz = None
    c_d = None
    pending = None
    result = {"state": None}

    def judge(state):
        """Convergence test of a queued cycle (:186-194); records it as the current result."""
        nonlocal conv_flag, scf_energy_prev
        nonlocal guess_status
        cycle, handle = state[0], state[1]
        sc = handle.get()
        if guess_status is not None:
            gst, guess_status = guess_status.get(), None
            if np.any(gst <= 0):
                raise _PurificationFailed(f"initial guess: status {gst.tolist()}")
        # launch-count policy for the eigensolver (results do not depend on it): once a cycle's
        # matrices were all accepted after ONE refinement iteration, queue only one from now on
        # (anything it does not finish falls through to Jacobi on the device); otherwise three
        st = handle.get_extra()
        purified = len(state) > 7 and state[7]
        if st is not None and purified:
            logger.debug("cycle %s purification steps %s", cycle, st.tolist())
            if np.any(st <= 0):
                raise _PurificationFailed(f"cycle {cycle}: status {st.tolist()}")
            warm["pur_iters"] = int(np.max(st)) + 8
            if float(np.max(sc[2:])) < PURIFY_DM_CHANGE:
                warm["purify"] = False  # the next cycle queued solves the eigenproblem (cold: there are no vectors)
        elif st is not None:
            logger.debug("cycle %s eigensolver status %s tracked=%s", cycle, st.tolist(), state[6])
            if state[6] and np.any(st <= 0):
                raise _TrackedEigensolveFailed(f"cycle {cycle}: status {st.tolist()}")
            if (allow_purify != "0" and not state[6] and np.all(st > 0) and np.all(st < 1000)
                    and int(np.max(st)) >= PURIFY_MIN_SWEEPS and float(np.max(sc[2:])) > PURIFY_DM_CHANGE):
                warm["purify"] = True  # warm starts are not helping: densities by purification until they would
            # every matrix accepted by refinement within two iterations: from the next cycle queued
            # on, refine without the guard -- with one iteration in reserve while the density still
            # moves; anything else (re-)arms the guarded solver
            accepted = bool(np.all(st >= 1001))
            needed = int(np.max(st)) - 1000 if accepted else 99
            warm["tracked"] = bool(allow_tracked and can_track and needed <= 2 and not warm["purify"])
            if warm["tracked"]:
                warm["iters"] = needed + (1 if float(np.max(sc[2:])) > 1e-8 else 0)
            else:  # guarded: refinement converges quadratically from max|E| < 0.25, so up to six
                # iterations (25 us each) are worth queueing ahead of a ~1 ms Jacobi solve
                warm["iters"] = 1 if needed == 1 else (min(MAX_REFINE_ITERS, needed + 1) if accepted
                                                       else MAX_REFINE_ITERS)
        scf_energy = sc[:2].copy()
        norm_dm_diff = float(np.max(sc[2:]))
        run_diff = np.max(np.abs(scf_energy - scf_energy_prev))
        if history is not None:
            history.append((np.array(scf_energy, copy=True), norm_dm_diff))
        result["state"] = state
        if (run_diff < scf_method.conv_tol) and (norm_dm_diff < dm_conv_tol):
            conv_flag = True
            logger.debug("Huzinaga SCF converged in cycle %s", cycle)
            return True
        scf_energy_prev = scf_energy
        return False

    if use_cycle_call:
        if s_b is None:
            s_b = overlap_per_spin()
        packed_d = scf_method.eri_packed_device()
        hstate = be.huz_cycle_state(s_h.shape[0], scf_method.mol.nelec, packed_d, hv, ds_occ, s_b, x_d, dts_d,
                                    eri=None if packed_d is not None else scf_method.eri_device(), p0=sh.lo, p1=sh.hi)
        reduce = (lambda jk: sh.all_reduce(be, jk)) if split else None
        diis_state = {"first": True, "head": 0, "nd": 0, "space": 6}
        for i in range(scf_method.max_cycle):
            if callback is not None:
                callback(i)
            out = hstate.sets[i % 3]
            tracked_now = bool(warm["tracked"] and warm["c"] is not None)
            purify_now = bool(warm["purify"] and not tracked_now and allow_purify != "0")
            diis_mode = diis_slot = diis_nd = 0
            if use_DIIS and i > 1:  # pyscf.lib.diis bookkeeping (scf/diis.py): first update only remembers F
                if diis_state["first"]:
                    diis_state["first"] = False
                    diis_mode = 1
                else:
                    if diis_state["head"] >= diis_state["space"]:
                        diis_state["head"] = 0
                    diis_slot = diis_state["head"]
                    diis_state["head"] += 1
                    diis_state["nd"] = min(diis_state["nd"] + 1, diis_state["space"])
                    diis_mode, diis_nd = 2, diis_state["nd"]
            if purify_now:  # density by purification: this cycle has no orbitals, X F X is left in out["v"]
                pending_now = be.huz_cycle(hstate, dm_d, None, out, 2, warm["pur_iters"], diis_mode, diis_slot, diis_nd,
                                           dts_ready, reduce=reduce)
                warm["c"] = warm["v"] = None
            else:
                c_in = warm["c"] if tracked_now else warm["v"]
                pending_now = be.huz_cycle(hstate, dm_d, c_in, out, tracked_now, warm["iters"], diis_mode, diis_slot,
                                           diis_nd, dts_ready, reduce=reduce)
                warm["c"] = out["c"]
                if not tracked_now:
                    warm["v"] = out["v"]
            dts_ready = dts_d is not None
            dm_d, hz, c_d, mo_energy_h = out["dm"], out["hz"], out["c"], out["w"]
            state_now = (i, pending_now, out["c"], out["w"], out["dm"], out["hz"], tracked_now, purify_now, out["v"])
            if pending is not None and judge(pending):
                break
            pending = state_now

    for i in range(0 if use_cycle_call else scf_method.max_cycle):
        if callback is not None:
            callback(i)
        # ---- Fock build (:156-160)
        if fused:
            if nb == 2 and hasattr(scf_method, "fock_device"):
                fock, vhf = scf_method.fock_device(dm_d, hv, dts_ready=dts_ready)
            else:
                fock, vhf = be.fock_uhf(hv, None, scf_method.jk_device(dm_d))
        else:
            vhf_h = scf_method.get_veff(dm=unbatch(be.to_host(dm_d)))
            vhf = _as3(be, np.asarray(vhf_h))
            fock = be.copy(hv)
            be.axpby(1.0, vhf, 1.0, fock)
        if ds_virt is None and hasattr(be, "huzinaga_fused"):
            hz, fock = be.huzinaga_fused(fock, ds_occ, kappa)  # product + symmetrisation + F += Hz
        else:
            hz = _huzinaga_device(be, fock, ds_occ, ds_virt, kappa, fock_io=fock)

        if use_DIIS and (i > 1):
            fock = adiis.update(fock)

        tracked_now = bool(warm["tracked"] and warm["c"] is not None)
        e_d, c_d = diagonalise(fock)
        mo_energy_h, occ_h = occupations(e_d, c_d)
        dm_old = dm_d
        dm_d = density(c_d, occ_h)

        # ---- energy and convergence scalars (:176-194)
        if ks:
            scf_energy = calculate_ks_energy(
                scf_method, embedding_potential, unbatch(be.to_host(dm_d)), unbatch(be.to_host(hz)), backend=be
            )
            diff = be.copy(dm_d)
            be.axpby(-1.0, dm_old, 1.0, diff)
            norm_dm_diff = float(np.max(np.sqrt(be.trace_prod(diff, be.transpose(diff)))))
        elif lookahead:
            scf_energy = norm_dm_diff = None
            # (the scalars kernel reads D anyway: it also leaves the Dtot' table of the next build)
            pending_now = be.huz_cycle_scalars_async(hv, None, vhf, hz, dm_d, dm_old,
                                                     extra=getattr(be, "last_eigh_status_d", None), dts=dts_d)
            dts_ready = dts_d is not None
        elif nb == 2:
            sc = be.huz_cycle_scalars(hv, None, vhf, hz, dm_d, dm_old)
            scf_energy = sc[:2].copy()
            norm_dm_diff = float(np.max(sc[2:]))
        else:
            ham = be.copy(hv)
            be.axpby(0.5, vhf, 1.0, ham)
            be.axpby(1.0, hz, 1.0, ham)
            scf_energy = float(be.trace_prod(ham, dm_d)[0])
            diff = be.copy(dm_d)
            be.axpby(-1.0, dm_old, 1.0, diff)
            norm_dm_diff = float(np.sqrt(be.trace_prod(diff, be.transpose(diff))[0]))

        if lookahead:
            # judge the PREVIOUS cycle now that this one is queued behind it
            state_now = (i, pending_now, c_d, mo_energy_h, dm_d, hz, tracked_now)
            if pending is not None and judge(pending):
                break
            pending = state_now
            continue

        run_diff = np.max(np.abs(scf_energy - scf_energy_prev))
        if history is not None:
            history.append((np.array(scf_energy, copy=True), norm_dm_diff))
        if (run_diff < scf_method.conv_tol) and (norm_dm_diff < dm_conv_tol):
            conv_flag = True
            logger.debug("Huzinaga SCF converged in cycle %s", i)
            break
        scf_energy_prev = scf_energy

    if lookahead:
        if not conv_flag and pending is not None:
            judge(pending)  # the last cycle queued
        if result["state"] is not None:
            _, _, c_d, mo_energy_d, dm_d, hz = result["state"][:6]
            if len(result["state"]) > 7 and result["state"][7]:
                # the run ended on a purified cycle: its orbitals now, from the X F X it left behind (:166-169)
                mo_energy_d, v_d = be.eigh(result["state"][8])
                c_d = be.gemm(x_d, v_d)
            mo_energy_h = mo_energy_d  # (still on the device: it travels with the other results below)

    if hasattr(be, "to_host_many") and all(isinstance(a, be.torch.Tensor) for a in (c_d, mo_energy_h, dm_d, hz)):
        # the results start on their way to the host (one launch) before the warning the reference logs at this
        # point (:203-204) is formatted and written: 0.1 ms of host work that the transfer hides
        results = be.to_host_many([c_d, mo_energy_h, dm_d, hz], wait=False)
        if conv_flag is False:
            logger.warning("Huzinaga SCF has NOT converged.")
        c_h, mo_energy_h, dm_h, hz_h = results.get()
    else:
        if conv_flag is False:
            logger.warning("Huzinaga SCF has NOT converged.")
        c_h, mo_energy_h, dm_h, hz_h = (be.to_host(a) for a in (c_d, mo_energy_h, dm_d, hz))
    return unbatch(c_h), unbatch(mo_energy_h), unbatch(dm_h), unbatch(hz_h), conv_flag
