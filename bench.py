#!/usr/bin/env python
"""bench.py -- embedded-SCF cycles/s (+ active-ERI transform GFLOP/s) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one cycle of the product's own ``nbed_amd.scf.huzinaga_scf`` (fused GPU path;
SURVEY.md section 3.2), i.e. one Huzinaga-projected UHF SCF cycle of the hot path:
J/K build from the dense (pq|rs) in HBM, Fock assembly, Huzinaga projector products, DIIS
extrapolation, Loewdin orthogonalisation, symmetric eigensolve for both spins, density and
the per-cycle energy/convergence scalars.  The stopping rule is disabled (conv_tol < 0), so the
timed call runs exactly K cycles (plus the call's O(N^3) set-up and the final result download).

Workload (default): synthetic tensors shaped like BASELINE.json configs[2] (octane/6-31G*,
the largest single-GPU configuration): N_AO = 148, (33,33) occupied, 20 environment MOs
projected out -> (13,13) active electrons, n = 128 embedded MOs for the transform.
The N_AO = 2000 headline config does not fit one GPU (a dense (pq|rs) is 128 TB).

After the timed SCF cycles the same run times the AO->active-MO four-index transform
(3 unique spin blocks of HamiltonianBuilder, nbed/ham_builder.py:119-124) and reports it
under "transform".  rank 0 prints ONE JSON line.

Multi-GPU (strong scaling, fixed problem): the first AO index of (pq|rs) is sharded over
the ranks (each rank generates only its slab); every cycle all-gathers the J/K row slabs
(RCCL); the transform shards the outer MO index and all-gathers the (ij|kl) slabs.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (datasheet); measured, registers only (profiles/r03/fp64_rate_probe.txt):
#                               v_mfma_f64_4x4x4_4b_f64 75, v_mfma_f64_16x16x4_f64 38-48 TFLOP/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nao", type=int, default=148)
    ap.add_argument("--nocc", type=int, default=33)
    ap.add_argument("--nenv", type=int, default=20)
    ap.add_argument("--nact", type=int, default=128, help="embedded MOs kept for the transform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-transform", action="store_true")
    ap.add_argument("--no-n2000", action="store_true", help="skip the bounded N_AO=2000 streamed sample")
    ap.add_argument("--no-tts", action="store_true", help="skip the cold-start time-to-solution run")
    ap.add_argument("--no-mu", action="store_true", help="skip the mu-shift projector's SCF leg")
    ap.add_argument("--no-real", action="store_true", help="skip the real-molecule leg (octane / 6-31G*, nothing injected)")
    ap.add_argument("--no-small", action="store_true", help="skip the small-config legs (BASELINE configs[0], [1], [4] shape)")
    ap.add_argument("--no-scaling", action="store_true", help="skip the scaling workloads (N_AO = 256 packed, 384 symmetric)")
    ap.add_argument("--no-df", action="store_true", help="skip the density-fitted J/K build at N_AO = 2000 (an extra: the GEMM-shaped J/K)")
    ap.add_argument("--df-naux", type=int, default=4000, help="auxiliary functions of the N_AO = 2000 density-fitted build (128 GB at 4000)")
    ap.add_argument("--n2000-rslabs", type=int, default=4, help="r-slabs of the N_AO=2000 transform per rank")
    ap.add_argument("--jk-event-every", type=int, default=4,
                    help="HIP events bracket the J/K kernel of one timed cycle in this many (1 = every cycle)")
    ap.add_argument("--cpu-cycles", type=int, default=8)
    return ap.parse_args()


def host_cores() -> int:
    """Cores this process may run on (the GPU box gives a share of the host, not all of it)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota, if the box sets one
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    # a one-GPU box's CPU share is 16 cores; never start more worker threads than that
    return max(1, min(n, int(os.environ.get("NBED_BENCH_CORES", "16"))))


def transform_flops(N, n):
    """One spin block, sequential quarter transforms, no symmetry."""
    return 2.0 * n * N**4 + 2.0 * n**2 * N**3 + 2.0 * n**3 * N**2 + 2.0 * n**4 * N


def gemm_traffic(N, n_act):
    """HBM bytes of the quarter-1 GEMM launch from the committed PMC passes (profiles/r03/gemm_traffic.json:
    FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes), if they were taken at this shape."""
    tfile = REPO / "profiles" / "r03" / "gemm_traffic.json"
    try:
        tj = json.loads(tfile.read_text())
        if f"N_AO={N}, n_act={n_act}:" in tj.get("workload", ""):
            return tj.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def cpu_baseline_cycle(pr, eri_h, ncycles):
    """The CPU oracle's cycle on the host cores: C/OpenMP one-pass J/K + numpy/LAPACK rest.
    ``pr``: dict with nao, nelec, S, hcore, V_emb, D_env (host arrays)."""
    import tempfile

    os.environ["OMP_NUM_THREADS"] = str(host_cores())  # libgomp reads it when the C oracle is loaded
    from oracle import cref
    from oracle.huzinaga import huzinaga_scf
    from oracle.pyscf_like import ToyMol, ToyUHF

    lib = _CPU_LIB.get("lib")
    if lib is None:
        try:
            lib = cref.load(cref.build(tempfile.mkdtemp(prefix="jkref_")))
        except Exception:
            lib = cref.load()
        _CPU_LIB["lib"] = lib

    class CUHF(ToyUHF):
        def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
            jk = cref.jk(self._eri, np.asarray(dm), lib=lib)
            return jk[0] - jk[1:]

    mf = CUHF(ToyMol(pr["nao"], pr["nelec"]), pr["S"], pr["hcore"], eri_h)
    mf.max_cycle = ncycles
    mf.conv_tol = 0.0  # never "converged": exactly ncycles cycles
    t0 = time.perf_counter()
    huzinaga_scf(mf, pr["V_emb"], pr["D_env"], use_DIIS=True)
    dt = time.perf_counter() - t0
    return ncycles / dt


_CPU_LIB: dict = {}


def df_jk_leg(be, args, sync):
    """An EXTRA, not on the parity path of the exact integrals: one J/K build at N_AO = 2000 from a three-index factor
    (pq|rs) ~ sum_L B_L[pq] B_L[rs] held in HBM (nbx_jk_df) -- the only form of J/K that is
    GEMM shaped, and the only one a single GPU can hold at this size (the dense tensor is 128 TB)."""
    import torch

    from nbed_amd import synth

    N, naux, nocc = 2000, int(args.df_naux), (512, 512)
    need = (naux * N * N + 2 * 64 * max(nocc) * N + 8 * N * N) * 8.0
    free, _total = torch.cuda.mem_get_info()
    if need > 0.9 * free:
        return {"skipped": f"needs {need / 1e9:.0f} GB of HBM, {free / 1e9:.0f} GB free"}
    t0 = time.perf_counter()
    b = be.df_synth(N, 0, naux)
    sync()
    t_gen = time.perf_counter() - t0
    _w, c1 = be.eigh(be.asarray(synth.sym_matrix(11, N)))  # orthonormal columns
    c = torch.stack([c1, torch.flip(c1, dims=[1])]).contiguous()
    be.jk_df(b, c, nocc)  # first touch of the workspace
    sync()
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        jk = be.jk_df(b, c, nocc)
    sync()
    dt = (time.perf_counter() - t0) / reps
    flops = 4.0 * sum(nocc) * float(N) ** 2 * naux  # per spin: 2 nocc N^2 naux (B_L C) + 2 nocc N^2 naux (Y Y^T)
    check = float(torch.linalg.norm(jk[0] - jk[0].T) / torch.linalg.norm(jk[0]))
    out = {
        "workload": f"N_AO={N}, N_aux={naux}, n_occ={nocc}: synthetic three-index factor B (N_aux, N, N) in HBM "
                    f"({naux * N * N * 8 / 1e9:.0f} GB, generated on the device in {t_gen * 1e3:.0f} ms), orthonormal orbitals",
        "note": "EXTRA: the GEMM-shaped J/K (density fitting); the exact-integral path of the reference is the streamed one above",
        "ms_per_build": dt * 1e3,
        "jk_df_tflops": flops / dt / 1e12,
        "frac_of_fp64_mfma_peak": flops / dt / 1e12 / FP64_MFMA_PEAK_TFLOPS,
        "flop_count": "8 n_occ N^2 N_aux: B_L C and (B_L C)(B_L C)^T per spin; J is two streaming passes on top",
        "factor_bytes_read_per_build": 3.0 * naux * N * N * 8,
        "j_symmetry_defect": check,
    }
    # ---- one FULL Huzinaga UHF cycle at this size on the factor that is resident (nbed/scf/huzinaga_scf.py:154-201):
    # J/K from the occupied orbitals (nbx_jk_df), F = h + V_emb + J - K, Hz = -(F DS + (F DS)^T), F += Hz, DIIS,
    # X F X, eigensolve (both spins as a batch), C = X C', aufbau density, energy and |dD| -- every step a libnbx call,
    # step by step (no fused entry point exists for the density-fitted build); three cycles: the first with a COLD
    # eigensolve (csrc/eigh_grid.hip), the others warm-started from the cycle before.
    try:
        out["n2000_cycle"] = n2000_cycle(be, b, N, nocc, sync)
    except Exception as exc:
        out["n2000_cycle"] = {"error": f"{type(exc).__name__}: {exc}"}
    del b, c, jk
    be.release_workspaces()
    torch.cuda.empty_cache()
    return out


def n2000_cycle(be, b, N, nocc, sync):
    import torch

    from nbed_amd import synth

    pr = synth.problem(be, N, nocc, N // 12)
    nocc = tuple(int(x) for x in pr["nelec"])  # the embedded system's electrons (the environment's are projected out)
    s_d = be.asarray(pr["S"])
    hv = be.asarray(np.asarray(pr["hcore"])[None] + np.asarray(pr["V_emb"]))
    x_d = be.sym_pow_fast(s_d, -0.5, pr["S"]) if hasattr(be, "sym_pow_fast") else be.sym_pow(s_d, -0.5)
    ds = be.gemm(be.asarray(pr["D_env"]), s_d)  # D_env S (huzinaga_scf.py:132)
    n2 = N * N
    space = 6
    xs, es = be.zeros((space, 2 * n2)), be.zeros((space, 2 * n2))
    hm = np.zeros((space + 1, space + 1))
    hm[0, 1:] = hm[1:, 0] = 1  # (the border of the Pulay matrix: pyscf.lib.diis)
    hmat, coef = be.asarray(hm), be.diis_coef_buffer(space)
    xprev = be.zeros((2, N, N))
    x2 = torch.stack([x_d, x_d]).contiguous()
    # core guess: orbitals of X (h + V_emb) X
    fo = be.gemm(be.gemm(x2, hv), x2)
    sync()
    w, v = be.eigh(fo)
    c = be.gemm(x2, v)
    dm = be.density_occ(c, nocc)
    sync()
    times, phases = [], []
    v_prev = None
    e_hist = []
    for cyc in range(3):
        sync()
        t0 = time.perf_counter()
        jk = be.jk_df(b, c, nocc)
        sync()
        t_jk = time.perf_counter() - t0
        fock, vhf = be.fock_uhf(hv, None, jk)
        hz, f2 = be.huzinaga_fused(fock, ds, 1.0)
        if cyc == 0:  # (pyscf.lib.diis: the first update only remembers F)
            xprev.copy_(f2)
            f_use = f2
        else:
            f_use = be.diis_update(space, (cyc - 1) % space, min(cyc, space), f2.reshape(-1), xprev.reshape(-1), xs, es, hmat,
                                   coef).reshape(2, N, N)
        fo = be.gemm(be.gemm(x2, f_use), x2)
        sync()
        t1 = time.perf_counter()
        w, v = be.eigh(fo, v0=v_prev)
        sync()
        t_eig = time.perf_counter() - t1
        c = be.gemm(x2, v)
        dm_new = be.density_occ(c, nocc)
        # E_x = tr[(h + V_emb + vhf/2 + Hz)_x D_x], |dD|_F (huzinaga_scf.py:181-194; the fused scalars kernel's scratch
        # is sized for N <= 1024: here the same sums by nbx_axpby / nbx_trace_prod / nbx_dots)
        ham = be.copy(hv)
        be.axpby(0.5, vhf, 1.0, ham)
        be.axpby(1.0, hz, 1.0, ham)
        sc = be.trace_prod(ham, dm_new)
        be.axpby(-1.0, dm_new, 1.0, dm)
        dd = float(np.sqrt(be.dots(dm.reshape(-1), dm.reshape(1, -1))[0]))
        sync()
        dt = time.perf_counter() - t0
        times.append(dt)
        phases.append({"jk_df_ms": t_jk * 1e3, "eigensolve_ms": t_eig * 1e3, "rest_ms": (dt - t_jk - t_eig) * 1e3})
        e_hist.append([float(sc[0]), float(sc[1]), dd])
        dm, v_prev = dm_new, v
    # (the checks too on libnbx's GEMM: no kernel of another library in this process's profile)
    res = float((be.gemm(fo, v) - v * w[:, None, :]).abs().max() / fo.abs().max())
    orth = float((be.gemm(v, v, "T", "N") - torch.eye(N, dtype=v.dtype, device=v.device)).abs().max())
    return {"workload": f"one Huzinaga UHF cycle at N_AO={N}, n_occ={nocc}: density-fitted J/K on the resident factor + Fock + "
                        "Huzinaga operator + DIIS + X F X + batched eigensolve + density + energy, libnbx calls step by step",
            "cold_cycle_ms": times[0] * 1e3, "warm_cycle_ms": times[-1] * 1e3, "cycles_ms": [t * 1e3 for t in times],
            "breakdown": phases, "eigen_residual_last_relative": res, "eigenvector_orthonormality_defect": orth,
            "eigenvalue_range_last": [float(w.min()), float(w.max())], "energy_alpha_beta_and_dm_change_per_cycle": e_hist}


def timed_huzinaga_run(mf, emb_args, kw, warmup, steps, sync):
    """The bench protocol on one SCF object: ONE huzinaga_scf run of warmup + steps cycles with the stopping rule
    off and DIIS on; the clock starts (GPU drained) when cycle ``warmup`` is about to be queued and stops when the
    run has returned its results to the host.  A run that the loop repeats (rejected purified / tracked cycle) is
    timed from its LAST start and reported (``scf_restarts``)."""
    from nbed_amd.scf import History, huzinaga_scf

    mf.conv_tol, mf.max_cycle = -1.0, warmup + steps
    hist, clock, stamps = History(), {}, []

    def on_cycle(i):
        if i == 0:
            del stamps[:]  # (a repeated run starts over: its stamps must not mix with the discarded attempt's)
        stamps.append(time.perf_counter())
        if i == warmup:
            sync()
            clock["t0"] = time.perf_counter()

    huzinaga_scf(mf, *emb_args, use_DIIS=True, history=hist, callback=on_cycle, **kw)
    sync()
    dt = time.perf_counter() - clock["t0"]
    diffs = np.diff(stamps)
    return {"cycles_per_sec": steps / dt, "ms_per_cycle": dt / steps * 1e3,
            "ms_per_cycle_settled": float(np.median(diffs[-8:])) * 1e3 if len(diffs) else None,
            "queue_pace_ms_first_cycles": [round(float(x) * 1e3, 3) for x in diffs[:12]],
            "scf_restarts": list(hist.info.get("restarts", [])), "one_call_per_cycle": bool(hist.info.get("cycle_call")),
            "energy_last_cycle": [float(x) for x in hist[-1][0]]}


def cpu_baseline_mu_cycle(pr, eri_h, h3, ncycles):
    """The CPU oracle's mu-shift cycle on the host cores: oracle.pyscf_like.ToyUHF.kernel() (scf.hf.kernel: CDIIS,
    scipy generalised eigh, energy and gradient) with the C/OpenMP one-pass J/K behind get_veff, on the patched
    hcore ``h3``; exactly ``ncycles`` cycles."""
    cpu_baseline_cycle  # (the C library is loaded by the Huzinaga leg: same helper)
    from oracle import cref
    from oracle.pyscf_like import ToyMol, ToyUHF

    lib = _CPU_LIB.get("lib")
    if lib is None:
        import tempfile

        os.environ["OMP_NUM_THREADS"] = str(host_cores())
        try:
            lib = cref.load(cref.build(tempfile.mkdtemp(prefix="jkref_")))
        except Exception:
            lib = cref.load()
        _CPU_LIB["lib"] = lib

    class CUHF(ToyUHF):
        def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
            jk = cref.jk(self._eri, np.asarray(dm), lib=lib)
            return jk[0] - jk[1:]

    mf = CUHF(ToyMol(pr["nao"], pr["nelec"]), pr["S"], pr["hcore"], eri_h)
    mf.get_hcore = lambda *a: h3
    mf.max_cycle, mf.conv_tol = ncycles, -1.0
    t0 = time.perf_counter()
    mf.kernel()
    return ncycles / (time.perf_counter() - t0)


def mu_shift_leg(be, mf_inputs, pr, args, sync, mu=1.0e6):
    """The OTHER projector of north_star's sentence, the reference's default (nbed/config.py:110,128): the mu-shift
    SCF of nbed/driver.py:500-538 -- PySCF's scf.hf.kernel on the hcore patched with mu S D_env S + V_emb -- on the
    same inputs as ``value`` (configs[2] shape), by the same protocol: ONE kernel() run of warmup + steps cycles with
    the stopping rule off, CDIIS on; the clock runs from the moment cycle ``warmup`` is about to be queued until the
    run has returned its results.  One cycle = one nbx_mu_cycle call."""
    from nbed_amd.scf import GpuUHF, Mole

    N = int(pr["nao"])
    S, hcore = np.asarray(pr["S"]), np.asarray(pr["hcore"])
    h3 = hcore[None] + mu * (S @ np.asarray(pr["D_env"]) @ S) + np.asarray(pr["V_emb"])
    eri, packed, shards = mf_inputs
    steps, warmup = args.steps, args.warmup
    res = {}
    for attempt in ("first_call", "timed"):
        mf = GpuUHF(Mole(N, pr["nelec"]), S, hcore, eri, backend=be, shards=shards, eri_packed=packed)
        mf.get_hcore = lambda *a: h3
        mf.conv_tol, mf.max_cycle = -1.0, warmup + steps
        clock, stamps = {}, []

        def on_cycle(i):
            if i == 0:
                del stamps[:]
            stamps.append(time.perf_counter())
            if i == warmup:
                sync()
                clock["t0"] = time.perf_counter()

        mf.cycle_callback = on_cycle
        mf.kernel()
        sync()
        dt = time.perf_counter() - clock["t0"]
        diffs = np.diff(stamps)
        res = {"cycles_per_sec": steps / dt, "ms_per_cycle": dt / steps * 1e3,
               "ms_per_cycle_settled": float(np.median(diffs[-8:])) * 1e3 if len(diffs) else None,
               "queue_pace_ms_first_cycles": [round(float(x) * 1e3, 3) for x in diffs[:12]],
               "kernel_info": dict(mf.kernel_info), "e_tot": float(mf.e_tot)}
    # time to solution with PySCF's stopping rule (conv_tol = Nbed's convergence 1e-6; conv_check cycle included)
    mf = GpuUHF(Mole(N, pr["nelec"]), S, hcore, eri, backend=be, shards=shards, eri_packed=packed)
    mf.get_hcore = lambda *a: h3
    mf.conv_tol, mf.max_cycle = 1e-6, 50
    sync()
    t0 = time.perf_counter()
    mf.kernel()
    sync()
    dts = time.perf_counter() - t0
    res["time_to_solution"] = {"ms": dts * 1e3, "cycles": int(mf.cycles), "converged": bool(mf.converged),
                               "stopping_rule": "|dE| < 1e-6 and |g_orb|/sqrt(size) < 1e-3 (scf.hf.kernel, conv_tol = nbed/config.py:110), "
                                                "core-Hamiltonian guess, CDIIS, conv_check cycle included",
                               "e_tot": float(mf.e_tot), "kernel_info": dict(mf.kernel_info)}
    res["workload"] = (f"mu-shift SCF (nbed/driver.py:500-538), mu = {mu:g}, same N_AO={N} inputs as `value`: hcore + mu S D_env S "
                       "+ V_emb, GpuUHF.kernel() = scf.hf.kernel control flow, one nbx_mu_cycle call per cycle")
    res["steps"], res["warmup"] = steps, warmup
    return res, h3


def embedded_molecule_run(be, cfg, prov):
    """``nbed(cfg)`` end to end on the device with nothing injected; returns (driver, seconds, the arguments its
    embedded Huzinaga SCF was called with)."""
    import torch

    import nbed_amd.driver as drv_mod
    from nbed_amd import nbed

    seen = {}
    inner = drv_mod.huzinaga_scf

    def recording(scf_method, *a, **k):
        seen["mol"], seen["args"], seen["kwargs"] = scf_method.mol, a, dict(k)
        return inner(scf_method, *a, **k)

    drv_mod.huzinaga_scf = recording
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv = nbed(cfg, provider=prov, backend=be, hamiltonian_format="spatial")
        torch.cuda.synchronize()
        e2e = time.perf_counter() - t0
    finally:
        drv_mod.huzinaga_scf = inner
    seen["kw"] = {k: v for k, v in seen["kwargs"].items() if k not in ("history", "callback", "dm_conv_tol", "use_DIIS")}
    return drv, e2e, seen


def small_configs_leg(be, args) -> list:
    """BASELINE configs[0], [1] and the shape of [4] on REAL integrals (nothing injected): the embedded Huzinaga SCF
    of each molecule by the bench's protocol -- cycles/s of the product loop (one nbx_huz_cycle per cycle; N = 7,
    24, 72 run the symmetric J/K kernel on the dense tensor) -- beside the CPU oracle's loop on the same integrals,
    embedding potential and environment density."""
    import torch

    from nbed_amd import NbedConfig
    from nbed_amd.driver import BuiltinHFProvider

    water = "3\n\nO   0.0000  0.000  0.115\nH   0.0000  0.754  -0.459\nH   0.0000  -0.754  -0.459"
    ch3 = "4\n\nC 0 0 0\nH 1.079 0 0\nH -0.5395 0.9344 0\nH -0.5395 -0.9344 0"
    cases = [
        ("configs[0]: H2O / STO-3G, 1 active atom (tests/test_config.json), SPADE; Huzinaga SCF of the embedded system "
         "(the config's mu-shift run is the same molecule through GpuUHF.kernel())",
         dict(geometry=water, n_active_atoms=1, basis="sto-3g", projector="both")),
        ("configs[1]: H2O / cc-pVDZ, Huzinaga projector", dict(geometry=water, n_active_atoms=2, basis="cc-pvdz",
                                                              projector="huzinaga")),
        ("configs[4] shape: CH3+ / cc-pVTZ (72 AOs, f shells), run unrestricted; the radical itself raises in the "
         "reference's SPADE (ragged alpha/beta partition) and here",
         dict(geometry=ch3, n_active_atoms=2, basis="cc-pvtz", projector="huzinaga", charge=1)),
    ]
    out = []
    steps, warmup = max(args.steps, 40), args.warmup
    for label, kw in cases:
        try:
            cfg = NbedConfig(xc_functional="b3lyp", convergence=1e-8, localization="spade", virtual_localization="cl",
                             max_shells=4, max_hf_cycles=100, max_dft_cycles=100, **kw)
            prov = BuiltinHFProvider(be)
            drv, e2e, seen = embedded_molecule_run(be, cfg, prov)
            mf = prov.local_hf(cfg, seen["mol"], backend=be)
            run = timed_huzinaga_run(mf, seen["args"], seen["kw"], warmup, steps, torch.cuda.synchronize)
            ints = prov._integrals(cfg)
            n = int(ints["nao"])
            entry = {"workload": label, "nao": n, "active_electrons": [int(x) for x in seen["mol"].nelec],
                     "end_to_end_s": e2e, "e_rhf": float(drv.huzinaga["e_rhf"]), "steps": steps, "warmup": warmup,
                     "embedded_scf_cycles_per_sec": run["cycles_per_sec"], **{k: run[k] for k in (
                         "ms_per_cycle", "ms_per_cycle_settled", "scf_restarts", "one_call_per_cycle")}}
            if not args.no_cpu_baseline:
                v_emb, d_env = seen["args"][0], seen["args"][1]
                prh = {"nao": n, "nelec": tuple(int(x) for x in seen["mol"].nelec), "S": ints["S"], "hcore": ints["hcore"],
                       "V_emb": np.asarray(v_emb), "D_env": np.asarray(d_env)}
                ncyc = 200 if n <= 24 else 40
                entry["cpu_baseline"] = {
                    "value": cpu_baseline_cycle(prh, np.asarray(ints["eri"]), ncyc), "unit": "cycles/s",
                    "cores": host_cores(), "kind": "port",
                    "sample": f"{ncyc} cycles of the oracle loop (C/OpenMP dense J/K + numpy/LAPACK) on the same integrals, "
                              "V_emb and D_env"}
            out.append(entry)
        except Exception as exc:  # informative leg: never takes the bench line down
            out.append({"workload": label, "error": f"{type(exc).__name__}: {exc}"})
        be.release_workspaces()
    return out


def gather_per_rank(entry: dict, world: int, distributed: bool) -> list:
    """One small dict per rank, gathered on every rank (rank order)."""
    if not distributed or world == 1:
        return [entry]
    import torch.distributed as dist

    rows = [None] * world
    dist.all_gather_object(rows, entry)
    return rows


def packed_kernel_name(n_ao: int) -> str:
    """Which kernel nbx_jk_packed runs a size on (csrc/jk_s4.hip's dispatch)."""
    if 97 <= n_ao <= 148 and os.environ.get("NBX_JK_M8", "1") != "0":
        return "jk_m8_kernel (8-fold packed tiles: every integral once; MFMA walk, four chunks per full tile)"
    if 97 <= n_ao <= 148 and os.environ.get("NBX_JK_M4", "1") != "0":
        return "jk_m4_kernel (4-fold packed tiles, MFMA walk, four chunks per tile)"
    if 148 < n_ao <= 400 and os.environ.get("NBX_JK_MX", "1") != "0":
        return "jk_mx_kernel (4-fold packed tiles, MFMA walk, tile in whole-row / band-segment chunks)"
    return "jk_s4_kernel (4-fold packed tiles)"


def scaling_workload_leg(be, args, world, rank, distributed, barrier) -> list:
    """Sizes at which sharding the J/K build pays (it grows as N^4, the replicated rest of a cycle as N^3), for every
    --gpus N: N = 256 (8.7 GB of packed tiles per build) and N = 384 (43.7 GB), each rank holding its equal-work slab
    of (pq|rs) rows in the packed form; cycles/s of the product loop and the all-reduce of the (3,N,N) J/K partials
    timed on its own."""
    import torch
    import torch.distributed as dist

    from nbed_amd import synth
    from nbed_amd.dist import Shards
    from nbed_amd.scf import GpuUHF, Mole

    out = []
    for n_ao in (256, 384):
        try:
            nocc, nenv = n_ao // 6, n_ao // 12
            pr = synth.problem(be, n_ao, (nocc, nocc), nenv)
            sh = Shards.for_packed_jk(be, n_ao, world, rank, force_collective=distributed)
            eri = be.synth_eri(n_ao, sh.lo, sh.hi)
            mf = GpuUHF(Mole(n_ao, pr["nelec"]), pr["S"], pr["hcore"], eri, backend=be, shards=sh)
            packed = mf.eri_packed_device() is not None
            if packed:
                mf._eri_d = None  # (the packed tiles are what the loop reads: the dense slab can go)
                del eri
            steps, warmup = 10, 3
            from nbed_amd import _nbx

            be.profile(True, slots=[_nbx.PROF_JK_DENSE], every=2)  # (HIP events around every other J/K launch)
            be.profile_reset()
            run = timed_huzinaga_run(mf, (pr["V_emb"], pr["D_env"]), {}, warmup, steps, barrier)
            be.profile(False)
            jk_ms_tot, jk_n = be.profile_read(_nbx.PROF_JK_DENSE)
            jk_ms = jk_ms_tot / jk_n if jk_n else None
            dt = steps / run["cycles_per_sec"]
            ar_ms = None
            if distributed:
                buf = be.zeros((3, n_ao, n_ao))
                for _ in range(3):
                    dist.all_reduce(buf)
                barrier()
                t0 = time.perf_counter()
                for _ in range(20):
                    dist.all_reduce(buf)
                barrier()
                ar_ms = (time.perf_counter() - t0) / 20 * 1e3
            if world > 1:
                tmax = torch.tensor([dt], dtype=torch.float64, device=be.device)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
            ntiles = (sh.hi * (sh.hi + 1) - sh.lo * (sh.lo + 1)) // 2
            slab_bytes = float(be.lib.nbx_eri_packed_bytes(n_ao, sh.lo, sh.hi)) if packed else 8.0 * n_ao * n_ao * ntiles
            traffic = None
            tfile = REPO / "profiles" / "r04" / f"jk_mx_traffic_n{n_ao}.json"
            if packed and world == 1 and tfile.exists():
                try:
                    traffic = json.loads(tfile.read_text()).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            per_rank = gather_per_rank({"rank": rank, "slab_rows": [int(sh.lo), int(sh.hi)], "slab_bytes_read_per_build": slab_bytes,
                                        "jk_kernel_ms": jk_ms}, world, distributed)
            ms_cycle = dt / steps * 1e3
            out.append({"nao": n_ao, "jk_kernel": packed_kernel_name(n_ao) if packed else "jk_sym_kernel (dense tensor, tiles q <= p)",
                        "jk_kernel_ms_this_rank": jk_ms,
                        "jk_frac_of_hbm_peak_this_rank": (slab_bytes / (jk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if jk_ms else None,
                        "jk_hbm_traffic_bytes_per_launch_pmc": traffic,
                        "per_rank": per_rank,
                        "replicated_part_ms": (ms_cycle - max(r["jk_kernel_ms"] or 0.0 for r in per_rank) - (ar_ms or 0.0))
                        if jk_ms else None,
                        "replicated_part_note": "ms_per_cycle - the slowest rank's J/K kernel - the all-reduce timed alone: Fock assembly, "
                                                "Huzinaga operator, DIIS, eigensolve, density, scalars (N^3, identical on every rank)",
                        "cycles_per_sec": steps / dt, "ms_per_cycle": dt / steps * 1e3, "steps": steps, "warmup": warmup,
                        "n_gpus": world, "slab_rows": [int(sh.lo), int(sh.hi)], "slab_bytes_read_per_build": slab_bytes,
                        "allreduce_ms_alone": ar_ms, "allreduce_bytes": 24 * n_ao * n_ao,
                        "one_call_per_cycle": run["one_call_per_cycle"], "scf_restarts": run["scf_restarts"],
                        "energy_last_cycle": run["energy_last_cycle"]})
            del mf
        except Exception as exc:
            out.append({"nao": n_ao, "error": f"{type(exc).__name__}: {exc}"})
        be.release_workspaces()
        torch.cuda.empty_cache()
    return out


def real_molecule_leg(be, args) -> dict:
    """BASELINE configs[2] on REAL integrals: octane / 6-31G* (148 AOs), 4 active atoms, SPADE + concentric
    localization, B3LYP-in-HF -- integrals from libnbx's host engine, quadrature on the device, nothing
    injected, nothing synthetic.  Reports the end-to-end time of ``nbed(config)`` and, with the bench's own
    protocol (warm-up cycles untimed, stopping rule off, DIIS on), the cycles/s of the embedded Huzinaga SCF
    of that molecule: the same kernels as ``value``, on the real tensor."""
    import torch

    from nbed_amd import NbedConfig
    from nbed_amd.driver import BuiltinHFProvider
    from nbed_amd.scf import History, huzinaga_scf

    sys.path.insert(0, str(Path(__file__).resolve().parent / "tools"))
    from molecules import octane_xyz

    cfg = NbedConfig(geometry=octane_xyz(), n_active_atoms=4, basis="6-31g*", xc_functional="b3lyp", convergence=1e-8,
                     projector="huzinaga", localization="spade", virtual_localization="cl", max_shells=4,
                     max_hf_cycles=100, max_dft_cycles=100)
    prov = BuiltinHFProvider(be)
    drv, e2e, seen = embedded_molecule_run(be, cfg, prov)
    kw = seen["kw"]
    # time to solution of the embedded SCF as the driver runs it (Nbed's stopping rule, cold start), second of two runs
    for _ in range(2):
        mf_t = prov.local_hf(cfg, seen["mol"], backend=be)
        mf_t.conv_tol, mf_t.max_cycle = cfg.convergence, cfg.max_hf_cycles
        hist_t = History()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out_t = huzinaga_scf(mf_t, *seen["args"], dm_conv_tol=1e-6, use_DIIS=True, history=hist_t, **kw)
        torch.cuda.synchronize()
        tts_s = time.perf_counter() - t0
    mf = prov.local_hf(cfg, seen["mol"], backend=be)  # a fresh embedded object over the resident integrals
    run = timed_huzinaga_run(mf, seen["args"], kw, args.warmup, args.steps, torch.cuda.synchronize)
    res = drv.huzinaga
    return {
        "workload": "octane / 6-31G* (148 AOs, spherical d), 4 active atoms, SPADE + concentric localization, B3LYP-in-HF, "
                    "Huzinaga projector; AO integrals nbx_host_1e / nbx_host_eri, nothing injected",
        "nao": int(mf.get_ovlp().shape[0]),
        "active_electrons": [int(x) for x in seen["mol"].nelec],
        "end_to_end_s": e2e,
        "global_ks_e_tot": float(drv._global_ks.e_tot),
        "e_rhf": float(res["e_rhf"]),
        "embedded_scf_converged": bool(res["scf"].converged),
        "embedded_scf_time_to_solution_ms": tts_s * 1e3,
        "embedded_scf_cycles_to_solution": len(hist_t),
        "embedded_scf_converged_cold_run": bool(out_t[4]),
        "embedded_scf_restarts_cold_run": list(hist_t.info.get("restarts", [])),
        "embedded_scf_cycles_per_sec": run["cycles_per_sec"],
        "ms_per_cycle": run["ms_per_cycle"],
        "ms_per_cycle_settled": run["ms_per_cycle_settled"],
        "queue_pace_ms_first_cycles": run["queue_pace_ms_first_cycles"],
        "scf_restarts": run["scf_restarts"],
        "protocol": f"as value: {args.steps} cycles after {args.warmup} warm-up cycles of one huzinaga_scf run, DIIS on, stopping rule off; "
                    "the first cycles of a real molecule defeat the warm-started eigensolvers (levels 4e-4 Ha apart in the "
                    "virtual space, a Fock matrix that still moves), so the first cycles take their density from purification "
                    "(nbx_purify, no eigenvectors), one cold eigensolve follows when the density has settled (cycle 7 here), "
                    "then the refinement runs at the pace of the synthetic workload",
    }


def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` with no launcher around it: start N ranks (one per GPU) as a child
    ``torch.distributed.run`` BEFORE this process has touched the GPU (it never does), relay their
    output (rank 0 prints the JSON line) and return the child's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # NBED_FORCE_DIST=1 exercises the RCCL code path (init, all-gathers, barrier) with one rank
    distributed = world > 1 or os.environ.get("NBED_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # Rehearsal knob for a one-GPU box: NBED_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses
        # gloo (RCCL refuses two ranks on one device), to exercise the N > 1 code path end to end.
        rehearse = os.environ.get("NBED_BENCH_REHEARSE") == "1"
        dev_index = 0 if rehearse else local_rank
        torch.cuda.set_device(dev_index)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    else:
        dev_index = 0
        torch.cuda.set_device(0)

    from nbed_amd import _nbx
    from nbed_amd.backend import HipBackend
    from nbed_amd.dist import Shards
    from nbed_amd.scf import GpuUHF, Mole
    from nbed_amd import synth

    be = HipBackend(dev_index)
    N, n_act = args.nao, args.nact
    pr = synth.problem(be, N, (args.nocc, args.nocc), args.nenv)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    # ---------------- inputs resident in HBM before any timed region
    # slabs of equal J/K work = equal bytes of the packed form (4-fold: the triangular cut, row p has p + 1 tiles of one length; the
    # 8-fold form of N <= 148 cuts the tiles of row p at row p as well: its rows grow like p^3)
    shards = Shards.for_packed_jk(be, N, world, rank, force_collective=distributed)
    eri = be.synth_eri(N, shards.lo, shards.hi)
    mf = GpuUHF(Mole(N, pr["nelec"]), pr["S"], pr["hcore"], eri, backend=be, shards=shards)
    mf.eri_packed_device()  # the J/K kernel's packed copy of the slab: part of the resident inputs
    from nbed_amd.scf import huzinaga_scf

    # ---------------- time to solution of a REAL run (no warm-up, the reference's stopping rule):
    # Nbed's defaults convergence = 1e-6 (nbed/config.py:110) and dm_conv_tol = 1e-6 (nbed/driver.py:587),
    # cold start from the projected core Hamiltonian, DIIS on.  The first call pays one-off costs
    # (code-object loading, first-touch workspaces) that belong to the process, not to an SCF: one
    # untimed run, then a fresh SCF object is timed from the call to the results on the host.
    from nbed_amd.scf import History

    tts = None
    if not args.no_tts:
        for attempt in range(2):
            mf_t = GpuUHF(Mole(N, pr["nelec"]), pr["S"], pr["hcore"], eri, backend=be, shards=shards,
                          eri_packed=mf.eri_packed_device())  # the resident packed integrals are inputs
            mf_t.conv_tol, mf_t.max_cycle = 1e-6, 50
            hist_t = History()
            barrier()
            t_s = time.perf_counter()
            out_t = huzinaga_scf(mf_t, pr["V_emb"], pr["D_env"], dm_conv_tol=1e-6, use_DIIS=True, history=hist_t)
            barrier()
            dt_s = time.perf_counter() - t_s
        if world > 1:
            tmax = torch.tensor([dt_s], dtype=torch.float64, device=be.device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_s = float(tmax.item())
        tts = {
            "time_to_solution_ms": dt_s * 1e3,
            "cycles": len(hist_t),
            "converged": bool(out_t[4]),
            "cycles_per_s_to_solution": len(hist_t) / dt_s,
            "stopping_rule": "|dE| < 1e-6 and |dD|_F < 1e-6 (nbed/config.py:110, nbed/driver.py:587), cold start, DIIS",
            "energy": [float(x) for x in hist_t[-1][0]],
            "scf_restarts": list(hist_t.info.get("restarts", [])),
        }
        del mf_t, out_t

    # ---------------- SURVEY 8d's protocol line: "fixed at 10 cycles, DIIS on, no early exit" -- a COLD run of exactly
    # ten cycles from the projected core-Hamiltonian guess, set-up (S^-1/2, D_env S, workspaces of the SCF object,
    # cold eigensolve, DIIS filling its space) and the return of the results included; second of two runs, as above.
    fixed10 = None
    if not args.no_tts:
        for attempt in range(2):
            mf_c = GpuUHF(Mole(N, pr["nelec"]), pr["S"], pr["hcore"], eri, backend=be, shards=shards,
                          eri_packed=mf.eri_packed_device())
            mf_c.conv_tol, mf_c.max_cycle = -1.0, 10
            hist_c = History()
            barrier()
            t_s = time.perf_counter()
            huzinaga_scf(mf_c, pr["V_emb"], pr["D_env"], use_DIIS=True, history=hist_c)
            barrier()
            dt_c = time.perf_counter() - t_s
        if world > 1:
            tmax = torch.tensor([dt_c], dtype=torch.float64, device=be.device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_c = float(tmax.item())
        fixed10 = {"ms": dt_c * 1e3, "cycles": len(hist_c), "cycles_per_sec": len(hist_c) / dt_c,
                   "protocol": "SURVEY 8d: ten cycles from the core guess of a fresh SCF object, DIIS on, stopping rule off, "
                               "set-up and results included (the packed integrals are resident inputs)",
                   "dm_change_last_cycle": float(hist_c[-1][1]), "scf_restarts": list(hist_c.info.get("restarts", []))}
        del mf_c

    mf.conv_tol = -1.0  # the stopping rule can never fire: exactly max_cycle cycles run
    barrier()
    # ONE embedded SCF run of warmup + steps cycles.  The first `warmup` cycles are untimed
    # (first-touch allocations, the cold-start eigensolve of cycle 0, DIIS filling its space);
    # when the loop is about to queue cycle `warmup` the callback drains the GPU, and the clock
    # runs from there until the run has returned its results to the host.
    mf.max_cycle = args.warmup + args.steps
    from nbed_amd.scf import History

    hist = History()
    clock = {}

    stamps = []

    def on_cycle(i):
        if i == 0:
            del stamps[:]  # (a run the loop repeats -- rejected purified / tracked cycle -- is timed from its last start)
        if i > args.warmup:
            stamps.append(time.perf_counter())
        if i == args.warmup:
            barrier()
            # HIP events around the J/K kernel only, and around one launch in --jk-event-every: a pair of events holds
            # the stream for ~11 us (two marker packets; rocprofv3 kernel trace with and without them), 4 % of a cycle
            be.profile(True, slots=[_nbx.PROF_JK_DENSE], every=max(1, args.jk_event_every))
            be.profile_reset()
            clock["t0"] = time.perf_counter()
            stamps.append(clock["t0"])

    if os.environ.get("NBED_BENCH_POISON_LDS") == "1":  # (debug knob: every CU's LDS holds NaN when the run starts)
        be.debug_fill_lds(float("nan"))
    huzinaga_scf(mf, pr["V_emb"], pr["D_env"], use_DIIS=True, history=hist, callback=on_cycle)
    barrier()
    dt = time.perf_counter() - clock["t0"]
    be.profile(False)
    scf_restarts = list(hist.info.get("restarts", []))
    one_call = bool(hist.info.get("cycle_call"))
    hist = hist[args.warmup:]
    assert len(hist) == args.steps
    jk_ms, jk_cnt = be.profile_read(_nbx.PROF_JK_DENSE)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=be.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    e_last = [float(x) for x in hist[-1][0]]
    dm_change_last = float(hist[-1][1])

    # ---------------- the mu-shift projector's SCF on the same inputs (the reference's default projector)
    mu_leg, mu_h3 = None, None
    if not args.no_mu:
        try:
            mu_leg, mu_h3 = mu_shift_leg(be, (eri, mf.eri_packed_device(), shards), pr, args, barrier)
            if world > 1:
                tmax = torch.tensor([mu_leg["ms_per_cycle"]], dtype=torch.float64, device=be.device)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                mu_leg["ms_per_cycle"] = float(tmax.item())
                mu_leg["cycles_per_sec"] = 1e3 / mu_leg["ms_per_cycle"]
        except Exception as exc:  # informative leg: it must not take the bench line down with it
            mu_leg = {"error": f"{type(exc).__name__}: {exc}"}

    # ---------------- four-index transform (3 unique spin blocks), outer index sharded
    transform = None
    if not args.no_transform:
        c_full = pr["C"]
        ca = be.asarray(c_full[:, :n_act])
        cb = be.asarray(np.ascontiguousarray(c_full[:, ::-1][:, :n_act]))
        if world > 1:
            full_eri = be.synth_eri(N)  # the transform needs every p: replicate (generated, not sent)
        else:
            full_eri = eri
        ish = Shards(n_act, world, rank, force_collective=distributed)
        single = world == 1 and not distributed
        # Everything a build writes is allocated ONCE, outside the timed region: three (n^4) results
        # (2.1 GB each at n = 128; allocating them per build stalls the host for tens of ms while the
        # GPU idles) and, on one GPU, the rs-packed integrals -- (pq|rs) with the pairs (r, s <= r)
        # packed, a per-molecule input format like the J/K kernel's packed copy (GpuUHF.eri_rs_device()
        # keeps it with the SCF object; both Hamiltonian builds of a run share it).
        nfull = (n_act,) * 4
        o_aa, o_ab, o_bb = be.empty(nfull), be.empty(nfull), be.empty(nfull)
        pack_ms = None
        if single:
            barrier()
            tp = time.perf_counter()
            eri_rs = mf.eri_rs_device()
            barrier()
            pack_ms = (time.perf_counter() - tp) * 1e3
        else:
            nslab = (ish.size, n_act, n_act, n_act)
            w_aa, w_ab, w_bb = be.empty(nslab), be.empty(nslab), be.empty(nslab)

        def run_transform():
            # the three spin blocks of the unrestricted Hamiltonian (nbed/ham_builder.py:127-133);
            # (aa|aa) and (aa|bb) share quarters 1-2, as HamiltonianBuilder runs them
            if single:
                # whole outer range here: (ij|kl) = (ji|kl) for quarters 3-4, (pq|rs) = (pq|sr) for 1-2
                be.ao2mo_pair_sym(eri_rs, ca, ca, ca, cb, cb, rs_packed=True, out=o_aa, out2=o_ab)
                be.ao2mo_pair_sym(eri_rs, cb, cb, cb, rs_packed=True, out=o_bb)
                return o_aa, o_bb, o_ab
            be.ao2mo_pair(full_eri, ca, ca, ca, ca, cb, cb, i0=ish.lo, i1=ish.hi, out=w_aa, out2=w_ab)
            be.ao2mo(full_eri, cb, cb, cb, cb, i0=ish.lo, i1=ish.hi, out=w_bb)
            return [ish.all_gather(be, w, axis=0, out=o) for w, o in ((w_aa, o_aa), (w_bb, o_bb), (w_ab, o_ab))]

        for _ in range(2):  # workspaces allocated, allocator settled
            run_transform()
        barrier()
        be.profile(True)
        be.profile_reset()
        reps = 5
        t1 = time.perf_counter()
        for _ in range(reps):
            outs = run_transform()
        barrier()
        dtt = (time.perf_counter() - t1) / reps
        be.profile(False)
        q1_ms, q1_cnt = be.profile_read(_nbx.PROF_AO2MO_Q1)
        all_ms, all_cnt = be.profile_read(_nbx.PROF_AO2MO)
        device_ms = all_ms / reps  # HIP events around the quarter transforms of one build (this rank)
        if world > 1:
            tmax = torch.tensor([dtt], dtype=torch.float64, device=be.device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dtt = float(tmax.item())
        ref_flops = 3 * transform_flops(N, n_act)  # as the reference does it: three independent blocks
        executed = ref_flops - (2.0 * n_act * N**4 + 2.0 * n_act**2 * N**3)  # quarters 1-2 of (aa|bb) shared
        if single:  # quarters 3-4 of all three blocks on the pairs j <= i only,
            q34 = 2.0 * n_act**3 * N**2 + 2.0 * n_act**4 * N  # quarters 1-2 on the columns s <= r only
            executed -= 3 * q34 * (1.0 - (n_act + 1) / (2.0 * n_act))
            executed -= 2 * (2.0 * n_act * N**4 + 2.0 * n_act**2 * N**3) * (1.0 - (N + 1) / (2.0 * N))
        q1_flops = 2.0 * ish.size * N**4
        if single:
            q1_flops *= (N + 1) / (2.0 * N)  # the quarter-1 GEMM runs on the packed columns
        wall_over_device = dtt * 1e3 / device_ms if (device_ms and world == 1) else None
        stalled = wall_over_device is not None and wall_over_device > 1.3
        if stalled:
            print(f"bench.py: TRANSFORM WALL/DEVICE CHECK FAILED: wall {dtt * 1e3:.2f} ms per build vs {device_ms:.2f} ms of "
                  "device time -- the host is stalling inside the timed region", file=sys.stderr, flush=True)
        transform = {
            "metric": "active_eri_transform_gflops",
            "value": executed / dtt / 1e9,
            "unit": "GFLOP/s",
            "value_counts": "EXECUTED flops over wall time per build (flops not done are not counted)",
            "wall_ms": dtt * 1e3,
            "device_ms": device_ms,
            "wall_over_device": wall_over_device,
            "wall_device_check": "FAILED" if stalled else ("ok" if wall_over_device is not None else "n/a (multi-GPU)"),
            "ms_per_build": dtt * 1e3,
            "rs_pack_ms_once_per_molecule": pack_ms,
            "builds_timed": reps,
            "spin_blocks": 3,
            "nao": N,
            "n_act": n_act,
            "reference_count_gflops": ref_flops / dtt / 1e9,
            "flop_count": "reference_count: 2nN^4+2n^2N^3+2n^3N^2+2n^4N per block, no symmetry, three independent blocks "
                          "as in the reference; executed (= value): (aa|aa) and (aa|bb) share quarters 1-2 and on one GPU "
                          "quarters 1-2 use (pq|rs) = (pq|sr) (rs-packed input made once per molecule, timed separately) "
                          "and quarters 3-4 use (ij|kl) = (ji|kl)",
            "roofline": {
                "bound": "mfma",
                "kernel": "gemm_m4_tn_kernel (quarter-1: (n x N).(N x N^3); v_mfma_f64_4x4x4_4b_f64, k-tiles by LDS-DMA)",
                "achieved": q1_flops / (q1_ms / max(q1_cnt, 1) * 1e-3) / 1e12 if q1_cnt else None,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": (q1_flops / (q1_ms / max(q1_cnt, 1) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS) if q1_cnt else None,
                "traffic": gemm_traffic(N, n_act) if single else None,
            },
            "device_ms_per_block_all_quarters": all_ms / max(all_cnt, 1),
        }
        del outs, o_aa, o_ab, o_bb

    # ---------------- BASELINE configs[3]: N_AO = 2000, integrals generated in registers (a dense
    # tensor would be 128 TB).  A bounded sample of the full job, sharded as the full job would be:
    # each rank transforms its own `rslabs` values of r (partial tensors are summed by one
    # all-reduce at the end of a full run) and builds J/K rows for its own `jrows` values of p.
    n2000 = None
    if not args.no_n2000:
        NB, nb_act, rsl, jrows = 2000, 128, args.n2000_rslabs, 4
        cb = be.asarray(np.ascontiguousarray(synth.sym_matrix(7, NB)[:, :nb_act]))
        # the streamed transform generates only the pairs s <= r, so a slab costs ~(r+1)/N of a full
        # one: sample around r = N/2, whose cost is the average over r
        r0 = NB // 2 - (rsl * world) // 2 + rank * rsl
        cb2 = be.asarray(np.ascontiguousarray(synth.sym_matrix(6, NB)[:, :nb_act]))
        be.ao2mo_synth_pair(NB, cb, cb, cb, cb, cb2, cb2, r0=r0, r1=r0 + 1)  # first-touch of the workspaces
        barrier()
        be.profile(True, slots=[_nbx.PROF_AO2MO_Q1])
        be.profile_reset()
        t2 = time.perf_counter()
        part = be.ao2mo_synth(NB, cb, cb, cb, cb, r0=r0, r1=r0 + rsl)
        barrier()
        dts = time.perf_counter() - t2
        be.profile(False)
        # the three spin blocks of an unrestricted Hamiltonian: (aa|aa)+(aa|bb) as a pair, then (bb|bb)
        t2 = time.perf_counter()
        p_aa, p_ab = be.ao2mo_synth_pair(NB, cb, cb, cb, cb, cb2, cb2, r0=r0, r1=r0 + rsl)
        p_bb = be.ao2mo_synth(NB, cb2, cb2, cb2, cb2, r0=r0, r1=r0 + rsl)
        barrier()
        dt3 = time.perf_counter() - t2
        del p_aa, p_ab, p_bb
        q1s_ms, q1s_cnt = be.profile_read(_nbx.PROF_AO2MO_Q1)
        allreduce_ms = None
        if distributed:
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            dist.all_reduce(part)
            barrier()
            allreduce_ms = (time.perf_counter() - t3) * 1e3
        dmb = be.asarray(np.stack([synth.sym_matrix(8, NB), synth.sym_matrix(9, NB)]))
        # symmetric form: row p generates the p + 1 tiles q <= p, so rows around N/2 cost the average
        p0 = NB // 2 - (jrows * world) // 2 + rank * jrows
        be.jk_synth_sym(NB, dmb, p0, p0 + 1)
        barrier()
        t4 = time.perf_counter()
        be.jk_synth_sym(NB, dmb, p0, p0 + jrows)
        barrier()
        dtj = time.perf_counter() - t4
        generated = sum((p + 1) for p in range(p0, p0 + jrows)) * float(NB) ** 2
        if world > 1:
            tm = torch.tensor([dts, dtj, dt3], dtype=torch.float64, device=be.device)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dts, dtj, dt3 = float(tm[0].item()), float(tm[1].item()), float(tm[2].item())
        fl_slab = transform_flops(NB, nb_act) / NB  # one r-slab of one spin block, reference count (no symmetry)
        rmid = r0 + 0.5 * (rsl - 1)
        q1_flops_slab = 2.0 * nb_act * NB**2 * (rmid + 1)  # executed: s <= r only
        ex_slab = (2.0 * nb_act * NB**2 * (rmid + 1) + 2.0 * nb_act**2 * NB * (rmid + 1)
                   + 2.0 * nb_act**3 * (2 * rmid + 1) + 4.0 * nb_act**4)
        n2000 = {
            "workload": f"BASELINE configs[3] sample: N_AO={NB}, n_act={nb_act}, (pq|rs) generated in registers; "
                        f"{rsl} r-slabs and {jrows} J/K rows per rank (full job: {NB} of each)",
            "transform_executed_tflops": ex_slab * rsl * world / dts / 1e12,
            "transform_quarter1_tflops_per_gpu": (q1_flops_slab * rsl / (q1s_ms * 1e-3) / 1e12) if q1s_cnt else None,
            "transform_frac_of_fp64_mfma_peak": ex_slab * rsl / dts / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "transform_reference_count_tflops": fl_slab * rsl * world / dts / 1e12,
            "symmetry": "(pq|rs) = (pq|sr): pairs s <= r only (quarters 1-2 halved); sampled at r ~ N/2 = average cost",
            "projected_full_spin_block_s": dts * NB / (rsl * world),
            "uhf_three_blocks_reference_count_tflops": 3 * fl_slab * rsl * world / dt3 / 1e12,
            "projected_full_uhf_three_blocks_s": dt3 * NB / (rsl * world),
            "final_allreduce_ms": allreduce_ms,
            "jk_generated_gintegrals_per_s": generated * world / dtj / 1e9,
            "jk_full_tensor_equivalent_gintegrals_per_s": jrows * world * float(NB) ** 3 / dtj / 1e9,
            "projected_full_jk_build_s": dtj * NB / (jrows * world),
        }
        del part, dmb, cb, cb2
        be.release_workspaces()

    # what the roofline block below needs from the main workload, before its tensors are released
    packed = mf.eri_packed_device() is not None
    packed_bytes = float(be.lib.nbx_eri_packed_bytes(N, shards.lo, shards.hi)) if packed else None
    ca_h = be.to_host(ca) if transform is not None else None
    eri = full_eri = eri_rs = mf = None
    if transform is not None:
        ca = cb = None
    be.release_workspaces()
    torch.cuda.empty_cache()

    # ---------------- EXTRA: density-fitted J/K at N_AO = 2000 (rank 0, N=1 only)
    df_leg = None
    if rank == 0 and world == 1 and not args.no_df:
        try:
            df_leg = df_jk_leg(be, args, barrier)
        except Exception as exc:  # informative leg: it must not take the bench line down with it
            df_leg = {"error": f"{type(exc).__name__}: {exc}"}
            be.release_workspaces()
            torch.cuda.empty_cache()

    # ---------------- sizes at which sharding the J/K build pays: every --gpus N
    scaling = None
    if not args.no_scaling:
        scaling = scaling_workload_leg(be, args, world, rank, distributed, barrier)

    # ---------------- the same path on a REAL molecule (rank 0, N=1 only)
    real = None
    if rank == 0 and world == 1 and not args.no_real:
        try:
            real = real_molecule_leg(be, args)
        except Exception as exc:  # the leg is informative: it must not take the bench line down with it
            real = {"error": f"{type(exc).__name__}: {exc}"}
        be.release_workspaces()

    # ---------------- BASELINE configs[0], [1], [4]-shape on real integrals (rank 0, N=1 only)
    small = None
    if rank == 0 and world == 1 and not args.no_small:
        small = small_configs_leg(be, args)

    # ---------------- CPU baseline (rank 0, N=1 only): the oracle on the host cores
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        eri_h = be.to_host(be.synth_eri(N))  # bit-identical to oracle.synth.eri_dense(N), far faster to obtain
        cps = cpu_baseline_cycle(pr, eri_h, args.cpu_cycles)
        cpu = {
            "value": cps,
            "unit": "cycles/s",
            "cores": host_cores(),
            "kind": "port",
            "sample": f"{args.cpu_cycles} Huzinaga UHF cycles at N_AO={N}: oracle loop with C/OpenMP one-pass dense J/K "
                      "(oracle/c/jk_ref.c, -march=native) + numpy/LAPACK eigh, same inputs",
        }
        if mu_leg is not None and "error" not in mu_leg:
            mu_leg["cpu_baseline"] = {
                "value": cpu_baseline_mu_cycle(pr, eri_h, mu_h3, args.cpu_cycles), "unit": "cycles/s", "cores": host_cores(),
                "kind": "port",
                "sample": f"{args.cpu_cycles} cycles of the oracle's scf.hf.kernel loop (oracle/pyscf_like.py: CDIIS, scipy "
                          "generalised eigh, gradient) with the C/OpenMP one-pass dense J/K, same patched hcore"}
        if transform is not None:
            from oracle import hamiltonian

            tcpu0 = time.perf_counter()
            hamiltonian.ao2mo_full(eri_h, ca_h[:, :32], ca_h, ca_h, ca_h)  # 1/4 of one block's quarter-1 work
            tcpu = time.perf_counter() - tcpu0
            fl = 2.0 * 32 * N**4 + 2.0 * 32 * n_act * N**3 + 2.0 * 32 * n_act**2 * N**2 + 2.0 * 32 * n_act**3 * N
            transform["cpu_baseline"] = {
                "value": fl / tcpu / 1e9, "unit": "GFLOP/s", "cores": host_cores(), "kind": "port",
                "sample": "one (32 x n x n x n) outer-index slab of one spin block, numpy tensordot (OpenBLAS dgemm)",
            }
        del eri_h

    # ---------------- what a scaling curve needs to be read (every rank takes part: collectives): per-rank slab bytes
    # and J/K kernel time, the two collectives of the path timed on their own, the replicated remainder of a cycle
    multi = None
    if distributed:
        from nbed_amd.dist import Shards as _Shards

        per_rank = gather_per_rank({"rank": rank, "slab_rows": [int(shards.lo), int(shards.hi)],
                                    "slab_bytes_read_per_build": packed_bytes if packed else 8.0 * N * N * (
                                        (shards.hi * (shards.hi + 1) - shards.lo * (shards.lo + 1)) // 2),
                                    "jk_kernel_ms": jk_ms / max(jk_cnt, 1) if jk_cnt else None}, world, distributed)
        buf = be.zeros((3, N, N))
        for _ in range(3):
            dist.all_reduce(buf)
        barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(buf)
        barrier()
        ar_ms = (time.perf_counter() - t0) / 20 * 1e3
        ish2 = _Shards(n_act, world, rank, force_collective=True)
        slab = be.zeros((ish2.chunk, n_act, n_act, n_act))
        full = be.empty((ish2.chunk * world, n_act, n_act, n_act))
        be.all_gather_stack(slab, None, out=full)
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            be.all_gather_stack(slab, None, out=full)
        barrier()
        ag_ms = (time.perf_counter() - t0) / 3 * 1e3
        part = be.zeros((ish2.chunk * world, n_act, n_act, n_act))
        ish2.reduce_scatter_all_gather(be, part)
        barrier()
        t0 = time.perf_counter()
        ish2.reduce_scatter_all_gather(be, part)
        barrier()
        rsag_ms = (time.perf_counter() - t0) * 1e3
        del slab, full, part, buf
        ms_cycle = dt / args.steps * 1e3
        slowest = max((r["jk_kernel_ms"] or 0.0) for r in per_rank)
        multi = {"nao": N, "per_rank": per_rank, "jk_allreduce_ms_alone": ar_ms, "jk_allreduce_bytes": 24 * N * N,
                 "replicated_part_ms": ms_cycle - slowest - ar_ms,
                 "replicated_part_note": "ms_per_step - the slowest rank's J/K kernel - the all-reduce timed alone: Fock assembly, "
                                         "Huzinaga operator, DIIS, eigensolve, density, scalars (N^3, identical on every rank)",
                 "transform_allgather_ms_alone_per_block": ag_ms, "transform_allgather_bytes_per_block": 8.0 * n_act**4,
                 "streamed_transform_reduce_scatter_all_gather_ms_per_block": rsag_ms,
                 "backend": "gloo rehearsal" if os.environ.get("NBED_BENCH_REHEARSE") == "1" else "RCCL"}
        torch.cuda.empty_cache()

    if rank == 0:
        cycles_per_s = args.steps / dt
        jk_avg_ms = jk_ms / max(jk_cnt, 1)
        # The symmetric kernel reads the tiles q <= p of the slab once: 8 N^2 bytes per tile
        # ((pq|rs) = (qp|rs); the full tensor would be 8 N^4).  Even N <= 512 only; else the plain kernel.
        sym = (N % 2 == 0 and N <= 512)
        ntiles = (shards.hi * (shards.hi + 1) - shards.lo * (shards.lo + 1)) // 2 if sym else shards.size * N
        alg_bytes = 8.0 * N * N * ntiles
        # The packed kernel (the one GpuUHF uses where it applies) reads q <= p AND s <= r: the packed
        # slab, once per build.
        m8 = packed and int(be.lib.nbx_jk_packed_fold(N)) == 8  # (csrc/jk_m8.hip: N = 97 .. 148 unless NBX_JK_M8=0)
        m4 = packed and not m8 and N % 4 == 0 and 100 <= N <= 148 and os.environ.get("NBX_JK_M4", "1") != "0"  # (the sizes csrc/jk_m4.hip serves)
        alg_bytes_4fold = None
        if packed:
            # the 4-fold unique integrals of the slab's tiles: N(N+1)/2 doubles per tile -- what jk_s4's layout holds
            # exactly; jk_m4's 4 x 4 blocks store the zeros above the diagonal of the diagonal blocks as well
            # (bytes_read_by_kernel), which are not required bytes
            alg_bytes = alg_bytes_4fold = 8.0 * ntiles * (N * (N + 1) // 2)
        if m8:
            # the 8-fold unique integrals of the slab's tiles: tile T = p (p + 1) / 2 + q holds the T + 1 pairs (rs) <= (pq)
            t0_, t1_ = shards.lo * (shards.lo + 1) // 2, shards.hi * (shards.hi + 1) // 2
            alg_bytes = 8.0 * (t1_ * (t1_ + 1) - t0_ * (t0_ + 1)) / 2
        jk_kernel = ("jk_m8_kernel" if m8 else "jk_m4_kernel" if m4 else "jk_s4_kernel") if packed else ("jk_sym_kernel" if sym else "jk_dense_kernel")
        achieved = alg_bytes / (jk_avg_ms * 1e-3) / 1e9 if jk_cnt else None
        traffic = None
        tfile = REPO / "profiles" / ("r04/jk_m8_traffic_n148.json" if m8 else "r04/jk_m4_traffic_n148.json" if m4 else "jk_traffic.json")
        if m4 and not tfile.exists():
            tfile = REPO / "profiles" / "r03" / "jk_m4_traffic.json"
        if tfile.exists() and world == 1 and N == 148:
            try:
                tj = json.loads(tfile.read_text())
                if tj.get("kernel", "").startswith(jk_kernel):  # measured on the kernel this run uses
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "embedded_scf_cycles_per_sec",
            "value": cycles_per_s,
            "unit": "cycles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic octane/6-31G*-shaped embedded UHF (BASELINE configs[2]): N_AO={N}, "
                            f"n_occ=({args.nocc},{args.nocc}), n_env={args.nenv}, n_act_mo={n_act}; (pq|rs) in HBM"
                            + ((", 8-fold packed for J/K" if m8 else ", 4-fold packed for J/K") + " (packed once, outside the timed region)" if packed else ""),
                "nao": N,
                "eri_bytes": 8 * N**4,
                "parallelism": (f"equal-work p-row slabs x{world} + "
                                + ("gloo all-reduce, every rank on cuda:0 (NBED_BENCH_REHEARSE=1: a rehearsal of the code path, "
                                   "not a measurement)" if os.environ.get("NBED_BENCH_REHEARSE") == "1" else "RCCL all-reduce")
                                + "; one cycle = nbx_huz_cycle_jk | all-reduce | nbx_huz_cycle_post") if distributed
                else "single GPU; one cycle = one nbx_huz_cycle call",
                "diis": True,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": jk_kernel,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_read_by_kernel": packed_bytes,
                "algorithmic_bytes_note": (
                    "the 8-fold unique integrals ((pq|rs) = (qp|rs) = (pq|sr) = (rs|pq)): every integral read once per build, an "
                    "eighth of SURVEY 8d's 8 N^4 -- the floor of the contraction; the kernel reads bytes_read_by_kernel (tiles "
                    "cut at whole chunks + the zeros of the diagonal blocks); frac_on_4fold_bytes prices the same launch on the "
                    "4-fold bytes round 3's kernel read, dense_equivalent_gbs at 8 N^4"
                    if m8 else
                    "the 4-fold packed slab (q <= p, s <= r; (pq|rs) = (qp|rs) = (pq|sr)), read once per build: a "
                    "quarter of SURVEY 8d's 8 N^4; dense_equivalent_gbs prices the same launch at 8 N^4"
                    if packed else
                    "tiles q <= p of the dense (pq|rs) slab, read once: (pq|rs) = (qp|rs) halves "
                    "SURVEY 8d's 8 N^4; dense_equivalent_gbs prices the same launch at 8 N^4"),
                "dense_equivalent_gbs": (8.0 * shards.size * N**3) / (jk_avg_ms * 1e-3) / 1e9 if jk_cnt else None,
                "frac_on_4fold_bytes": (alg_bytes_4fold / (jk_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (jk_cnt and alg_bytes_4fold) else None,
                # the floor of the contraction itself: the 8-fold unique integrals ((pq|rs) = (rs|pq) as well),
                # which PySCF's in-core get_jk reads (jk_m8_kernel's algorithmic bytes; jk_m4_kernel does not use that last symmetry)
                "bytes_8fold_floor": float(4 * (N * (N + 1) // 2) * (N * (N + 1) // 2 + 1)) if world == 1 else None,
                "frac_vs_8fold_floor": (4.0 * (N * (N + 1) // 2) * (N * (N + 1) // 2 + 1) / (jk_avg_ms * 1e-3) / 1e9
                                        / HBM_PEAK_GBS) if (jk_cnt and world == 1) else None,
                "avg_launch_ms": jk_avg_ms,
                "launches": jk_cnt,
                "launches_note": f"HIP events around the J/K launch of one timed cycle in {max(1, args.jk_event_every)} "
                                 "(--jk-event-every; a bracket holds the stream ~11 us)",
            },
            "cpu_baseline": cpu,
            "breakdown_ms_per_cycle": {
                "jk_kernel": jk_avg_ms, "everything_else": dt / args.steps * 1e3 - jk_avg_ms,
                # the pace at which the loop queued the timed cycles (it runs one cycle ahead of the GPU, so
                # this is the GPU's pace): early cycles queue more eigensolver iterations than settled ones
                "median_cycle": float(np.median(np.diff(stamps))) * 1e3 if len(stamps) > 2 else None,
                "first_cycles": [round(float(x) * 1e3, 4) for x in np.diff(stamps)[:8]],
                # from queueing the last cycle to the results on the host (C, eps, D, Hz downloaded)
                "last_cycle_and_results": (dt - (stamps[-1] - stamps[0])) * 1e3 if world == 1 and stamps else None,
            },
            "check": {"energy_last_cycle": e_last, "dm_change_last_cycle": dm_change_last, "scf_restarts": scf_restarts,
                      "one_call_per_cycle": one_call},
            "time_to_solution": tts,
            "fixed10_cold": fixed10,
            "mu_shift": mu_leg,
            "transform": transform,
            "n2000_streamed": n2000,
            "n2000_density_fitted_jk": df_leg,
            "real_molecule": real,
            "small_configs": small,
            "scaling_workload": scaling,
            "multi_gpu": multi,
        }
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
