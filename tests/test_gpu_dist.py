"""The N > 1 code path ON THE GPU (libnbx kernels on slabs, collectives between them), as far as one GPU allows:
RCCL with a single rank (the collective library itself: init, all-reduce on the compute stream's order, all-gather,
destroy) and two gloo ranks sharing the device (real slabs, real sums).  Both must reproduce the one-process run."""

import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import canon_sign
from oracle import synth

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(tmp_path, world, n, backend):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(HERE / "_gpu_dist_worker.py"),
           str(tmp_path), str(n), backend]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "GPU DIST OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return [dict(np.load(tmp_path / f"rank{k}.npz")) for k in range(world)]


@pytest.fixture(scope="module")
def be():
    from nbed_amd.backend import HipBackend

    return HipBackend()


def single_process(be, n):
    from nbed_amd.ham_builder import HamiltonianBuilder
    from nbed_amd.scf import GpuUHF, History, Mole, huzinaga_scf

    nocc, n_env, nmo = (n // 6 + 1, n // 6), max(1, n // 12), min(n - n // 12 - 1, 24)
    pr = synth.problem(n, nocc, n_env)
    mf = GpuUHF(Mole(n, pr["nelec"]), pr["S"], pr["hcore"], be.synth_eri(n), backend=be)
    mf.max_cycle, mf.conv_tol = 60, 1e-10
    hist = History()
    c, e, d, hz, conv = huzinaga_scf(mf, pr["V_emb"], pr["D_env"], dm_conv_tol=1e-8, backend=be, history=hist)
    assert conv and hist.info["cycle_call"] and not hist.info["split"]
    c = canon_sign(c)
    mf.mo_coeff, mf.mo_occ = c[:, :, :nmo], mf.get_occ(e, c)[:, :nmo]
    const, h1, h2 = HamiltonianBuilder(mf, 0.25, backend=be).build()
    return {"e": e, "d": d, "hz": hz, "h1": h1, "h2": h2, "ncycles": len(hist),
            "energies": np.array([h[0] for h in hist])}


def compare(ranks, ref):
    for r in ranks:
        assert bool(r["conv"]) and bool(r["cycle_call"]) and bool(r["split"]) and int(r["restarts"]) == 0
        # the slabs' partial sums are added in another order than the one-process kernel adds them: same
        # trajectory to rounding, so the same number of cycles unless the stopping test is met by a hair
        assert abs(int(r["ncycles"]) - ref["ncycles"]) <= 1
        k = min(int(r["ncycles"]), ref["ncycles"])
        np.testing.assert_allclose(r["energies"][:k], ref["energies"][:k], rtol=0, atol=1e-9)
        for key in ("e", "d", "hz"):
            np.testing.assert_allclose(r[key], ref[key], rtol=0, atol=1e-8, err_msg=key)
        for key in ("h1", "h2"):  # (through the MO coefficients: 1e-7, as the golden tests hold them)
            np.testing.assert_allclose(r[key], ref[key], rtol=0, atol=1e-6, err_msg=key)


@pytest.mark.parametrize("n", [24, 104])
def test_rccl_single_rank_split_cycle_and_all_gather(be, tmp_path, n):
    """RCCL executes: one rank, backend "nccl".  The SCF cycle is nbx_huz_cycle_jk | RCCL all-reduce |
    nbx_huz_cycle_post (n = 24: symmetric kernel on the dense tensor; n = 104: packed kernel), the transform ends
    in an RCCL all-gather; results against the one-call-per-cycle run of this process."""
    ranks = run_world(tmp_path, 1, n, "nccl")
    compare(ranks, single_process(be, n))


@pytest.mark.parametrize("n", [24, 104])
def test_two_ranks_on_one_gpu_split_cycle(be, tmp_path, n):
    """Two processes, each with its own slab of (pq|rs) rows on the same device, summed by gloo: every rank keeps
    the look-ahead loop (purified early cycles, tracked eigensolver) and all ranks end with the same bits."""
    ranks = run_world(tmp_path, 2, n, "gloo")
    assert (int(ranks[0]["lo"]), int(ranks[0]["hi"])) != (0, n) and int(ranks[0]["hi"]) == int(ranks[1]["lo"])
    compare(ranks, single_process(be, n))
    for key in ("e", "d", "hz", "h2"):
        np.testing.assert_array_equal(ranks[0][key], ranks[1][key])


def test_bench_child_process_over_single_rank_rccl(tmp_path):
    """``bench.py`` itself with NBED_FORCE_DIST=1 in a child process: one rank, backend "nccl" -- process-group
    init, the per-cycle all-reduce of the J/K partials inside the timed SCF (nbx_huz_cycle_jk | RCCL | _post), the
    all-gather of the transform slabs, the all-reduces of the scaling workloads, barrier, destroy -- and ONE JSON line
    that says so."""
    import gc
    import json

    import torch

    # the child needs the card's memory for its N_AO = 384 dense tensor (162 GiB): hand back what this process's
    # caching allocator still holds from the earlier tests
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, NBED_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(HERE.parent / "bench.py"), "--no-real", "--no-n2000", "--no-small", "--no-cpu-baseline",
           "--no-tts", "--steps", "3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["metric"] == "embedded_scf_cycles_per_sec" and out["n_gpus"] == 1 and out["steps"] == 3
    assert "RCCL all-reduce" in out["config"]["parallelism"] and out["check"]["one_call_per_cycle"]
    assert out["value"] > 0 and out["transform"]["value"] > 0
    for leg in out["scaling_workload"]:
        assert "error" not in leg, leg
        assert leg["allreduce_ms_alone"] is not None and leg["allreduce_ms_alone"] > 0
