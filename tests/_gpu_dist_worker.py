"""One rank of a multi-process run ON THE GPU (tests/test_gpu_dist.py): the product's sharded hot path on libnbx --
J/K slabs of (pq|rs) rows, one all-reduce per SCF cycle between the two halves of a cycle (nbx_huz_cycle_jk |
collective | nbx_huz_cycle_post), the outer-index-sharded four-index transform with its all-gather -- over

* ``nccl`` (= RCCL) with ONE rank: init, all-reduce, all-gather, barrier, destroy -- the collective library and
  the stream ordering around it as on a multi-GPU node (RCCL refuses two ranks on one device);
* ``gloo`` with two ranks sharing the one GPU of the test box: real slabs, real sums (through the host).

usage: _gpu_dist_worker.py OUT_DIR N BACKEND"""

import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))

from conftest import canon_sign  # noqa: E402
from oracle import synth  # noqa: E402  (problem definition only: seeded S, hcore, V_emb, D_env)

from nbed_amd.backend import HipBackend  # noqa: E402
from nbed_amd.dist import Shards  # noqa: E402
from nbed_amd.ham_builder import HamiltonianBuilder  # noqa: E402
from nbed_amd.scf import GpuUHF, History, Mole, huzinaga_scf  # noqa: E402


def main():
    out_dir, n, backend = Path(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    nocc, n_env, nmo = (n // 6 + 1, n // 6), max(1, n // 12), min(n - n // 12 - 1, 24)
    torch.cuda.set_device(0)
    dist.init_process_group(backend, device_id=torch.device("cuda", 0) if backend == "nccl" else None)
    rank, world = dist.get_rank(), dist.get_world_size()
    be = HipBackend(0)
    pr = synth.problem(n, nocc, n_env)
    sh = Shards(n, world, rank, force_collective=True, balance="triangular")
    mf = GpuUHF(Mole(n, pr["nelec"]), pr["S"], pr["hcore"], be.synth_eri(n, sh.lo, sh.hi), backend=be, shards=sh)
    mf.max_cycle, mf.conv_tol = 60, 1e-10
    hist = History()
    c, e, d, hz, conv = huzinaga_scf(mf, pr["V_emb"], pr["D_env"], dm_conv_tol=1e-8, backend=be, history=hist)

    # four-index transform: every rank holds (generates) the full tensor and transforms its shard of the outer MO
    # index; one all-gather per spin block (nbed/ham_builder.py:127-133)
    full = GpuUHF(Mole(n, pr["nelec"]), pr["S"], pr["hcore"], be.synth_eri(n), backend=be)
    c = canon_sign(c)  # (MO signs are a gauge: fixed before the Hamiltonian is compared across runs)
    full.mo_coeff, full.mo_occ = c[:, :, :nmo], full.get_occ(e, c)[:, :nmo]
    ish = Shards(nmo, world, rank, force_collective=True)
    const, h1, h2 = HamiltonianBuilder(full, 0.25, backend=be, shards=ish).build()
    np.savez(out_dir / f"rank{rank}.npz", c=c, e=e, d=d, hz=hz, conv=conv, h1=h1, h2=h2, lo=sh.lo, hi=sh.hi,
             cycle_call=bool(hist.info.get("cycle_call")), split=bool(hist.info.get("split")), ncycles=len(hist),
             restarts=len(hist.info["restarts"]), energies=np.array([h[0] for h in hist]))
    dist.barrier()
    dist.destroy_process_group()
    print(f"GPU DIST OK rank {rank}/{world} backend {backend} cycles {len(hist)}")


if __name__ == "__main__":
    main()
