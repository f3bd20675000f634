"""Worker for tests/test_dist_gloo.py: one rank of a world_size-N gloo run on CPU.

Each rank holds only its slab of (pq|rs) rows, runs the product's sharded code path
(GpuUHF.jk_device -> Shards.all_gather; HamiltonianBuilder with an outer-index shard) with the
checker backend, and writes its results for the parent test to compare with a 1-rank run."""

import os
import sys
from pathlib import Path

import numpy as np
import torch.distributed as dist

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))

from oracle import synth  # noqa: E402
from oracle_backend import OracleBackend, OracleLookaheadBackend  # noqa: E402

from nbed_amd.dist import Shards, streamed_transform  # noqa: E402
from nbed_amd.ham_builder import HamiltonianBuilder  # noqa: E402
from nbed_amd.scf import GpuUHF, History, Mole, huzinaga_scf  # noqa: E402


def main():
    out_dir = Path(sys.argv[1])
    n, nocc, n_env, nmo = int(sys.argv[2]), (5, 4), 1, 7
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mode = sys.argv[3] if len(sys.argv) > 3 else "uniform"
    # "lookahead": the checker WITH the one-call-per-cycle interface -- the product's look-ahead loop runs, a cycle
    # being nbx_huz_cycle_jk | all-reduce | nbx_huz_cycle_post as on the GPUs
    be = OracleLookaheadBackend() if mode == "lookahead" else OracleBackend()
    pr = synth.problem(n, nocc, n_env)
    if mode in ("triangular", "lookahead"):  # the symmetric J/K path: equal-work slabs, full-size partials, all-reduce
        be.use_sym = True
        sh = Shards(n, world, rank, balance="triangular")
    else:
        sh = Shards.from_env(n)
    assert (sh.world, sh.rank) == (world, rank)
    eri_slab = synth.eri_block(n, sh.lo, sh.hi)  # this rank's rows only
    mf = GpuUHF(Mole(n, pr["nelec"]), pr["S"], pr["hcore"], eri_slab, backend=be, shards=sh)
    mf.max_cycle, mf.conv_tol = 40, 1e-10
    hist = History()
    c, e, d, hz, conv = huzinaga_scf(mf, pr["V_emb"], pr["D_env"], dm_conv_tol=1e-8, backend=be, history=hist)

    # four-index transform: every rank has the full ERI (generated, not sent), shards index i
    full = GpuUHF(Mole(n, pr["nelec"]), pr["S"], pr["hcore"], synth.eri_dense(n), backend=be)
    full.mo_coeff, full.mo_occ = c[:, :, :nmo], mf.get_occ(e, c)[:, :nmo]
    ish = Shards.from_env(nmo)
    const, h1, h2 = HamiltonianBuilder(full, 0.25, backend=be, shards=ish).build()
    # streamed (generated-integral) transform, the N_AO = 2000 path: r-sharded at equal work; the partial tensors
    # summed by a reduce-scatter over the outer MO index + the all-gather of the shards (north_star's collective)
    rsh = Shards(n, world, rank, balance="triangular")
    ca_d, cb_d = be.asarray(c[0][:, :nmo]), be.asarray(c[1][:, :nmo])
    s_aa, s_ab, s_bb = (be.to_host(t) for t in streamed_transform(be, n, ca_d, cb_d, shards=rsh))
    np.savez(out_dir / f"rank{rank}.npz", c=c, e=e, d=d, hz=hz, conv=conv, h1=h1, h2=h2, lo=sh.lo, hi=sh.hi,
             jk_calls=be.calls.get("jk", 0), cycle_call=bool(hist.info.get("cycle_call")),
             split=bool(hist.info.get("split")), ncycles=len(hist), restarts=len(hist.info["restarts"]), s_aa=s_aa, s_ab=s_ab, s_bb=s_bb, r_lo=rsh.lo, r_hi=rsh.hi,
             reduce_scatters=be.calls.get("reduce_scatter", 0))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
