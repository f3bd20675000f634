"""The N > 1 path on CPU: world_size-2 (and 3, uneven shards) gloo runs of the product's
sharded J/K and sharded four-index transform, compared with the single-rank result."""

import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from nbed_amd.dist import Shards

HERE = Path(__file__).resolve().parent


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(tmp_path, world, n, mode="uniform"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(HERE / "_dist_worker.py"),
           str(tmp_path), str(n), mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [dict(np.load(tmp_path / f"rank{k}.npz")) for k in range(world)]


def test_shards_partition():
    for n, world in [(10, 1), (10, 2), (10, 3), (7, 8), (148, 8)]:
        bounds = [Shards(n, world, r) for r in range(world)]
        assert bounds[0].lo == 0 and bounds[-1].hi == n
        covered = sum(b.size for b in bounds)
        assert covered == n
        for a, b in zip(bounds[:-1], bounds[1:]):
            assert a.hi == b.lo
        assert all(b.bounds(b.rank) == (b.lo, b.hi) for b in bounds)


def test_triangular_shards_balance_the_symmetric_jk_work():
    for n, world in [(10, 2), (10, 3), (148, 8), (7, 8)]:
        bounds = [Shards(n, world, r, balance="triangular") for r in range(world)]
        assert bounds[0].lo == 0 and bounds[-1].hi == n
        for a, b in zip(bounds[:-1], bounds[1:]):
            assert a.hi == b.lo
        assert all(b.bounds(b.rank) == (b.lo, b.hi) for b in bounds)
    work = [sum(p + 1 for p in range(b.lo, b.hi)) for b in (Shards(148, 8, r, balance="triangular") for r in range(8))]
    assert max(work) < 1.1 * (148 * 149 / 2 / 8)  # within 10 % of the ideal share
    with pytest.raises(ValueError):
        Shards(10, 2, 0, balance="nope")


def test_cost_balanced_shards_follow_the_packed_form():
    """Shards.for_packed_jk cuts the rows at equal cost of the packed J/K form (bytes of a row's tiles + a constant per tile,
    from libnbx's own size function -- host arithmetic, no GPU): the triangular cut for a 4-fold packed size, a much later
    one for the 8-fold form of N = 148, whose rows grow like p^3 (the triangular cut would give the last of eight ranks 1.65
    times its share)."""
    from nbed_amd import _nbx

    class Host:
        lib = _nbx.load_library()

    be = Host()
    assert be.lib.nbx_jk_packed_fold(148) == 8 and be.lib.nbx_jk_packed_fold(256) == 4
    for n, world in [(148, 2), (148, 8), (256, 8), (148, 1)]:
        bounds = [Shards.for_packed_jk(be, n, world, r) for r in range(world)]
        assert bounds[0].lo == 0 and bounds[-1].hi == n
        assert all(a.hi == b.lo for a, b in zip(bounds[:-1], bounds[1:]))
        assert all(b.bounds(r) == (bounds[r].lo, bounds[r].hi) for b in bounds for r in range(world))
        byts = [be.lib.nbx_eri_packed_bytes(n, b.lo, b.hi) + 25600 * (b.hi * (b.hi + 1) - b.lo * (b.lo + 1)) // 2 for b in bounds]
        assert max(byts) < 1.08 * sum(byts) / world, (n, world, byts)
    tri = [Shards(256, 8, r, balance="triangular") for r in range(8)]  # (the same cut to a row: rounded differently)
    assert all(abs(a.hi - b.hi) <= 1 for a, b in zip(tri, (Shards.for_packed_jk(be, 256, 8, r) for r in range(8))))
    assert Shards.for_packed_jk(be, 148, 2, 0).hi > Shards(148, 2, 0, balance="triangular").hi + 8
    with pytest.raises(ValueError):
        Shards(10, 2, 0, balance="cost")
    # explicit costs: the boundary nearest to the equal share
    assert [Shards(4, 2, r, balance="cost", row_cost=[1, 1, 1, 3]).hi for r in range(2)] == [3, 4]


def test_symmetric_jk_slabs_all_reduce_match_single_rank(tmp_path):
    """World-2 gloo run of the additive-slab J/K path (equal-work slabs, one all-reduce)."""
    n = 10
    (tmp_path / "w1").mkdir()
    (tmp_path / "w2").mkdir()
    single = run_world(tmp_path / "w1", 1, n)[0]
    ranks = run_world(tmp_path / "w2", 2, n, mode="triangular")
    assert (int(ranks[0]["lo"]), int(ranks[0]["hi"])) != (0, 5)  # not the uniform cut
    for r in ranks:
        assert bool(r["conv"])
        for key in ("e", "d", "hz", "h1", "h2"):
            np.testing.assert_allclose(r[key], single[key], rtol=0, atol=1e-10, err_msg=key)
    for key in ("e", "d", "h2"):
        np.testing.assert_array_equal(ranks[0][key], ranks[1][key])


def test_lookahead_cycle_call_over_two_ranks_matches_single_rank(tmp_path):
    """World-2 gloo run of the ONE-CALL-PER-CYCLE loop (nbx_huz_cycle_jk | all-reduce | nbx_huz_cycle_post, restated
    by the look-ahead checker backend): the ranks keep the purified early cycles, the tracked eigensolver and the
    one-cycle-late convergence test, take the same number of cycles as one rank and end with the same bits."""
    n = 10
    (tmp_path / "w1").mkdir()
    (tmp_path / "w2").mkdir()
    single = run_world(tmp_path / "w1", 1, n, mode="lookahead")[0]
    ranks = run_world(tmp_path / "w2", 2, n, mode="lookahead")
    assert bool(single["cycle_call"]) and not bool(single["split"])
    for r in ranks:
        assert bool(r["cycle_call"]) and bool(r["split"]) and bool(r["conv"])
        assert int(r["ncycles"]) == int(single["ncycles"]) and int(r["restarts"]) == 0
        for key in ("e", "d", "hz", "h1", "h2"):
            np.testing.assert_allclose(r[key], single[key], rtol=0, atol=1e-10, err_msg=key)
    for key in ("e", "d", "hz", "h2"):
        np.testing.assert_array_equal(ranks[0][key], ranks[1][key])
    # the step-by-step loop on the plain checker backend reaches the same fixed point
    plain = run_world(tmp_path / "w1", 1, n, mode="triangular")[0]
    assert not bool(plain["cycle_call"])
    np.testing.assert_allclose(single["d"], plain["d"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_scf_and_transform_match_single_rank(tmp_path, world):
    n = 10  # 10 rows over 3 ranks -> 4,4,2: exercises the padded all-gather
    single = run_world(tmp_path / "w1" if (tmp_path / "w1").mkdir() is None else tmp_path, 1, n)[0]
    wdir = tmp_path / f"w{world}"
    wdir.mkdir()
    ranks = run_world(wdir, world, n)
    assert sorted((int(r["lo"]), int(r["hi"])) for r in ranks)[0][0] == 0
    for r in ranks:
        assert bool(r["conv"]) and bool(single["conv"])
        for key in ("e", "d", "hz", "h1", "h2"):
            np.testing.assert_allclose(r[key], single[key], rtol=0, atol=1e-10, err_msg=key)
        assert int(r["jk_calls"]) == int(single["jk_calls"])  # same cycle count on every rank
        # r-sharded streamed transform + reduce-scatter over the outer MO index + all-gather of the shards (one of each
        # per spin block) == the single-rank transform == the einsum oracle
        assert int(r["reduce_scatters"]) == 3
        for key in ("s_aa", "s_ab", "s_bb"):
            np.testing.assert_allclose(r[key], single[key], rtol=0, atol=1e-12, err_msg=key)
    from oracle import hamiltonian, synth

    ca, cb = single["c"][0][:, :7], single["c"][1][:, :7]
    eri = synth.eri_dense(n)
    np.testing.assert_allclose(single["s_aa"], hamiltonian.ao2mo_full(eri, ca, ca, ca, ca), rtol=0, atol=1e-12)
    np.testing.assert_allclose(single["s_ab"], hamiltonian.ao2mo_full(eri, ca, ca, cb, cb), rtol=0, atol=1e-12)
    np.testing.assert_allclose(single["s_bb"], hamiltonian.ao2mo_full(eri, cb, cb, cb, cb), rtol=0, atol=1e-12)
    r_bounds = sorted((int(r["r_lo"]), int(r["r_hi"])) for r in ranks)
    assert r_bounds[0][0] == 0 and r_bounds[-1][1] == n and all(a[1] == b[0] for a, b in zip(r_bounds, r_bounds[1:]))
    assert r_bounds[0][1] - r_bounds[0][0] > r_bounds[-1][1] - r_bounds[-1][0]  # equal work, not equal length
    np.testing.assert_array_equal(ranks[0]["s_ab"], ranks[1]["s_ab"])  # all-reduce: same bits everywhere
    # replicated N^3 work is deterministic: all ranks hold identical results
    for key in ("e", "d", "h2"):
        np.testing.assert_array_equal(ranks[0][key], ranks[1][key])
