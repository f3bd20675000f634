"""Multi-GPU sharding of the two parts of the path that shard.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).

* J/K: the first AO index p of (pq|rs) is partitioned; rank g holds the ERI slab
  ``eri[p0:p1]``.  The symmetric kernel reads only the tiles q <= p of it and returns full-size
  partial J and K matrices (contributions of the pairs (p, q <= p) and their mirror images);
  one all-reduce of 3*N*N doubles per SCF cycle sums them.  The work of row p grows as p + 1,
  so the slabs are cut at equal triangular work (``Shards(..., balance="triangular")``), not at
  equal height.  (Backends without the symmetric kernel compute plain row slabs and all-gather.)
* four-index transform: the outer MO index i of (ij|kl) is partitioned; each rank
  transforms its slab against the full ERI; one all-gather of n^4/G doubles.

* streamed four-index transform (N_AO = 2000: the integrals are generated in registers and never
  stored): the AO index r of (pq|rs) is partitioned instead -- rank g generates and half-transforms
  only the pairs (r in R_g, s <= r), so the hash evaluations and quarters 1-2 (94 % of the work)
  divide by G -- and the ranks' partial (ij|kl) tensors are summed by a reduce-scatter over the outer MO
  index followed by north_star's all-gather of the n^4/G shards (``streamed_transform``).  Sharding the
  outer MO index of the WORK as well would make every rank generate ALL N^4/2 integrals and leave the
  quarter-1 GEMM n/G = 16 rows tall.

Everything else in a cycle (Huzinaga products, DIIS, eigensolve) is N^3 and replicated:
it is deterministic, so every rank holds identical matrices.
"""

from __future__ import annotations

import math


def _takes_out(fn) -> bool:
    """Whether a backend's ``all_gather_stack`` accepts ``out=`` -- decided from its signature, never by trying the
    collective (a failed attempt on one rank would leave the ranks out of step)."""
    import inspect

    try:
        return "out" in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


class Shards:
    """Contiguous partition of ``range(n)`` over the ranks of a process group."""

    def __init__(self, n: int, world: int = 1, rank: int = 0, group=None, force_collective: bool = False,
                 balance: str = "uniform", row_cost=None):
        self.force_collective = force_collective  # run the collective even with one rank (testing)
        self.n = int(n)
        self.world = int(world)
        self.rank = int(rank)
        self.group = group
        self.balance = balance
        self.chunk = math.ceil(self.n / self.world) if self.world > 0 else self.n
        if balance == "triangular":
            # row p costs p + 1: boundaries at equal sum_{p<b} (p+1) = b(b+1)/2
            self._cuts = [min(self.n, int(round(self.n * math.sqrt(g / self.world)))) for g in range(self.world)]
            self._cuts.append(self.n)
            self.lo, self.hi = self._cuts[self.rank], self._cuts[self.rank + 1]
        elif balance == "cost":
            # boundaries at equal sums of a given cost per row (``row_cost``: n numbers) -- e.g. the bytes of each row's packed
            # tiles, which grow like p^3 in the 8-fold packed form (tile (p, q) holds the rows r <= p only) where the
            # 4-fold form's grow like p: the boundary nearest to g/world of the total for every rank g
            if row_cost is None or len(row_cost) != self.n:
                raise ValueError("balance='cost' needs row_cost with one entry per row")
            cum = [0.0]
            for c in row_cost:
                cum.append(cum[-1] + float(c))
            self._cuts = [0]
            for g in range(1, self.world):
                target = cum[-1] * g / self.world
                b = min(range(self._cuts[-1], self.n + 1), key=lambda i: abs(cum[i] - target))
                self._cuts.append(b)
            self._cuts.append(self.n)
            self.lo, self.hi = self._cuts[self.rank], self._cuts[self.rank + 1]
        elif balance == "uniform":
            self._cuts = None
            self.lo = min(self.n, self.rank * self.chunk)
            self.hi = min(self.n, self.lo + self.chunk)
        else:
            raise ValueError(f"unknown balance {balance!r}")

    @classmethod
    def for_packed_jk(cls, be, n: int, world: int = 1, rank: int = 0, group=None, force_collective: bool = False):
        """Row slabs of (pq|rs) at equal bytes of the packed J/K form of size ``n`` (nbx_eri_packed_bytes row by row: the
        streaming kernel's work is its bytes -- 4-fold packed: p + 1 tiles of one length per row, the triangular cut;
        8-fold packed, N = 97 .. 148: the tiles of row p are cut at row p as well); triangular where no packed form exists."""
        lib = getattr(be, "lib", None)
        if lib is not None and world > 1 and int(lib.nbx_jk_packed_supported(n)):
            # (an empty slab's size is the buffer's slack, if the form has one; a tile costs the kernels ~0.9 us beyond its
            #  stream -- the worth of 25 KB at a workgroup's 23 GB/s, profiles/r04/jk_m8_measurements.txt -- which matters
            #  where tiles are of very different lengths: row p has p + 1 of them)
            cost = [float(lib.nbx_eri_packed_bytes(n, p, p + 1)) - float(lib.nbx_eri_packed_bytes(n, p, p)) + 25600.0 * (p + 1)
                    for p in range(n)]
            if min(cost) > 0:
                return cls(n, world, rank, group, force_collective, balance="cost", row_cost=cost)
        return cls(n, world, rank, group, force_collective, balance="triangular")

    @classmethod
    def from_env(cls, n: int, group=None):
        """Partition over the default process group (single rank when not initialised)."""
        try:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized():
                return cls(n, dist.get_world_size(group), dist.get_rank(group), group)
        except Exception:
            pass
        return cls(n)

    @property
    def size(self) -> int:
        return self.hi - self.lo

    def bounds(self, rank: int) -> tuple[int, int]:
        if self._cuts is not None:
            return self._cuts[rank], self._cuts[rank + 1]
        lo = min(self.n, rank * self.chunk)
        return lo, min(self.n, lo + self.chunk)

    def all_reduce(self, be, partial):
        """Sum every rank's full-size ``partial`` (in place); identical result on all ranks."""
        if self.world == 1 and not self.force_collective:
            return partial
        return be.all_reduce_sum(partial, self.group)

    def reduce_scatter_all_gather(self, be, partial, keep_shard: bool = False):
        """Sum every rank's full-size ``partial`` over the ranks as north_star's collective: a reduce-scatter over the
        LEADING index (the outer MO index i of (ij|kl): rank g receives the sum of the rows i in its uniform shard),
        then the all-gather of those n/G-row shards -- the bytes of an all-reduce, with the summed tensor sharded until
        the last step.  ``keep_shard``: return ``(full, shard, (i0, i1))`` so a caller that goes on with its own rows
        (the spin-orbital scatter of ham_builder.py:158-216 shards the same way) need not slice them out again."""
        if self.world == 1 and not self.force_collective:
            return (partial, partial, (0, int(partial.shape[0]))) if keep_shard else partial
        n0 = int(partial.shape[0])
        rows = Shards(n0, self.world, self.rank, self.group)  # uniform shards of the leading index
        padded = be.pad_axis(partial, 0, rows.chunk * self.world)
        mine = be.reduce_scatter_sum(padded, self.group)  # (chunk, ...): this rank's rows, summed over the ranks
        gathered = be.all_gather_stack(mine, self.group)  # (world, chunk, ...)
        full = be.unstack_concat(gathered, 0, n0)
        if keep_shard:
            return full, mine[: rows.size], (rows.lo, rows.hi)
        return full

    def all_gather(self, be, slab, axis: int = 0, out=None):
        """Concatenate every rank's ``slab`` (its ``lo:hi`` piece along ``axis``) into the
        full-length array.  ``slab`` may be shorter than ``chunk`` on the last ranks.  ``out``: a
        full-length tensor to gather into -- used when the shards are even and ``axis`` is the
        leading one, where the gathered pieces ARE the result (no padding, no reshuffling copy)."""
        if self.world == 1 and not self.force_collective:
            return slab
        if self._cuts is not None:
            raise ValueError("all_gather needs uniform shards")
        if out is not None and axis == 0 and self.n == self.chunk * self.world and _takes_out(be.all_gather_stack):
            be.all_gather_stack(slab, self.group, out=out)
            return out
        padded = be.pad_axis(slab, axis, self.chunk)
        gathered = be.all_gather_stack(padded, self.group)  # (world, ...)
        return be.unstack_concat(gathered, axis, self.n)


def streamed_transform(be, nao: int, ca, cb=None, shards: Shards | None = None, seed: int = 20250829):
    """Spin blocks of the active-space two-electron tensor from GENERATED (pq|rs) (BASELINE
    configs[3]; stands in for the ao2mo.kernel calls of nbed/ham_builder.py:127-133 when a dense
    tensor cannot exist): ``(aa|aa)`` alone, or ``((aa|aa), (aa|bb), (bb|bb))`` when ``cb`` is given.

    ``shards``: a ``Shards(nao, world, rank, balance="triangular")`` over the AO index r -- a range
    costs ~sum (r + 1), so equal-work ranges are not equal-length.  Each rank transforms its range
    (``nbx_ao2mo_synth[_pair]``); the partial tensors are then summed by north_star's collective: a
    reduce-scatter over the outer MO index i followed by the all-gather of the n^4/G shards
    (``Shards.reduce_scatter_all_gather``: the bytes of an all-reduce, the summed tensor sharded until the
    last step); every rank returns the full tensors (the same bits on every rank)."""
    sh = shards if shards is not None else Shards(nao, balance="triangular")
    if sh.n != nao:
        raise ValueError(f"shards partition range({sh.n}), the AO index has {nao} values")
    if cb is None:
        parts = (be.ao2mo_synth(nao, ca, ca, ca, ca, r0=sh.lo, r1=sh.hi, seed=seed),)
    else:
        aa, ab = be.ao2mo_synth_pair(nao, ca, ca, ca, ca, cb, cb, r0=sh.lo, r1=sh.hi, seed=seed)
        bb = be.ao2mo_synth(nao, cb, cb, cb, cb, r0=sh.lo, r1=sh.hi, seed=seed)
        parts = (aa, ab, bb)
    parts = tuple(sh.reduce_scatter_all_gather(be, t) for t in parts)
    return parts[0] if cb is None else parts
