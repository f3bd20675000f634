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

Everything else in a cycle (Huzinaga products, DIIS, eigensolve) is N^3 and replicated:
it is deterministic, so every rank holds identical matrices.
"""

from __future__ import annotations

import math


class Shards:
    """Contiguous partition of ``range(n)`` over the ranks of a process group."""

    def __init__(self, n: int, world: int = 1, rank: int = 0, group=None, force_collective: bool = False,
                 balance: str = "uniform"):
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
        elif balance == "uniform":
            self._cuts = None
            self.lo = min(self.n, self.rank * self.chunk)
            self.hi = min(self.n, self.lo + self.chunk)
        else:
            raise ValueError(f"unknown balance {balance!r}")

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

    def all_gather(self, be, slab, axis: int = 0):
        """Concatenate every rank's ``slab`` (its ``lo:hi`` piece along ``axis``) into the
        full-length array.  ``slab`` may be shorter than ``chunk`` on the last ranks."""
        if self.world == 1 and not self.force_collective:
            return slab
        if self._cuts is not None:
            raise ValueError("all_gather needs uniform shards")
        padded = be.pad_axis(slab, axis, self.chunk)
        gathered = be.all_gather_stack(padded, self.group)  # (world, ...)
        return be.unstack_concat(gathered, axis, self.n)
