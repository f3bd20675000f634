"""Multi-GPU sharding of the two parts of the path that shard.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).

* J/K: the first AO index p of (pq|rs) is partitioned; rank g holds the ERI slab
  ``eri[p0:p1]`` and computes rows ``J[p0:p1,:]``, ``K[p0:p1,:]`` with no communication;
  one all-gather of 3*N*N/G doubles per SCF cycle rebuilds the full matrices.
* four-index transform: the outer MO index i of (ij|kl) is partitioned; each rank
  transforms its slab against the full ERI; one all-gather of n^4/G doubles.

Everything else in a cycle (Huzinaga products, DIIS, eigensolve) is N^3 and replicated:
it is deterministic, so every rank holds identical matrices.
"""

from __future__ import annotations

import math


class Shards:
    """Contiguous partition of ``range(n)`` over the ranks of a process group."""

    def __init__(self, n: int, world: int = 1, rank: int = 0, group=None, force_collective: bool = False):
        self.force_collective = force_collective  # run the all-gather even with one rank (testing)
        self.n = int(n)
        self.world = int(world)
        self.rank = int(rank)
        self.group = group
        self.chunk = math.ceil(self.n / self.world) if self.world > 0 else self.n
        self.lo = min(self.n, self.rank * self.chunk)
        self.hi = min(self.n, self.lo + self.chunk)

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
        lo = min(self.n, rank * self.chunk)
        return lo, min(self.n, lo + self.chunk)

    def all_gather(self, be, slab, axis: int = 0):
        """Concatenate every rank's ``slab`` (its ``lo:hi`` piece along ``axis``) into the
        full-length array.  ``slab`` may be shorter than ``chunk`` on the last ranks."""
        if self.world == 1 and not self.force_collective:
            return slab
        padded = be.pad_axis(slab, axis, self.chunk)
        gathered = be.all_gather_stack(padded, self.group)  # (world, ...)
        return be.unstack_concat(gathered, axis, self.n)
