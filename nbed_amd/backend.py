"""Device backend: every numeric operation of the hot path as a libnbx call.

``HipBackend`` owns one ``nbx_ctx`` (one GPU, launched on torch's current HIP
stream).  torch is used only as plumbing: device memory (``torch.empty``),
host<->device copies and, in multi-GPU runs, ``torch.distributed`` collectives;
no arithmetic of the path is done with torch or numpy here.

The methods form the interface the host-side mirror of the reference
(``nbed_amd.scf``, ``nbed_amd.localizers``, ``nbed_amd.ham_builder``) is written
against.  There is deliberately no CPU implementation in this package: without
libnbx.so or without a GPU ``get_backend()`` raises ``NbxUnavailableError``.
"""

from __future__ import annotations

import ctypes
import os
import time
from ctypes import c_double, c_int

import numpy as np

from . import _nbx
from ._nbx import NbxError, NbxUnavailableError  # noqa: F401  (re-exported)

_default_backend = None


def get_backend(device: int | None = None):
    """Process-wide default ``HipBackend`` (created on first use)."""
    global _default_backend
    if _default_backend is None:
        _default_backend = HipBackend(device)
    return _default_backend


def set_backend(backend) -> None:
    """Install ``backend`` as the default (tests inject their own checker backend)."""
    global _default_backend
    _default_backend = backend


class _PendingScalars:
    """A few doubles on their way to pinned host memory: either a stream-ordered copy of ``d_vals``
    (an event marks the point after which they can be read) or, with ``host`` given, values a queued
    kernel stores there itself (pinned memory is mapped into the device's address space) followed by
    a word that turns 1.0 once they are all visible: ``get()`` polls that word.  No event goes on the
    stream for those -- an event record between two SCF cycles holds the next cycle's first kernel
    back by ~10 us."""

    _POLL_TIMEOUT_S = 120.0

    def __init__(self, torch, d_vals, ntail=0, ncore=4, host=None, ndtail=0):
        self._ncore = ncore
        self._ndtail = ndtail  # doubles behind the status words (the mu-shift cycle's gradient sums)
        self._torch = torch
        self._event = None
        self._queried = False
        if host is None:
            self._host = torch.empty(d_vals.shape, dtype=d_vals.dtype, pin_memory=True)
            self._host.copy_(d_vals, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
            self._flag = None
        else:
            self._host = host  # (ncore + ntail + ndtail + 1,): the caller cleared the last word before queueing the kernel
            self._flag = host.numpy()[ncore + ntail + ndtail:ncore + ntail + ndtail + 1]
        self._keep = d_vals  # the source must outlive the copy
        self._ntail = ntail

    def _wait(self):
        if self._event is not None:
            self._event.synchronize()
            return
        flag = self._flag
        if not self._queried:
            # hipStreamQuery does not wait; it lets the runtime retire the commands that have finished.  Left
            # to the end of a run they cost its last synchronisation 0.1-0.2 ms (measured at N = 148: 2590-2640
            # cycles/s without the query, 2630-2670 with one every fourth cycle, 2640-2720 with one per cycle)
            self._queried = True
            self._torch.cuda.current_stream().query()
        if flag[0] == 1.0:
            return
        deadline = time.monotonic() + self._POLL_TIMEOUT_S
        spins = 0
        while flag[0] != 1.0:
            spins += 1
            if (spins & 0xfff) == 0 and time.monotonic() > deadline:
                self._torch.cuda.synchronize()  # surfaces a device fault, if that is why nothing arrived
                if flag[0] != 1.0:
                    raise RuntimeError("the SCF cycle's scalars never reached the host")

    def get(self) -> np.ndarray:
        self._wait()
        self._keep = None
        return self._host.numpy()[: self._ncore].copy()

    def get_extra(self):
        """The appended status words (as ints), or None."""
        self._wait()
        if not self._ntail:
            return None
        return self._host.numpy()[self._ncore:self._ncore + self._ntail].astype(np.int64)

    def get_dtail(self):
        """The doubles stored behind the status words, or None."""
        self._wait()
        if not self._ndtail:
            return None
        o = self._ncore + self._ntail
        return self._host.numpy()[o:o + self._ndtail].copy()


class _Deferred:
    """Result of a call that is still running: ``get()`` waits for it and returns the value."""

    def __init__(self, fn):
        self._fn = fn
        self._done = False
        self._value = None

    def get(self):
        if not self._done:  # (collected once: a later transfer may reuse the buffer the value came through)
            self._value = self._fn()
            self._done = True
        return self._value


class HipBackend:
    """libnbx on one MI355X."""

    name = "hip"
    _PIN_SLOTS, _PIN_WIDTH = 256, 4 + 64 + 4  # 4 scalars + up to 64 status words + the ready word (nbx_huz_cycle_scalars_dev)

    def __init__(self, device: int | None = None):
        import torch

        self.torch = torch
        self.lib = _nbx.load_library()
        if not torch.cuda.is_available():
            raise NbxUnavailableError("no HIP device visible: the nbed_amd compute path needs an MI355X")
        if device is None:
            device = torch.cuda.current_device()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.device_index)
        stream = torch.cuda.current_stream(self.device_index).cuda_stream
        ctx = ctypes.c_void_p()
        _nbx.check(self.lib, self.lib.nbx_ctx_create(self.device_index, ctypes.c_void_p(stream), 0, ctypes.byref(ctx)))
        self.ctx = ctx
        self._work: dict[str, object] = {}
        self._poison = os.environ.get("NBED_POISON_EMPTY") == "1"
        # result slots for kernels that store straight into host memory (huz_cycle_scalars_async): a ring
        # owned by the backend for its whole life, so that a handle dropped before its kernel ran
        # (look-ahead cycles discarded after convergence) can never leave a queued kernel writing
        # into memory the pinned allocator has handed to someone else.  A slot is reused after
        # _PIN_SLOTS further calls; the SCF loops read a handle one call late.
        self._pin_ring = torch.empty((self._PIN_SLOTS, self._PIN_WIDTH), dtype=torch.float64, pin_memory=True)
        self._pin_next = 0
        # to_host_many's landing buffer: 4 MB of pinned memory, enough for the results of an SCF up to N_AO ~ 290 (C, D, Hz
        # of two spins + eps).  It was 1 MB -- 648 bytes short of an N = 148 run's results, so the first SCF of a process
        # ended on a hipHostMalloc of a bigger one: 0.35 ms behind its last cycle, 6 % of a 20-cycle benchmark run.
        self._pin_results = torch.empty(1 << 19, dtype=torch.float64, pin_memory=True)
        self._gather_args = ((ctypes.c_void_p * 8)(), (ctypes.c_int64 * 8)())
        self.to_host_many([torch.zeros(1, dtype=torch.float64, device=self.device)])  # (first launch of its kernel: 0.1 ms)

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.nbx_ctx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def _call(self, name, *args):
        _nbx.check(self.lib, getattr(self.lib, name)(self.ctx, *args))

    def use_current_stream(self):
        stream = self.torch.cuda.current_stream(self.device_index).cuda_stream
        self._call("nbx_ctx_set_stream", ctypes.c_void_p(stream))

    def synchronize(self):
        self._call("nbx_sync")

    def side(self):
        """A second context of this device on a stream of its own (``.stream``), for work that runs BESIDE what
        this backend queues -- e.g. the two-workgroup Jacobi eigensolve of a purified SCF cycle while the next
        cycles use the other 254 CUs.  Order the two with events: ``fork_to(side)`` / ``join_from(side)``."""
        if getattr(self, "_side", None) is None:
            torch = self.torch
            stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(stream):
                other = type(self)(self.device_index)
            other.stream = stream
            self._side = other
        return self._side

    def fork_to(self, other):
        """Everything queued here so far happens before what ``other`` (a ``side()`` backend) queues next."""
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device_index))
        other.stream.wait_event(ev)

    def join_from(self, other):
        """Everything ``other`` has queued so far happens before what this backend queues next."""
        ev = self.torch.cuda.Event()
        ev.record(other.stream)
        self.torch.cuda.current_stream(self.device_index).wait_event(ev)

    def empty(self, *shape):
        t = self.torch.empty(*shape, dtype=self.torch.float64, device=self.device)
        if self._poison:  # (NBED_POISON_EMPTY=1, a test knob: what a kernel does not write reads as NaN)
            t.fill_(float("nan"))
        return t

    def zeros(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.float64, device=self.device)

    def asarray(self, a):
        """Host (numpy) or device array -> contiguous float64 device array."""
        t = self.torch
        if isinstance(a, t.Tensor):
            return a.to(device=self.device, dtype=t.float64).contiguous()
        h = np.ascontiguousarray(a, dtype=np.float64)
        if h.size >= (1 << 26):
            return self._to_device_pipelined(h)
        return t.from_numpy(h).to(self.device)

    def _to_device_pipelined(self, h: np.ndarray, chunk: int = 32 * 1024 * 1024, nbuf: int = 3):
        """Large host -> device copy (a dense (pq|rs) from the integral provider is GBs): host threads
        fill a ring of pinned buffers, each shipped by an asynchronous copy on a copy stream."""
        from concurrent.futures import ThreadPoolExecutor

        torch = self.torch
        flat = h.reshape(-1)
        n = flat.size
        out = torch.empty(n, dtype=torch.float64, device=self.device)
        try:
            threads = max(1, min(8, len(os.sched_getaffinity(0))))
        except AttributeError:
            threads = 4
        bufs = [torch.empty(chunk, dtype=torch.float64, pin_memory=True) for _ in range(nbuf)]
        events = [None] * nbuf
        stream = torch.cuda.Stream(device=self.device)
        # ``out`` may be a block the caching allocator handed back while kernels queued on the compute
        # stream still use it: the copy stream starts behind them, and the block is marked as in use
        # on it
        stream.wait_stream(torch.cuda.current_stream(self.device_index))
        out.record_stream(stream)
        nchunks = (n + chunk - 1) // chunk
        with ThreadPoolExecutor(threads) as pool:
            for k in range(nchunks):
                b = k % nbuf
                lo, hi = k * chunk, min(n, (k + 1) * chunk)
                if events[b] is not None:
                    events[b].synchronize()  # the copy that last read this buffer is done
                piece = bufs[b].numpy()[: hi - lo]
                step = (hi - lo + threads - 1) // threads
                for f in [pool.submit(np.copyto, piece[i * step: min(hi - lo, (i + 1) * step)],
                                      flat[lo + i * step: min(hi, lo + (i + 1) * step)]) for i in range(threads)]:
                    f.result()
                with torch.cuda.stream(stream):
                    out[lo:hi].copy_(bufs[b][: hi - lo], non_blocking=True)
                    events[b] = torch.cuda.Event()
                    events[b].record()
        torch.cuda.current_stream(self.device_index).wait_stream(stream)
        for ev in events:
            if ev is not None:
                ev.synchronize()  # the pinned buffers are about to be released
        return out.reshape(tuple(h.shape))

    def asarray_many(self, arrays):
        """Several small host arrays -> device arrays with ONE copy (a pageable upload costs ~50 us whatever its
        size below a megabyte: an SCF that starts with four of them waits 0.2 ms for nothing)."""
        hs = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays]
        starts, total = [], 0
        for h in hs:  # every array starts on a 256-byte boundary, like an allocation of its own
            starts.append(total)
            total += (h.size + 31) & ~31
        stage = np.zeros(total)
        for h, o in zip(hs, starts):
            stage[o:o + h.size] = h.reshape(-1)
        flat = self.torch.from_numpy(stage).to(self.device)
        return [flat[o:o + h.size].view(h.shape) for h, o in zip(hs, starts)]

    def to_host_many(self, tensors, wait: bool = True):
        """Several small device arrays -> host arrays with ONE launch.  ``wait=False``: returns a handle at once --
        the kernel stores straight into pinned memory and a word behind the data tells when it is all there -- whose
        ``get()`` polls that word and hands out the arrays: host work in between overlaps the transfer."""
        torch = self.torch
        total = sum(int(t.numel()) for t in tensors)
        # a deferred transfer still in flight owns the landing buffer: finish it first (its arrays are then held by its
        # handle), so that a second call -- from a callback, a logging hook -- can neither clear its ready word nor
        # overwrite data the first gather is still storing
        pending = getattr(self, "_pin_pending", None)
        if pending is not None:
            self._pin_pending = None
            pending.get()
        pin = getattr(self, "_pin_results", None)  # (a pageable destination costs a staging copy per call)
        if pin is None or pin.numel() < total + 1:
            pin = self._pin_results = torch.empty(max(total + 1, 1 << 19), dtype=torch.float64, pin_memory=True)
        flat = pin.numpy()
        direct = len(tensors) <= 8 and all(t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() for t in tensors)
        if direct:
            # one kernel that stores straight into the pinned buffer (nbx_gather_to_host)
            flat[total] = 0.0
            # (argument arrays of the full eight entries, made once: ctypes builds a new array TYPE for every new length,
            #  0.2-0.4 ms the first time -- which used to be behind the last cycle of the first SCF run of a process)
            srcs, sizes = self._gather_args
            for i, t in enumerate(tensors):
                srcs[i] = t.data_ptr()
                sizes[i] = int(t.numel())
            self._call("nbx_gather_to_host", len(tensors), srcs, sizes, self._p(pin), 1 if wait else 0)
        else:
            pin[:total].copy_(torch.cat([t.reshape(-1) for t in tensors]), non_blocking=True)
            torch.cuda.current_stream(self.device_index).synchronize()
        shapes = [tuple(t.shape) for t in tensors]

        def collect():
            if direct and not wait and flat[total] != 1.0:
                deadline = time.monotonic() + 120.0
                spins = 0
                while flat[total] != 1.0:
                    spins += 1
                    if (spins & 0xfff) == 0 and time.monotonic() > deadline:
                        torch.cuda.synchronize()
                        if flat[total] != 1.0:
                            raise RuntimeError("the gathered results never reached the host")
            out, off = [], 0
            for shp in shapes:
                n = int(np.prod(shp)) if len(shp) else 1
                out.append(flat[off:off + n].reshape(shp).copy())
                off += n
            return out

        if wait:
            return collect()
        handle = _Deferred(collect)
        self._pin_pending = handle
        return handle

    def to_host(self, a) -> np.ndarray:
        if isinstance(a, self.torch.Tensor):
            if a.is_cuda and a.dtype == self.torch.float64 and a.numel() >= (1 << 26):
                return self._to_host_pipelined(a)
            return a.detach().cpu().numpy()
        return np.asarray(a)

    def _to_host_pipelined(self, a, chunk: int = 32 * 1024 * 1024, nbuf: int = 3) -> np.ndarray:
        """Large device -> host copy (the (2n)^4 spin-orbital tensor is tens of GB): asynchronous
        chunks into a ring of pinned buffers on a copy stream, drained into the pageable result by a
        few host threads (numpy releases the GIL while copying).  ~54 GB/s against ~12.5 GB/s for
        ``tensor.cpu()`` on an MI355X host."""
        from concurrent.futures import ThreadPoolExecutor

        torch = self.torch
        src = a.detach().contiguous().reshape(-1)
        n = src.numel()
        out = np.empty(n, dtype=np.float64)
        try:
            threads = max(1, min(8, len(os.sched_getaffinity(0))))
        except AttributeError:
            threads = 4
        bufs = [torch.empty(chunk, dtype=torch.float64, pin_memory=True) for _ in range(nbuf)]
        events = [torch.cuda.Event() for _ in range(nbuf)]
        stream = torch.cuda.Stream(device=self.device)
        stream.wait_stream(torch.cuda.current_stream(self.device_index))
        nchunks = (n + chunk - 1) // chunk
        pending = [None] * nbuf

        def drain(pool, k):
            b = k % nbuf
            lo, hi = k * chunk, min(n, (k + 1) * chunk)
            events[b].synchronize()
            piece = bufs[b].numpy()[: hi - lo]
            step = (hi - lo + threads - 1) // threads
            return [pool.submit(np.copyto, out[lo + i * step: min(hi, lo + (i + 1) * step)],
                                piece[i * step: min(hi - lo, (i + 1) * step)]) for i in range(threads)]

        with ThreadPoolExecutor(threads) as pool:
            for k in range(nchunks + nbuf - 1):
                if k < nchunks:
                    b = k % nbuf
                    if pending[b] is not None:
                        for f in pending[b]:
                            f.result()
                    lo, hi = k * chunk, min(n, (k + 1) * chunk)
                    with torch.cuda.stream(stream):
                        bufs[b][: hi - lo].copy_(src[lo:hi], non_blocking=True)
                        events[b].record()
                j = k - (nbuf - 1)
                if j >= 0:
                    pending[j % nbuf] = drain(pool, j)
            for p in pending:
                if p is not None:
                    for f in p:
                        f.result()
        return out.reshape(tuple(a.shape))

    def copy(self, a):
        return a.clone()

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)

    def _workspace(self, key: str, nbytes: int):
        buf = self._work.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = None
            self._work[key] = None
            buf = self.torch.empty(max(int(nbytes), 256), dtype=self.torch.uint8, device=self.device)
            self._work[key] = buf
        if self._poison:  # every use of a workspace starts from NaN bit patterns (0xff bytes)
            buf.fill_(255)
        return buf

    def release_workspaces(self):
        self._work.clear()

    # ------------------------------------------------------------------ in-library HIP-event timing
    def profile(self, on: bool = True, slots=None, every: int = 1):
        """Bracket launches with HIP events; ``slots``: only these NBX_PROF_* slots; ``every``: one launch in
        this many of each slot (an event pair holds the stream ~11 us)."""
        self._call("nbx_profile_sample", int(every))
        if on and slots is not None:
            mask = 0
            for sl in slots:
                mask |= 1 << int(sl)
            self._call("nbx_profile_enable", 2 | (mask << 2))
        else:
            self._call("nbx_profile_enable", 1 if on else 0)

    def debug_fill_lds(self, value: float = float("nan")):
        """Test support: every CU's LDS filled with ``value`` (nbx_debug_fill_lds)."""
        self._call("nbx_debug_fill_lds", float(value))

    def profile_reset(self):
        self._call("nbx_profile_reset")

    def profile_read(self, slot: int) -> tuple[float, int]:
        """(milliseconds summed over launches, launch count) of a NBX_PROF_* slot."""
        ms = c_double()
        cnt = ctypes.c_int64()
        self._call("nbx_profile_read", slot, ctypes.byref(ms), ctypes.byref(cnt))
        return ms.value, cnt.value

    # ------------------------------------------------------------------ collectives (plumbing)
    def pad_axis(self, a, axis: int, length: int):
        """Zero-pad ``a`` along ``axis`` up to ``length`` (equal-sized all-gather pieces)."""
        if a.shape[axis] == length:
            return a.contiguous()
        shape = list(a.shape)
        shape[axis] = length
        out = self.zeros(shape)
        out.narrow(axis, 0, a.shape[axis]).copy_(a)
        return out

    def all_gather_stack(self, a, group=None, out=None):
        """RCCL all-gather of equal-shaped pieces -> (world, *a.shape).  ``out``: a contiguous tensor
        of world * a.numel() elements to receive the pieces (no allocation in the caller's loop)."""
        import torch.distributed as dist

        world = dist.get_world_size(group)
        a = a.contiguous()
        # output in the concatenated form (world * n0, ...): the one every backend accepts
        if out is None:
            out = self.empty((world * a.shape[0],) + tuple(a.shape[1:]))
        else:
            out = out.view((world * a.shape[0],) + tuple(a.shape[1:]))
        dist.all_gather_into_tensor(out, a, group=group)
        return out.view((world,) + tuple(a.shape))

    def all_reduce_sum(self, a, group=None):
        """RCCL all-reduce (sum) in place; every rank ends with the same bits."""
        import torch.distributed as dist

        dist.all_reduce(a, op=dist.ReduceOp.SUM, group=group)
        return a

    def reduce_scatter_sum(self, a, group=None):
        """RCCL reduce-scatter (sum) of ``a`` (world * chunk, ...) over its leading axis: this rank's (chunk, ...) piece."""
        import torch.distributed as dist

        world = dist.get_world_size(group)
        a = a.contiguous()
        chunk = a.shape[0] // world
        if dist.get_backend(group) == "gloo":  # (the one-GPU rehearsal of bench.py: gloo has no reduce-scatter --
            rank = dist.get_rank(group)        #  one reduce per destination rank, which is what a reduce-scatter is)
            mine = None
            for dst in range(world):
                piece = a[dst * chunk:(dst + 1) * chunk].clone()
                dist.reduce(piece, dst=dst, op=dist.ReduceOp.SUM, group=group)
                if dst == rank:
                    mine = piece
            return mine
        out = self.empty((chunk,) + tuple(a.shape[1:]))
        dist.reduce_scatter_tensor(out, a, op=dist.ReduceOp.SUM, group=group)
        return out

    def unstack_concat(self, stacked, axis: int, n: int):
        """(world, ..., chunk, ...) -> (..., world*chunk, ...)[:n] along ``axis``."""
        world = stacked.shape[0]
        moved = stacked.movedim(0, axis)  # (..., world, chunk, ...)
        shape = list(moved.shape)
        merged = moved.reshape(shape[:axis] + [world * shape[axis + 1]] + shape[axis + 2:])
        return merged.narrow(axis, 0, n).contiguous()

    # ------------------------------------------------------------------ synthetic ERI
    def synth_eri(self, nao: int, p0: int = 0, p1: int | None = None, seed: int = 20250829):
        p1 = nao if p1 is None else p1
        out = self.empty((p1 - p0, nao, nao, nao))
        self._call("nbx_synth_eri", nao, p0, p1, seed, self._p(out))
        return out

    # ------------------------------------------------------------------ J/K
    def jk(self, eri, dm, p0: int = 0, p1: int | None = None):
        """(1+ndm, p1-p0, N): J rows from sum(dm), then K rows per dm (nbx_jk_dense)."""
        nao = dm.shape[-1]
        p1 = nao if p1 is None else p1
        dm3 = dm.reshape(-1, nao, nao)
        ndm = dm3.shape[0]
        nbytes = self.lib.nbx_jk_dense_worksize(nao, p1 - p0, ndm)
        work = self._workspace("jk", nbytes)
        out = self.empty((1 + ndm, p1 - p0, nao))
        self._call("nbx_jk_dense", nao, p0, p1, self._p(eri), self._p(dm3), ndm, self._p(out), self._p(work),
                   work.numel())
        return out

    def jk_sym(self, eri, dm, p0: int = 0, p1: int | None = None):
        """(1+ndm, N, N) full-size J/K contributions of slab rows [p0,p1) (nbx_jk_dense_sym: only the
        tiles q <= p are read); with the whole tensor they are J and K, across slabs they add up."""
        nao = dm.shape[-1]
        p1 = nao if p1 is None else p1
        dm3 = dm.reshape(-1, nao, nao)
        ndm = dm3.shape[0]
        nbytes = self.lib.nbx_jk_dense_sym_worksize(nao, p0, p1, ndm)
        work = self._workspace("jk", nbytes)
        out = self.empty((1 + ndm, nao, nao))
        self._call("nbx_jk_dense_sym", nao, p0, p1, self._p(eri), self._p(dm3), ndm, self._p(out), self._p(work),
                   work.numel())
        return out

    def jk_packed_supported(self, nao: int) -> bool:
        return bool(self.lib.nbx_jk_packed_supported(nao))

    def eri_pack(self, eri, nao: int, p0: int = 0, p1: int | None = None):
        """Slab rows [p0,p1) of the dense tensor in the 4-fold packed tile format of nbx_jk_packed
        (q <= p, s <= r: a quarter of the bytes).  Done once per SCF: the integrals are constant."""
        p1 = nao if p1 is None else p1
        nbytes = self.lib.nbx_eri_packed_bytes(nao, p0, p1)
        if p1 > p0 and nbytes == 0:
            raise ValueError(f"nbx_eri_pack does not cover N = {nao}")
        out = self.empty((max(nbytes // 8, 2),))
        self._call("nbx_eri_pack", nao, p0, p1, self._p(eri) if p1 > p0 else None, self._p(out))
        return out

    def jk_packed(self, packed, dm, p0: int = 0, p1: int | None = None):
        """(1+ndm, N, N) J/K contributions of slab rows [p0,p1) from the packed tiles of eri_pack
        (nbx_jk_packed); additive over slabs like jk_sym."""
        nao = dm.shape[-1]
        p1 = nao if p1 is None else p1
        dm3 = dm.reshape(-1, nao, nao)
        ndm = dm3.shape[0]
        nbytes = self.lib.nbx_jk_packed_worksize(nao, p0, p1, ndm)
        work = self._workspace("jk", nbytes)
        out = self.empty((1 + ndm, nao, nao))
        self._call("nbx_jk_packed", nao, p0, p1, self._p(packed) if p1 > p0 else None, self._p(dm3), ndm,
                   self._p(out), self._p(work), work.numel())
        return out

    def jk_dts_new(self, nao: int):
        """A zeroed Dtot' table for nbx_jk_packed_fock / nbx_huz_cycle_scalars_dts (one per SCF)."""
        nbytes = self.lib.nbx_jk_dts_bytes(nao)
        if nbytes == 0:
            raise ValueError(f"no packed J/K kernel for N = {nao}")
        dts = self.empty((nbytes // 8,))
        self._call("nbx_jk_dts_init", nao, self._p(dts))
        return dts

    def jk_packed_fock(self, packed, dm, hv, want_vhf: bool = True, dts=None):
        """UHF Fock matrices straight from the packed J/K build of the whole tensor
        (nbx_jk_packed_fock): returns (fock, vhf) = (hv + J - K[x], J - K[x]).  ``dts``: the Dtot'
        table of ``dm`` left by ``huz_cycle_scalars_async(..., dts=)`` (skips one launch)."""
        nao = dm.shape[-1]
        nbytes = self.lib.nbx_jk_packed_worksize(nao, 0, nao, 2)
        work = self._workspace("jk", nbytes)
        jk = self.empty((3, nao, nao))
        fock = self.empty((2, nao, nao))
        vhf = self.empty((2, nao, nao)) if want_vhf else None
        self._call("nbx_jk_packed_fock", nao, self._p(packed), self._p(dm), self._p(hv), self._p(jk), self._p(fock),
                   self._p(vhf), self._p(work), work.numel(), self._p(dts))
        return fock, vhf

    def jk_synth(self, nao: int, dm, p0: int = 0, p1: int | None = None, seed: int = 20250829):
        """J/K against the synthetic (pq|rs) generated in registers (no ERI in memory)."""
        p1 = nao if p1 is None else p1
        dm3 = dm.reshape(-1, nao, nao)
        ndm = dm3.shape[0]
        nbytes = self.lib.nbx_jk_dense_worksize(nao, p1 - p0, ndm)
        work = self._workspace("jk", nbytes)
        out = self.empty((1 + ndm, p1 - p0, nao))
        self._call("nbx_jk_synth", nao, p0, p1, seed, self._p(dm3), ndm, self._p(out), self._p(work), work.numel())
        return out

    def jk_synth_sym(self, nao: int, dm, p0: int = 0, p1: int | None = None, seed: int = 20250829):
        """Additive symmetric form of jk_synth: (1+ndm, N, N) contributions of slab rows [p0,p1),
        only the tiles q <= p generated (nbx_jk_synth_sym)."""
        p1 = nao if p1 is None else p1
        dm3 = dm.reshape(-1, nao, nao)
        ndm = dm3.shape[0]
        nbytes = self.lib.nbx_jk_synth_sym_worksize(nao, p0, p1, ndm)
        work = self._workspace("jk", nbytes)
        out = self.empty((1 + ndm, nao, nao))
        self._call("nbx_jk_synth_sym", nao, p0, p1, seed, self._p(dm3), ndm, self._p(out), self._p(work),
                   work.numel())
        return out

    # ------------------------------------------------------------------ GEMM
    def gemm(self, a, b, ta: str = "N", tb: str = "N", alpha: float = 1.0, beta: float = 0.0, out=None):
        """op(a) @ op(b) for 2-D operands, or batched over a shared leading axis (3-D)."""
        batched = a.dim() == 3 or b.dim() == 3
        a3 = a if a.dim() == 3 else a.unsqueeze(0)
        b3 = b if b.dim() == 3 else b.unsqueeze(0)
        batch = max(a3.shape[0], b3.shape[0])
        m, k = (a3.shape[2], a3.shape[1]) if ta == "T" else (a3.shape[1], a3.shape[2])
        k2, n = (b3.shape[2], b3.shape[1]) if tb == "T" else (b3.shape[1], b3.shape[2])
        if k != k2:
            raise ValueError(f"gemm: inner dimensions differ ({k} vs {k2})")
        if out is None:
            out = self.empty((batch, m, n)) if batched else self.empty((m, n))
        sa = a3.shape[1] * a3.shape[2] if a3.shape[0] > 1 else 0
        sb = b3.shape[1] * b3.shape[2] if b3.shape[0] > 1 else 0
        self._call("nbx_gemm", ta.encode(), tb.encode(), m, n, k, alpha, self._p(a3), a3.shape[2], sa,
                   self._p(b3), b3.shape[2], sb, beta, self._p(out), n, m * n, batch)
        return out

    def density_occ(self, c, nocc):
        """D[x] = C[x][:, :nocc[x]] C[x][:, :nocc[x]]^T for (batch, N, N) MO coefficients whose
        columns are in aufbau order: the occupied block is addressed through lda, nothing is copied."""
        batch, n = c.shape[0], c.shape[-1]
        if any(int(x) <= 0 for x in nocc):
            out = self.zeros((batch, n, n))
        else:
            out = self.empty((batch, n, n))  # beta = 0: every element is written
        if len(set(int(x) for x in nocc)) == 1:
            k = int(nocc[0])
            if k > 0:
                self.gemm_raw("N", "T", n, n, k, 1.0, c, n, n * n, c, n, n * n, 0.0, out, n, n * n, batch)
            return out
        for x in range(batch):
            k = int(nocc[x])
            if k > 0:
                self.gemm_raw("N", "T", n, n, k, 1.0, c[x], n, 0, c[x], n, 0, 0.0, out[x], n, 0, 1)
        return out

    def gemm_raw(self, ta, tb, m, n, k, alpha, a, lda, sa, b, ldb, sb, beta, c, ldc, sc, batch):
        self._call("nbx_gemm", ta.encode(), tb.encode(), m, n, k, alpha, self._p(a), lda, sa, self._p(b), ldb, sb,
                   beta, self._p(c), ldc, sc, batch)

    # ------------------------------------------------------------------ Fock pieces
    def fock_uhf(self, hcore, vemb, jk, want_vhf: bool = True):
        nao = jk.shape[-1]
        fock = self.empty((2, nao, nao))
        vhf = self.empty((2, nao, nao)) if want_vhf else None
        self._call("nbx_fock_uhf", nao, self._p(hcore), hcore.dim(), self._p(vemb), self._p(jk), self._p(fock),
                   self._p(vhf))
        return fock, vhf

    def huzinaga_sym(self, fds, kappa: float, fock_io=None):
        nao = fds.shape[-1]
        batch = 1 if fds.dim() == 2 else fds.shape[0]
        hz = self.torch.empty_like(fds)
        self._call("nbx_huzinaga_sym", nao, batch, self._p(fds), kappa, self._p(hz), self._p(fock_io))
        return hz

    def huzinaga_fused(self, fock, ds, kappa: float, want_fock: bool = True):
        """(hz, fock + hz) with hz = -kappa (F DS + (F DS)^T): product and symmetrisation in one
        launch (nbx_huzinaga_fused); ``fock`` itself is left untouched."""
        nao = fock.shape[-1]
        batch = 1 if fock.dim() == 2 else fock.shape[0]
        hz = self.torch.empty_like(fock)
        out = self.torch.empty_like(fock) if want_fock else None
        self._call("nbx_huzinaga_fused", nao, batch, self._p(fock), self._p(ds), kappa, self._p(hz), self._p(out))
        return hz, out

    def trace_prod(self, a, b) -> np.ndarray:
        """einsum('...ij,...ji->...') on device; returns host floats."""
        nao = a.shape[-1]
        batch = 1 if a.dim() == 2 else a.shape[0]
        out = (c_double * batch)()
        self._call("nbx_trace_prod", nao, batch, self._p(a), self._p(b), out)
        res = np.array(out[:], dtype=np.float64)
        return res[0] if a.dim() == 2 else res

    def huz_cycle_scalars(self, hcore, vemb, vhf, hz, dm, dm_old) -> np.ndarray:
        nao = dm.shape[-1]
        out = (c_double * 4)()
        self._call("nbx_huz_cycle_scalars", nao, self._p(hcore), hcore.dim(), self._p(vemb), self._p(vhf),
                   self._p(hz), self._p(dm), self._p(dm_old), out)
        return np.array(out[:], dtype=np.float64)

    def huz_cycle_scalars_async(self, hcore, vemb, vhf, hz, dm, dm_old, extra=None, dts=None):
        """Same four scalars without stalling the stream: returns a handle whose ``get()`` waits
        for (only) the copy of those 32 bytes, so later work can be queued before it is read.
        ``extra``: a small device tensor to bring back with them (``get_extra()``)."""
        nao = dm.shape[-1]
        ntail = 0 if extra is None else int(extra.numel())
        # the kernel's last workgroup stores the results straight into pinned host memory: no copy
        # (a device-to-host copy per cycle costs a launch and a cache flush in the middle of the chain)
        if ntail > 64:
            raise ValueError("huz_cycle_scalars_async: at most 64 status words")
        h_out = self._pin_ring[self._pin_next][: 4 + ntail + 1]
        h_out[4 + ntail] = 0.0  # the kernel's "all stored" word
        self._pin_next = (self._pin_next + 1) % self._PIN_SLOTS
        # ``dts``: a table from jk_dts_new() that the kernel fills with Dtot' of ``dm`` for the next
        # packed J/K build
        self._call("nbx_huz_cycle_scalars_dts", nao, self._p(hcore), hcore.dim(), self._p(vemb), self._p(vhf),
                   self._p(hz), self._p(dm), self._p(dm_old), self._p(h_out), self._p(extra), ntail, self._p(dts))
        return _PendingScalars(self.torch, None, ntail, host=h_out)

    # ------------------------------------------------------------------ fused SCF cycle (one call per cycle)
    def huz_cycle_state(self, nao, nelec, packed, hv, ds, s_b, x, dts, diis_space: int = 6, eri=None, p0: int = 0,
                        p1: int | None = None):
        """Everything one Huzinaga SCF keeps on the device for ``huz_cycle``: the state block of
        ``nbx_huz_cycle`` (workspaces, scratch matrices, the DIIS ring with its Pulay matrix) and three
        rotating sets of per-cycle results (the loop runs one cycle ahead and may hand back the cycle
        before the last).  Returns a small holder object.

        The J/K build of a cycle runs on ``packed`` (nbx_eri_pack of the rows [p0, p1) of the first AO index:
        the packed kernel) or, with ``packed`` None, on ``eri`` (those rows of the dense tensor: the symmetric
        kernel, every N); [p0, p1) = the whole range unless this rank holds a slab."""
        torch, lib = self.torch, self.lib
        n = int(nao)
        nsq = 2 * n * n
        p1 = n if p1 is None else int(p1)
        p0 = int(p0)
        if packed is None and eri is None and p1 > p0:
            raise ValueError("huz_cycle_state: the packed tiles or the dense (pq|rs) slab are needed")

        class Holder:
            pass

        h = Holder()
        h.n, h.nelec = n, (int(nelec[0]), int(nelec[1]))
        h.keep = [packed, eri, hv, ds, s_b, x, dts]
        h.jk = self.empty((3, n, n))
        h.fock, h.vhf, h.fock2, h.tmp, h.fo = (self.empty((2, n, n)) for _ in range(5))
        if packed is not None:
            jk_bytes = int(lib.nbx_jk_packed_worksize(n, p0, p1, 2))
        else:
            jk_bytes = int(lib.nbx_jk_dense_sym_worksize(n, p0, p1, 2))
        h.jk_work = torch.empty(max(jk_bytes, 256), dtype=torch.uint8, device=self.device)
        eig_bytes = max(int(lib.nbx_eigh_worksize(n, 2)), int(lib.nbx_purify_worksize(n, 2)))
        h.eig_work = torch.empty(max(eig_bytes, 256), dtype=torch.uint8, device=self.device)
        h.geig_work = torch.empty(max(int(lib.nbx_geig_refine_worksize(n, 2)), 256), dtype=torch.uint8, device=self.device)
        h.diis_xs, h.diis_es = self.empty((diis_space, nsq)), self.empty((diis_space, nsq))
        hm = np.zeros((diis_space + 1, diis_space + 1))
        hm[0, 1:] = hm[1:, 0] = 1
        h.diis_h = self.asarray(hm)
        h.diis_coef = self.diis_coef_buffer(diis_space)
        h.diis_xprev = self.empty(nsq)
        h.sets = [{"c": self.empty((2, n, n)), "v": self.empty((2, n, n)), "w": self.empty((2, n)),
                   "dm": self.empty((2, n, n)), "hz": self.empty((2, n, n)),
                   "status": torch.empty(2, dtype=torch.int32, device=self.device)} for _ in range(3)]
        if ds is None:  # the mu-shift cycle (mu_cycle): a cycle's Fock matrix and vhf belong to its result set
            for out in h.sets:
                out["fock"], out["vhf"] = out.pop("hz"), self.empty((2, n, n))
        st = _nbx.HuzState()
        st.nao, st.nocc_a, st.nocc_b = n, h.nelec[0], h.nelec[1]
        for name, t in (("d_packed", packed), ("d_hv", hv), ("d_ds", ds), ("d_sb", s_b), ("d_x", x), ("d_dts", dts),
                        ("d_jk", h.jk), ("d_fock", h.fock), ("d_vhf", h.vhf), ("d_fock2", h.fock2), ("d_tmp", h.tmp),
                        ("d_fo", h.fo), ("d_jk_work", h.jk_work), ("d_eig_work", h.eig_work),
                        ("d_geig_work", h.geig_work), ("d_diis_xs", h.diis_xs), ("d_diis_es", h.diis_es),
                        ("d_diis_h", h.diis_h), ("d_diis_coef", h.diis_coef), ("d_diis_xprev", h.diis_xprev),
                        ("d_eri", eri if packed is None else None)):
            setattr(st, name, (t.data_ptr() or None) if t is not None else None)
        st.jk_work_bytes, st.eig_work_bytes, st.geig_work_bytes = h.jk_work.numel(), h.eig_work.numel(), h.geig_work.numel()
        st.diis_space = diis_space
        st.jk_kind = _nbx.HUZ_JK_PACKED if packed is not None else _nbx.HUZ_JK_SYM
        st.jk_p0, st.jk_p1 = p0, p1
        h.st = st
        return h

    def huz_cycle(self, h, dm_in, c_in, out, tracked: bool, refine_iters: int, diis_mode: int, diis_slot: int,
                  diis_nd: int, dts_ready: bool, reduce=None):
        """Queue one SCF cycle (nbx_huz_cycle) writing into the result set ``out``; returns the handle of
        its scalars + status words (read one cycle late, like ``huz_cycle_scalars_async``).  ``tracked``:
        False / 0 guarded eigensolve, True / 1 tracked refinement, 2 density by purification (``refine_iters``
        then caps its steps; no orbitals are written).

        ``reduce``: for a run over several ranks -- a callable that sums the ranks' (3,N,N) J/K partials in
        place with a collective queued on this stream (``Shards.all_reduce``); the cycle is then two C calls
        around it (nbx_huz_cycle_jk, nbx_huz_cycle_post), with the same look-ahead as the one-rank cycle."""
        h_out = self._pin_ring[self._pin_next][:7]
        h_out[6] = 0.0  # the kernel's "all stored" word
        self._pin_next = (self._pin_next + 1) % self._PIN_SLOTS
        if reduce is None:
            self._call("nbx_huz_cycle", ctypes.byref(h.st), self._p(dm_in), self._p(c_in), self._p(out["dm"]),
                       self._p(out["c"]), self._p(out["v"]), self._p(out["w"]), self._p(out["hz"]), int(tracked),
                       int(refine_iters), int(diis_mode), int(diis_slot), int(diis_nd), 1 if dts_ready else 0,
                       self._p(h_out), self._p(out["status"]))
        else:
            self._call("nbx_huz_cycle_jk", ctypes.byref(h.st), self._p(dm_in))
            reduce(h.jk)
            self._call("nbx_huz_cycle_post", ctypes.byref(h.st), self._p(dm_in), self._p(c_in), self._p(out["dm"]),
                       self._p(out["c"]), self._p(out["v"]), self._p(out["w"]), self._p(out["hz"]), int(tracked),
                       int(refine_iters), int(diis_mode), int(diis_slot), int(diis_nd), self._p(h_out),
                       self._p(out["status"]))
        return _PendingScalars(self.torch, None, 2, host=h_out)

    # ------------------------------------------------------------------ fused mu-shift SCF cycle (one call per cycle)
    def mu_cycle_state(self, nao, nelec, packed, h1e, s_b, x, eri=None, p0: int = 0, p1: int | None = None):
        """The state block of ``nbx_mu_cycle`` (PySCF's scf.hf.kernel cycle behind nbed/driver.py:533): as
        ``huz_cycle_state`` with ``h1e`` (2,N,N) for ``hv``, no Huzinaga operands, the DIIS ring sized for CDIIS
        (space 8) and result sets that carry the cycle's Fock matrix and vhf."""
        return self.huz_cycle_state(nao, nelec, packed, h1e, None, s_b, x, None, diis_space=8, eri=eri, p0=p0, p1=p1)

    def _pin_slot(self, n: int):
        h_out = self._pin_ring[self._pin_next][:n]
        h_out[n - 1] = 0.0  # the kernel's "all stored" word
        self._pin_next = (self._pin_next + 1) % self._PIN_SLOTS
        return h_out

    def mu_cycle(self, h, dm_in, fock_in, c_in, out, tracked: bool, refine_iters: int, diis_on: bool, diis_slot: int,
                 diis_nd: int, want_grad: bool = True, reduce=None):
        """Queue one cycle of the mu-shift SCF (nbx_mu_cycle) from the previous cycle's density and Fock matrix
        into the result set ``out`` (dm, fock, vhf, c, v, w, status); returns the handle of its scalars
        (``get()``: E_alpha, E_beta, |dD| per spin; ``get_extra()``: the eigensolver's status words;
        ``get_dtail()``: the two orbital-gradient sums).  ``reduce``: as for ``huz_cycle`` -- the cycle is then
        nbx_mu_cycle_solve, nbx_huz_cycle_jk, the collective, nbx_mu_cycle_fock_post."""
        ndt = 2 if want_grad else 0
        h_out = self._pin_slot(4 + 2 + ndt + 1)
        mode = 1 if tracked else 0
        if reduce is None:
            self._call("nbx_mu_cycle", ctypes.byref(h.st), self._p(dm_in), self._p(fock_in), self._p(c_in),
                       self._p(out["dm"]), self._p(out["fock"]), self._p(out["vhf"]), self._p(out["c"]), self._p(out["v"]),
                       self._p(out["w"]), mode, int(refine_iters), 1 if diis_on else 0, int(diis_slot), int(diis_nd),
                       1 if want_grad else 0, self._p(h_out), self._p(out["status"]))
        else:
            self._call("nbx_mu_cycle_solve", ctypes.byref(h.st), self._p(dm_in), self._p(fock_in), self._p(c_in),
                       self._p(out["dm"]), self._p(out["c"]), self._p(out["v"]), self._p(out["w"]), mode,
                       int(refine_iters), 1 if diis_on else 0, int(diis_slot), int(diis_nd), self._p(out["status"]))
            self._call("nbx_huz_cycle_jk", ctypes.byref(h.st), self._p(out["dm"]))
            reduce(h.jk)
            self._call("nbx_mu_cycle_fock_post", ctypes.byref(h.st), self._p(out["dm"]), self._p(dm_in),
                       self._p(out["c"]) if want_grad else None, self._p(out["fock"]), self._p(out["vhf"]), mode,
                       self._p(out["status"]), self._p(h_out))
        return _PendingScalars(self.torch, None, 2, host=h_out, ndtail=ndt)

    def mu_cycle_fock(self, h, dm, out, reduce=None):
        """Fock matrix, vhf and energy of a given density (the starting density of ``kernel()``) into ``out``:
        nbx_mu_cycle_fock with no orbitals and no status words."""
        h_out = self._pin_slot(4 + 1)
        if reduce is None:
            self._call("nbx_mu_cycle_fock", ctypes.byref(h.st), self._p(dm), self._p(dm), None, self._p(out["fock"]),
                       self._p(out["vhf"]), -1, None, self._p(h_out))
        else:
            self._call("nbx_huz_cycle_jk", ctypes.byref(h.st), self._p(dm))
            reduce(h.jk)
            self._call("nbx_mu_cycle_fock_post", ctypes.byref(h.st), self._p(dm), self._p(dm), None,
                       self._p(out["fock"]), self._p(out["vhf"]), -1, None, self._p(h_out))
        return _PendingScalars(self.torch, None, 0, host=h_out)

    # ------------------------------------------------------------------ density-fitted J/K (an extra: SURVEY 7 step 5)
    def df_synth(self, nao: int, l0: int, l1: int, scale: float | None = None, seed: int | None = None):
        """(l1 - l0, N, N) synthetic three-index factor B_L (symmetric in its last two indices), generated on the
        device: ``synth.df_factor`` is the same array on the host."""
        from . import synth

        out = self.empty((l1 - l0, nao, nao))
        chunk = 32768
        for a in range(l0, l1, chunk):
            b = min(l1, a + chunk)
            self._call("nbx_df_synth", nao, a, b, int(synth.SEED if seed is None else seed),
                       float(synth.df_scale(nao) if scale is None else scale), self._p(out[a - l0:]))
        return out

    def jk_df(self, b, c, nocc):
        """J and K from the factor ``b`` (naux, N, N) and the occupied orbitals: ``c`` is (N, N) or (ndm, N, N) with
        the occupied orbitals in its first ``nocc[x]`` columns.  Returns (1 + ndm, N, N): J, K_a[, K_b] -- partial
        sums if ``b`` is a slab of the auxiliary basis (nbx_jk_df)."""
        c3 = c if c.dim() == 3 else c.unsqueeze(0)
        ndm, n = int(c3.shape[0]), int(c3.shape[-1])
        occ = [int(x) for x in (nocc if hasattr(nocc, "__len__") else [nocc])]
        if len(occ) != ndm:
            raise ValueError(f"jk_df: {ndm} coefficient matrices but {len(occ)} occupation counts")
        out = self.empty((1 + ndm, n, n))
        nbytes = int(self.lib.nbx_jk_df_worksize(n, ndm, max(occ) if occ else 0))
        work = self._workspace("jk_df", nbytes)
        self._call("nbx_jk_df", n, int(b.shape[0]), self._p(b), ndm, self._p(c3), (ctypes.c_int64 * ndm)(*occ), self._p(out),
                   self._p(work), nbytes)
        return out

    def async_to_host(self, d_vals):
        """Stream-ordered copy of a small device tensor to pinned memory; ``.get()`` waits for it only."""
        return _PendingScalars(self.torch, d_vals, 0, ncore=int(d_vals.numel()))

    def diis_coef_buffer(self, space: int):
        """Zeroed device buffer for the coefficients of a DIIS ring of ``space`` vectors and, behind them, the
        Pulay solver's warm-start state (nbx_diis_coef_doubles)."""
        return self.zeros(int(self.lib.nbx_diis_coef_doubles(space)))

    def diis_update(self, space: int, slot: int, nd: int, x, xprev, xs, es, h, coef):
        """Device-resident pyscf.lib.diis.DIIS.update step; ``xprev`` becomes the extrapolated vector."""
        self._call("nbx_diis_update", x.numel(), space, slot, nd, self._p(x), self._p(xprev), self._p(xs),
                   self._p(es), self._p(h), self._p(coef))
        return xprev

    def diis_update_err(self, space: int, slot: int, nd: int, x, err, out, xs, es, h, coef):
        """pyscf.lib.diis.DIIS.update(x, xerr) on the device (CDIIS); ``out`` receives the extrapolation."""
        self._call("nbx_diis_update_err", x.numel(), space, slot, nd, self._p(x), self._p(err), self._p(out),
                   self._p(xs), self._p(es), self._p(h), self._p(coef))
        return out

    def vo_sumsq(self, fmo, nocc):
        """Device (2,) tensor: squared Frobenius norm of the virtual-occupied block of each fmo[x]."""
        out = self.empty(2)
        self._call("nbx_vo_sumsq", fmo.shape[-1], self._p(fmo), int(nocc[0]), int(nocc[1]), self._p(out))
        return out

    def axpby(self, a: float, x, b: float, y):
        self._call("nbx_axpby", x.numel(), a, self._p(x), b, self._p(y))
        return y

    def add(self, x, y):
        """x + y (new array)."""
        out = y.clone()
        return self.axpby(1.0, x, 1.0, out)

    def lincomb(self, coef, vecs, out=None):
        nvec, n = vecs.shape[0], vecs[0].numel()
        if out is None:
            out = self.empty(vecs.shape[1:])
        c = (c_double * nvec)(*[float(x) for x in coef])
        self._call("nbx_lincomb", n, nvec, c, self._p(vecs), n, self._p(out))
        return out

    def dots(self, x, vecs) -> np.ndarray:
        nvec, n = vecs.shape[0], x.numel()
        out = (c_double * nvec)()
        self._call("nbx_dots", n, nvec, self._p(x), self._p(vecs), vecs[0].numel(), out)
        return np.array(out[:], dtype=np.float64)

    def transpose(self, a):
        """Swap the last two axes (materialised)."""
        if a.dim() == 2:
            rows, cols, batch = a.shape[0], a.shape[1], 1
            out = self.empty((cols, rows))
        else:
            batch, rows, cols = a.shape
            out = self.empty((batch, cols, rows))
        self._call("nbx_transpose", rows, cols, batch, self._p(a), self._p(out))
        return out

    def scale_cols(self, a, s):
        rows, cols = a.shape[-2], a.shape[-1]
        batch = 1 if a.dim() == 2 else a.shape[0]
        self._call("nbx_scale_cols", rows, cols, batch, self._p(s), self._p(a))
        return a

    # ------------------------------------------------------------------ eigh / svd
    def eigh(self, a, check: bool = False, v0=None, refine_iters: int = 3):
        """Eigenpairs of symmetric ``a`` (lower triangle, ascending); ``v0``: warm-start vectors;
        ``refine_iters``: refinement iterations queued ahead of the Jacobi fallback (a launch-count
        knob, results do not depend on it)."""
        n = a.shape[-1]
        batch = 1 if a.dim() == 2 else a.shape[0]
        nbytes = self.lib.nbx_eigh_worksize(n, batch)
        work = self._workspace("eigh", nbytes)
        w = self.empty(a.shape[:-1])
        v = self.torch.empty_like(a)
        self._call("nbx_eigh_warm_ex", n, batch, self._p(a), self._p(v0), self._p(w), self._p(v), self._p(work),
                   work.numel(), int(refine_iters))
        off = self.lib.nbx_eigh_status_offset(n, batch)
        # device view of the status words (1000 + k: refined in k iterations; else Jacobi sweeps)
        self.last_eigh_status_d = work[off:off + 4 * batch].view(self.torch.int32)
        if check:
            sweeps = (c_int * batch)()
            self._call("nbx_eigh_status", n, batch, self._p(work), sweeps)
            self.last_eigh_sweeps = list(sweeps)
        return w, v

    def eigh_approx(self, a):
        """Approximate eigenpairs of symmetric ``a`` with nothing read back (nbx_eigh_approx): ``(w, v, status)``
        device tensors, status[b] = 1 usable as warm-start vectors, -1 not.  Safe to queue on a ``side()`` stream."""
        torch = self.torch
        a3 = a if a.dim() == 3 else a.reshape(1, *a.shape)
        batch, n = int(a3.shape[0]), int(a3.shape[-1])
        nbytes = int(self.lib.nbx_eigh_approx_worksize(n, batch))
        work = self._workspace("eigh_approx", nbytes)
        w = self.empty(a.shape[:-1])
        v = torch.empty_like(a)
        status = torch.empty(batch, dtype=torch.int32, device=self.device)
        self._call("nbx_eigh_approx", n, batch, self._p(a), self._p(w), self._p(v), self._p(work), nbytes, self._p(status))
        return w, v, status

    def geig_refine(self, fock, ovlp_b, c0, refine_iters: int = 1):
        """Eigenpairs of the pencil (fock[b], S) refined from the previous cycle's S-orthonormal
        vectors ``c0`` (nbx_geig_refine: GEMMs only, no Loewdin transform, NO fallback solver).
        ``ovlp_b``: the overlap repeated per batch entry.  Sets ``last_eigh_status_d``
        (1000 + iterations: accepted; <= 0: not -- the outputs are then meaningless)."""
        n = fock.shape[-1]
        batch = 1 if fock.dim() == 2 else fock.shape[0]
        nbytes = self.lib.nbx_geig_refine_worksize(n, batch)
        work = self._workspace("geig", nbytes)
        w = self.empty(fock.shape[:-1])
        c = self.torch.empty_like(fock)
        status = self.torch.empty(batch, dtype=self.torch.int32, device=self.device)
        self._call("nbx_geig_refine", n, batch, self._p(fock), self._p(ovlp_b), self._p(c0), self._p(w), self._p(c),
                   self._p(status), self._p(work), work.numel(), int(refine_iters))
        self.last_eigh_status_d = status
        return w, c

    def becke_share(self, pts, centres, aij, inv_dist, owner: int):
        """Becke cell weight of atom ``owner`` at the device points ``pts`` (G, 3) -> device (G,) (nbx_becke_share)."""
        out = self.empty(pts.shape[0])
        self._call("nbx_becke_share", pts.shape[0], self._p(pts), centres.shape[0], self._p(centres), self._p(aij),
                   self._p(inv_dist), int(owner), self._p(out))
        return out

    def eval_ao(self, pts, table, deriv: bool = True):
        """Cartesian AO values (G, ncart) and gradients (3, G, ncart) at the device points ``pts`` (nbx_eval_ao);
        ``table``: the device arrays of ``ao_table``."""
        g, nc = int(pts.shape[0]), int(table["ncart"])
        out = self.empty((g, nc))
        dout = self.empty((3, g, nc)) if deriv else None
        self._call("nbx_eval_ao", g, self._p(pts), table["nshell"], self._p(table["shell_i"]), self._p(table["centre"]),
                   self._p(table["comp_lmn"]), self._p(table["exps"]), self._p(table["coefs"]), nc, table["max_prim"],
                   self._p(out), self._p(dout))
        return out, dout

    def xc_rho(self, ao, dao, dm):
        """Spin densities (2, G) and their gradients (2, 3, G) on the grid from the stored AO values ``ao`` (G, nao),
        gradients ``dao`` (3, G, nao) and a symmetric two-spin density matrix (nbx_xc_rho: c = ao D on the matrix
        cores, reduced against ao / dao in the epilogue; both spins in one pass)."""
        g, nao = int(ao.shape[0]), int(ao.shape[1])
        rho, grad = self.empty((2, g)), self.empty((2, 3, g))
        self._call("nbx_xc_rho", g, nao, self._p(ao), self._p(dao), self._p(dm), self._p(rho), self._p(grad))
        return rho, grad

    def xc_functional(self, code: int, rho, grad, w, rho_floor: float):
        """Weighted first derivatives of the exchange-correlation energy density on the grid (nbx_xc_functional,
        analytic): ``(vr (2, G), vec (2, 3, G), sums)`` with ``sums`` a device pair (E_xc, integrated electrons)."""
        g = int(rho.shape[-1])
        vr, vec, sums = self.empty((2, g)), self.empty((2, 3, g)), self.empty(2)
        nbytes = int(self.lib.nbx_xc_functional_worksize(g))
        work = self._workspace("xc_fn", nbytes)
        self._call("nbx_xc_functional", int(code), g, self._p(rho), self._p(grad), self._p(w), float(rho_floor), self._p(vr),
                   self._p(vec), self._p(sums), self._p(work), work.numel())
        return vr, vec, sums

    def xc_vmat(self, ao, dao, vr, vec):
        """v_xc (2, nao, nao), symmetrised, from the weighted derivatives of ``xc_functional`` (nbx_xc_vmat: the
        ``half`` factor is built on the fly as an operand of the product)."""
        g, nao = int(ao.shape[0]), int(ao.shape[1])
        out = self.empty((2, nao, nao))
        nbytes = int(self.lib.nbx_xc_vmat_worksize(g, nao))
        work = self._workspace("xc_vmat", nbytes)
        self._call("nbx_xc_vmat", g, nao, self._p(ao), self._p(dao), self._p(vr), self._p(vec), self._p(out), self._p(work),
                   work.numel())
        return out

    def ao_table(self, basis):
        """The shells of an ``integrals.Basis`` flattened into the device arrays nbx_eval_ao reads."""
        torch = self.torch
        shell_i, centre, comp, exps, coefs = [], [], [], [], []
        for sh, ao0 in zip(basis.shells, basis.shell_ao0):
            k0 = len(exps)
            assert int(ao0) == len(comp), "shells in AO order"
            exps.extend(float(a) for a in sh.exps)
            shell_i.append([int(ao0), len(sh.cart), k0, len(sh.exps)])
            centre.append([float(x) for x in sh.centre])
            for ic, lmn in enumerate(sh.cart):
                if max(lmn) > 3:
                    raise ValueError("nbx_eval_ao covers s, p, d and f shells")
                comp.append([int(lmn[0]), int(lmn[1]), int(lmn[2]), len(coefs)])
                coefs.extend(float(c) for c in sh.coefs[ic])
        assert len(comp) == basis.nao_cart

        def dev(a, dtype):
            return torch.as_tensor(np.asarray(a), dtype=dtype).contiguous().to(self.device)

        return {"nshell": len(shell_i), "ncart": int(basis.nao_cart), "max_prim": max(s_[3] for s_ in shell_i),
                "shell_i": dev(shell_i, torch.int32), "centre": dev(centre, torch.float64),
                "comp_lmn": dev(comp, torch.int32), "exps": dev(exps, torch.float64), "coefs": dev(coefs, torch.float64)}

    def purify(self, f, nocc, max_iter: int = 0):
        """Projector on the ``nocc[b]`` lowest eigenvectors of each symmetric matrix of ``f`` (batch, n, n) by
        trace-correcting purification (nbx_purify) -> ``(p, status)`` device tensors; status[b] > 0 = steps."""
        torch = self.torch
        f = f if f.dim() == 3 else f.reshape(1, *f.shape)
        batch, n = int(f.shape[0]), int(f.shape[-1])
        nocc = [int(nocc)] * 2 if np.isscalar(nocc) else [int(x) for x in nocc]
        if len(nocc) not in (1, 2, batch) or (len(nocc) > 2 and len(set(nocc)) > 1) or (batch > 2 and len(set(nocc)) > 1):
            raise ValueError("purify: one occupation per matrix; batches of more than two need equal occupations")
        p = torch.empty_like(f)
        status = torch.zeros(batch, dtype=torch.int32, device=self.device)
        nbytes = int(self.lib.nbx_purify_worksize(n, batch))
        work = self._workspace("purify", nbytes)
        self._call("nbx_purify", n, batch, self._p(f), nocc[0], nocc[-1], self._p(p), self._p(work), nbytes, int(max_iter),
                   self._p(status))
        return p, status

    def sym_pow_newton_schulz(self, s, p: float, s_host=None, max_iter: int = 28, check_every: int = 4):
        """S^p for p in {-1/2, +1/2, -1} of a symmetric positive definite matrix by the coupled
        Newton-Schulz iteration (Higham, Functions of Matrices, eq. 6.35): with A = S / c,
            Y_0 = A, Z_0 = I;  T = (3 I - Z Y) / 2;  Y <- Y T -> A^1/2,  Z <- T Z -> A^-1/2,
        three MFMA GEMMs per step and nothing else -- the Jacobi eigensolver behind ``sym_pow`` is one
        workgroup per matrix and takes 2.9 ms at N = 148, this takes ~0.2 ms.  c = ||S||_inf bounds the
        spectrum to (0, 1]; convergence (||I - Z Y||_F, quadratic at the end) is read back every
        ``check_every`` steps.  Returns None when it does not converge (S not positive definite, or a
        condition number beyond ~1e6, where the iteration would lose more than the eigen route does):
        the caller then takes the eigendecomposition route.
        Equal to scipy's fractional_matrix_power (nbed/scf/huzinaga_scf.py:128) to ~cond(S) * 1e-16."""
        torch = self.torch
        if p not in (-0.5, 0.5, -1.0) or s.dim() != 2:
            return None
        n = s.shape[-1]
        if s_host is not None:
            c = float(np.abs(np.asarray(s_host)).sum(axis=1).max())
        else:
            c = float(np.abs(self.to_host(s)).sum(axis=1).max())
        if not np.isfinite(c) or c <= 0.0:
            return None
        if hasattr(self.lib, "nbx_sym_pow_ns"):  # the whole iteration queued by one C call (csrc/eigh.hip)
            nbytes = int(self.lib.nbx_sym_pow_ns_worksize(n))
            work = self._workspace("sym_pow_ns", nbytes)
            out = torch.empty_like(s)
            iters = c_int(-1)
            self._call("nbx_sym_pow_ns", n, self._p(s), float(p), c, self._p(out), self._p(work), work.numel(), int(max_iter),
                       int(check_every), ctypes.byref(iters))
            return out if iters.value > 0 else None
        eye = torch.eye(n, dtype=torch.float64, device=self.device)
        eye3 = self.empty((n, n))
        self.axpby(3.0, eye, 0.0, eye3)
        y = self.empty((n, n))
        self.axpby(1.0 / c, s, 0.0, y)  # y = S / c
        z = self.copy(eye)
        prev = None
        done = False
        for it in range(max_iter):
            t = self.copy(eye3)
            self.gemm(z, y, alpha=-1.0, beta=1.0, out=t)  # T' = 3 I - Z Y  (= 2 T)
            if it % check_every == check_every - 1 or done:
                r = self.copy(t)
                self.axpby(-2.0, eye, 1.0, r)  # I - Z Y
                res = float(np.sqrt(self.dots(r.reshape(-1), r.reshape(1, -1))[0]))
                if not np.isfinite(res):
                    return None
                if done or res < 1e-14 * n:
                    break
                # quadratic phase reached: one more step takes the residual to rounding level
                if res < 1e-6 or (prev is not None and res > 0.5 * prev and res < 1e-9):
                    done = True
                prev = res
            y = self.gemm(y, t, alpha=0.5)
            z = self.gemm(t, z, alpha=0.5)
        else:
            return None
        if p == 0.5:
            out, scale = y, float(np.sqrt(c))
        elif p == -0.5:
            out, scale = z, float(1.0 / np.sqrt(c))
        else:
            out, scale = self.gemm(z, z), 1.0 / c
        # scale back and symmetrise (the iterates are symmetric up to rounding)
        sym = self.transpose(out)
        self.axpby(0.5 * scale, out, 0.5 * scale, sym)
        return sym

    def sym_pow_fast(self, s, p: float, s_host=None):
        """S^p: Newton-Schulz (GEMMs) where it applies and converges, else the eigendecomposition."""
        if s.shape[-1] >= 16:
            out = self.sym_pow_newton_schulz(s, p, s_host)
            if out is not None:
                return out
        return self.sym_pow(s, p)

    def sym_pow(self, s, p: float):
        n = s.shape[-1]
        nbytes = self.lib.nbx_sym_pow_worksize(n)
        work = self._workspace("sym_pow", nbytes)
        out = self.torch.empty_like(s)
        self._call("nbx_sym_pow", n, self._p(s), p, self._p(out), self._p(work), work.numel())
        return out

    def svd_right(self, a, check: bool = True):
        """Singular values (descending) and Vt (n x n) of the 2-D matrix ``a`` (m x n)."""
        m, n = a.shape
        nbytes = self.lib.nbx_svd_worksize(m, n)
        work = self._workspace("svd", nbytes)
        s = self.empty((min(m, n),))
        vt = self.empty((n, n))
        self._call("nbx_svd_right", m, n, self._p(a), self._p(s), self._p(vt), self._p(work), work.numel())
        if check:
            sweeps = (c_int * 1)()
            self._call("nbx_svd_status", m, n, self._p(work), sweeps)
            self.last_svd_sweeps = sweeps[0]
        return s, vt

    # ------------------------------------------------------------------ four-index transform
    def ao2mo(self, eri, c1, c2, c3, c4, i0: int = 0, i1: int | None = None, out=None):
        """(i1-i0, n2, n3, n4) chemist-order MO integrals (nbx_ao2mo)."""
        nao = c1.shape[0]
        n1, n2, n3, n4 = c1.shape[1], c2.shape[1], c3.shape[1], c4.shape[1]
        i1 = n1 if i1 is None else i1
        nbytes = self.lib.nbx_ao2mo_worksize(nao, i1 - i0, n2, n3, n4)
        work = self._workspace("ao2mo", nbytes)
        if out is None:
            out = self.empty((i1 - i0, n2, n3, n4))
        elif tuple(out.shape) != (i1 - i0, n2, n3, n4):
            raise ValueError("ao2mo: output tensor has the wrong shape")
        self._call("nbx_ao2mo", nao, self._p(eri), self._p(c1), n1, i0, i1, self._p(c2), n2, self._p(c3), n3,
                   self._p(c4), n4, self._p(out), self._p(work), work.numel())
        return out

    def ao2mo_pair(self, eri, c1, c2, c3, c4, c5, c6, i0: int = 0, i1: int | None = None, out=None, out2=None):
        """((C1 C2|C3 C4), (C1 C2|C5 C6)) in one pass: quarters 1-2 shared (nbx_ao2mo_pair)."""
        nao = c1.shape[0]
        n1, n2, n3, n4, n5, n6 = (c.shape[1] for c in (c1, c2, c3, c4, c5, c6))
        i1 = n1 if i1 is None else i1
        nbytes = self.lib.nbx_ao2mo_pair_worksize(nao, i1 - i0, n2, n4, n6)
        work = self._workspace("ao2mo", nbytes)
        out = self.empty((i1 - i0, n2, n3, n4)) if out is None else out
        out2 = self.empty((i1 - i0, n2, n5, n6)) if out2 is None else out2
        if tuple(out.shape) != (i1 - i0, n2, n3, n4) or tuple(out2.shape) != (i1 - i0, n2, n5, n6):
            raise ValueError("ao2mo_pair: output tensor has the wrong shape")
        self._call("nbx_ao2mo_pair", nao, self._p(eri), self._p(c1), n1, i0, i1, self._p(c2), n2, self._p(c3), n3,
                   self._p(c4), n4, self._p(out), self._p(c5), n5, self._p(c6), n6, self._p(out2), self._p(work),
                   work.numel())
        return out, out2

    def eri_pack_rs(self, eri, nao: int):
        """(pq|rs) with (r, s <= r) packed: (N, N, N(N+1)/2), the input of ao2mo_pair_sym(rs_packed=True)."""
        out = self.empty((nao, nao, nao * (nao + 1) // 2))
        self._call("nbx_eri_pack_rs", nao, self._p(eri), self._p(out))
        return out

    def ao2mo_pair_sym(self, eri, c12, c3, c4, c5=None, c6=None, rs_packed: bool = False, out=None, out2=None):
        """(C12 C12|C3 C4) [and (C12 C12|C5 C6)] over the whole outer range with (ij|kl) = (ji|kl):
        quarters 3-4 on the pairs j <= i only (nbx_ao2mo_pair_sym).  ``rs_packed``: ``eri`` comes from
        eri_pack_rs and quarters 1-2 use (pq|rs) = (pq|sr) as well.  Equal to ao2mo / ao2mo_pair up
        to rounding; the outputs are exactly symmetric in (i, j).  ``out`` / ``out2``: result tensors to
        reuse (a caller that builds repeatedly keeps the 8 n^4-byte allocations out of its loop)."""
        nao, n = c12.shape
        n3, n4 = c3.shape[1], c4.shape[1]
        pair = c5 is not None
        n5, n6 = (c5.shape[1], c6.shape[1]) if pair else (0, 0)
        name = "nbx_ao2mo_pair_sym_rs" if rs_packed else "nbx_ao2mo_pair_sym"
        nbytes = getattr(self.lib, name + "_worksize")(nao, n, n4, n6)
        work = self._workspace("ao2mo", nbytes)
        if out is None:
            out = self.empty((n, n, n3, n4))
        if pair and out2 is None:
            out2 = self.empty((n, n, n5, n6))
        if tuple(out.shape) != (n, n, n3, n4) or (pair and tuple(out2.shape) != (n, n, n5, n6)):
            raise ValueError("ao2mo_pair_sym: output tensor has the wrong shape")
        self._call(name, nao, self._p(eri), self._p(c12), n, self._p(c3), n3, self._p(c4), n4,
                   self._p(out), self._p(c5), n5, self._p(c6), n6, self._p(out2), self._p(work), work.numel())
        return (out, out2) if pair else out

    def ao2mo_synth(self, nao: int, c1, c2, c3, c4, r0: int = 0, r1: int | None = None, seed: int = 20250829):
        """Streamed transform of the synthetic ERI (never stored); partial sum over r in [r0,r1)."""
        n1, n2, n3, n4 = c1.shape[1], c2.shape[1], c3.shape[1], c4.shape[1]
        r1 = nao if r1 is None else r1
        nbytes = self.lib.nbx_ao2mo_synth_worksize(nao, n1, n2, n3, n4)
        work = self._workspace("ao2mo_synth", nbytes)
        out = self.empty((n1, n2, n3, n4))
        self._call("nbx_ao2mo_synth", nao, seed, r0, r1, self._p(c1), n1, self._p(c2), n2, self._p(c3), n3,
                   self._p(c4), n4, self._p(out), self._p(work), work.numel())
        return out

    def ao2mo_synth_pair(self, nao: int, c1, c2, c3, c4, c5, c6, r0: int = 0, r1: int | None = None,
                         seed: int = 20250829):
        """Streamed ((C1 C2|C3 C4), (C1 C2|C5 C6)): integrals and quarters 1-2 generated/computed once."""
        n1, n2, n3, n4, n5, n6 = (c.shape[1] for c in (c1, c2, c3, c4, c5, c6))
        r1 = nao if r1 is None else r1
        nbytes = self.lib.nbx_ao2mo_synth_pair_worksize(nao, n1, n2, n3, n4, n5, n6)
        work = self._workspace("ao2mo_synth", nbytes)
        out, out2 = self.empty((n1, n2, n3, n4)), self.empty((n1, n2, n5, n6))
        self._call("nbx_ao2mo_synth_pair", nao, seed, r0, r1, self._p(c1), n1, self._p(c2), n2, self._p(c3), n3,
                   self._p(c4), n4, self._p(out), self._p(c5), n5, self._p(c6), n6, self._p(out2), self._p(work),
                   work.numel())
        return out, out2

    def chem_to_phys(self, x):
        n1, n2, n3, n4 = x.shape
        out = self.empty((n1, n3, n4, n2))
        self._call("nbx_chem_to_phys", n1, n2, n3, n4, self._p(x), self._p(out))
        return out

    def spinorb_scatter_to_host(self, one_body, two_body, tol: float, h2_scale: float, chunk: int = 32 * 1024 * 1024,
                                nbuf: int = 3):
        """(h1, h2) as numpy arrays with h2 streamed out in pieces: each piece of the flattened
        (2n)^4 tensor is produced into a small device buffer (nbx_spinorb_scatter_range), copied to a
        pinned buffer and drained into the result by host threads -- the tensor is never held on the
        device (60 GB at n = 147) and the copy runs at the pinned-pipeline rate."""
        from concurrent.futures import ThreadPoolExecutor

        torch = self.torch
        n = one_body.shape[-1]
        nq = 2 * n
        total = nq**4
        if total < (1 << 26):
            h1, h2 = self.spinorb_scatter(one_body, two_body, tol, h2_scale)
            return self.to_host(h1), self.to_host(h2)
        h1 = self.empty((nq, nq))
        self._call("nbx_spinorb_scatter_h1", n, self._p(one_body), tol, self._p(h1))
        out = np.empty(total, dtype=np.float64)
        try:
            threads = max(1, min(8, len(os.sched_getaffinity(0))))
        except AttributeError:
            threads = 4
        dbufs = [self.empty(chunk) for _ in range(nbuf)]
        pbufs = [torch.empty(chunk, dtype=torch.float64, pin_memory=True) for _ in range(nbuf)]
        made = [torch.cuda.Event() for _ in range(nbuf)]
        copied = [torch.cuda.Event() for _ in range(nbuf)]
        main = torch.cuda.current_stream(self.device_index)
        copy_stream = torch.cuda.Stream(device=self.device)
        nchunks = (total + chunk - 1) // chunk
        pending = [None] * nbuf
        used = [False] * nbuf

        def drain(pool, k):
            b = k % nbuf
            lo, hi = k * chunk, min(total, (k + 1) * chunk)
            copied[b].synchronize()
            piece = pbufs[b].numpy()[: hi - lo]
            step = (hi - lo + threads - 1) // threads
            return [pool.submit(np.copyto, out[lo + i * step: min(hi, lo + (i + 1) * step)],
                                piece[i * step: min(hi - lo, (i + 1) * step)]) for i in range(threads)]

        with ThreadPoolExecutor(threads) as pool:
            for k in range(nchunks + nbuf - 1):
                if k < nchunks:
                    b = k % nbuf
                    if pending[b] is not None:
                        for f in pending[b]:
                            f.result()  # pinned buffer b has been drained
                    if used[b]:
                        main.wait_event(copied[b])  # device buffer b has been copied out
                    lo, hi = k * chunk, min(total, (k + 1) * chunk)
                    self._call("nbx_spinorb_scatter_range", n, self._p(two_body), tol, h2_scale, lo, hi - lo,
                               self._p(dbufs[b]))
                    made[b].record(main)
                    with torch.cuda.stream(copy_stream):
                        copy_stream.wait_event(made[b])
                        pbufs[b][: hi - lo].copy_(dbufs[b][: hi - lo], non_blocking=True)
                        copied[b].record(copy_stream)
                    used[b] = True
                j = k - (nbuf - 1)
                if j >= 0:
                    pending[j % nbuf] = drain(pool, j)
            for p in pending:
                if p is not None:
                    for f in p:
                        f.result()
        main.wait_stream(copy_stream)
        return self.to_host(h1), out.reshape((nq,) * 4)

    def threshold_scale(self, x, tol: float, scale: float):
        """x <- (|x| < tol ? 0 : x) * scale in place (nbx_threshold_scale)."""
        self._call("nbx_threshold_scale", x.numel(), tol, scale, self._p(x))
        return x

    def spinorb_scatter(self, one_body, two_body, tol: float, h2_scale: float):
        n = one_body.shape[-1]
        h1 = self.empty((2 * n, 2 * n))
        h2 = self.empty((2 * n, 2 * n, 2 * n, 2 * n))
        self._call("nbx_spinorb_scatter", n, self._p(one_body), self._p(two_body), tol, h2_scale, self._p(h1),
                   self._p(h2))
        return h1, h2
