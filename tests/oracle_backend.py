"""A CHECKER backend for CPU-only tests: same interface as nbed_amd.backend.HipBackend, with
every operation done by numpy/LAPACK (the oracle's arithmetic).

It lives under tests/ on purpose: the product (nbed_amd) has no CPU implementation and never
imports this.  Tests inject it (``backend=OracleBackend()``) to exercise the HOST logic of the
product -- loop control, DIIS schedule, convergence rule, localiser bookkeeping, sharding and
the gloo all-gather path -- on machines without a GPU, and as the reference side of the
GPU parity tests.  Arrays are torch CPU float64 tensors so the host code is unchanged.
"""

from __future__ import annotations

import numpy as np
import torch

from oracle import hamiltonian
from oracle import synth


class OracleBackend:
    name = "oracle"

    def __init__(self):
        self.torch = torch
        self.device = torch.device("cpu")
        self.calls: dict[str, int] = {}

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    # ---- plumbing
    def synchronize(self):
        pass

    def use_current_stream(self):
        pass

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float64)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64)

    def asarray(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(torch.float64).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())

    def to_host(self, a):
        return a.detach().numpy().copy() if isinstance(a, torch.Tensor) else np.asarray(a)

    def copy(self, a):
        return a.clone()

    def release_workspaces(self):
        pass

    @staticmethod
    def _np(a):
        return a.detach().numpy()

    # ---- collectives
    def pad_axis(self, a, axis, length):
        if a.shape[axis] == length:
            return a.contiguous()
        shape = list(a.shape)
        shape[axis] = length
        out = self.zeros(shape)
        out.narrow(axis, 0, a.shape[axis]).copy_(a)
        return out

    def all_gather_stack(self, a, group=None):
        import torch.distributed as dist

        world = dist.get_world_size(group)
        pieces = [torch.empty_like(a) for _ in range(world)]
        dist.all_gather(pieces, a.contiguous(), group=group)
        return torch.stack(pieces)

    def all_reduce_sum(self, a, group=None):
        import torch.distributed as dist

        dist.all_reduce(a, op=dist.ReduceOp.SUM, group=group)
        return a

    def reduce_scatter_sum(self, a, group=None):
        """(world * chunk, ...) -> this rank's (chunk, ...) piece of the sum.  gloo has no reduce-scatter: one reduce per
        destination rank (what a reduce-scatter is), each rank keeping the piece it is the destination of."""
        import torch.distributed as dist

        world, rank = dist.get_world_size(group), dist.get_rank(group)
        chunk = a.shape[0] // world
        self._count("reduce_scatter")
        mine = None
        for dst in range(world):
            piece = a[dst * chunk:(dst + 1) * chunk].clone().contiguous()
            dist.reduce(piece, dst=dst, op=dist.ReduceOp.SUM, group=group)
            if dst == rank:
                mine = piece
        return mine

    def unstack_concat(self, stacked, axis, n):
        world = stacked.shape[0]
        moved = stacked.movedim(0, axis)
        shape = list(moved.shape)
        merged = moved.reshape(shape[:axis] + [world * shape[axis + 1]] + shape[axis + 2:])
        return merged.narrow(axis, 0, n).contiguous()

    # ---- kernels
    def synth_eri(self, nao, p0=0, p1=None, seed=synth.SEED):
        p1 = nao if p1 is None else p1
        return self.asarray(synth.eri_block(nao, p0, p1, seed))

    def jk(self, eri, dm, p0=0, p1=None):
        self._count("jk")
        nao = dm.shape[-1]
        p1 = nao if p1 is None else p1
        e = self._np(eri)
        d = self._np(dm).reshape(-1, nao, nao)
        # slab rows only: J[p,q] = sum_rs e[p,q,r,s] Dtot[r,s]; K[p,r] = sum_qs e[p,q,r,s] D[q,s]
        dtot = d.sum(axis=0)
        out = np.empty((1 + d.shape[0], p1 - p0, nao))
        out[0] = np.einsum("pqrs,rs->pq", e, dtot)
        for x in range(d.shape[0]):
            out[1 + x] = np.einsum("pqrs,qs->pr", e, d[x])
        return self.asarray(out)

    use_sym = False

    def __getattribute__(self, name):
        if name == "jk_sym" and not object.__getattribute__(self, "use_sym"):
            raise AttributeError(name)
        return object.__getattribute__(self, name)

    def jk_sym(self, eri, dm, p0=0, p1=None):
        """Additive slab form (nbx_jk_dense_sym): full-size J/K contributions of the pairs
        (p in [p0,p1), q <= p) and their mirror images.  Enabled per instance (``use_sym``)."""
        self._count("jk")
        nao = dm.shape[-1]
        p1 = nao if p1 is None else p1
        e = self._np(eri)
        d = self._np(dm).reshape(-1, nao, nao)
        dtot = d.sum(axis=0)
        out = np.zeros((1 + d.shape[0], nao, nao))
        for p in range(p0, p1):
            for q in range(p + 1):
                t = e[p - p0, q]
                j = float(np.sum(t * dtot))
                out[0, p, q] = j
                out[0, q, p] = j
                for x in range(d.shape[0]):
                    out[1 + x, p] += d[x, q] @ t
                    if q < p:
                        out[1 + x, q] += d[x, p] @ t
        return self.asarray(out)

    def gemm(self, a, b, ta="N", tb="N", alpha=1.0, beta=0.0, out=None):
        self._count("gemm")
        an, bn = self._np(a), self._np(b)
        if ta == "T":
            an = np.swapaxes(an, -1, -2)
        if tb == "T":
            bn = np.swapaxes(bn, -1, -2)
        res = alpha * (an @ bn)
        if out is not None:
            out.copy_(torch.from_numpy(res + beta * self._np(out)))
            return out
        return self.asarray(res)

    def fock_uhf(self, hcore, vemb, jk, want_vhf=True):
        j = self._np(jk)
        vhf = j[0] - j[1:]
        f = self._np(hcore) + vhf
        if vemb is not None:
            f = f + self._np(vemb)
        return self.asarray(f), self.asarray(vhf)

    def huzinaga_sym(self, fds, kappa, fock_io=None):
        f = self._np(fds)
        hz = -kappa * (f + np.swapaxes(f, -1, -2))
        if fock_io is not None:
            fock_io += torch.from_numpy(hz)
        return self.asarray(hz)

    def trace_prod(self, a, b):
        res = np.einsum("...ij,...ji->...", self._np(a), self._np(b))
        return float(res) if res.ndim == 0 else res

    def huz_cycle_scalars(self, hcore, vemb, vhf, hz, dm, dm_old):
        ham = self._np(hcore) + 0.5 * self._np(vhf) + self._np(hz)
        if vemb is not None:
            ham = ham + self._np(vemb)
        e = np.einsum("xij,xji->x", ham, self._np(dm))
        d = np.linalg.norm(self._np(dm) - self._np(dm_old), axis=(-2, -1))
        return np.concatenate([e, d])

    def axpby(self, a, x, b, y):
        yn = self._np(y)
        if b == 0.0:
            yn[...] = a * self._np(x).reshape(yn.shape)
        else:
            yn[...] = a * self._np(x).reshape(yn.shape) + b * yn
        return y

    def add(self, x, y):
        return x + y

    def lincomb(self, coef, vecs, out=None):
        res = np.tensordot(np.asarray(coef, dtype=np.float64), self._np(vecs), axes=(0, 0))
        if out is not None:
            out.copy_(torch.from_numpy(res))
            return out
        return self.asarray(res)

    def dots(self, x, vecs):
        v = self._np(vecs)
        return v.reshape(v.shape[0], -1) @ self._np(x).reshape(-1)

    def transpose(self, a):
        return self.asarray(np.swapaxes(self._np(a), -1, -2))

    def scale_cols(self, a, s):
        an = self._np(a)
        an *= self._np(s)[..., None, :]
        return a

    def eigh(self, a, check=False, v0=None):
        self._count("eigh")
        w, v = np.linalg.eigh(self._np(a))
        self.last_eigh_sweeps = [1]
        return self.asarray(w), self.asarray(v)

    def sym_pow(self, s, p):
        w, u = np.linalg.eigh(self._np(s))
        return self.asarray((u * w**p) @ u.T)

    def svd_right(self, a, check=True):
        self._count("svd")
        _, s, vt = np.linalg.svd(self._np(a))
        self.last_svd_sweeps = 1
        return self.asarray(s), self.asarray(vt)

    def ao2mo(self, eri, c1, c2, c3, c4, i0=0, i1=None):
        self._count("ao2mo")
        c1n = self._np(c1)
        i1 = c1n.shape[1] if i1 is None else i1
        return self.asarray(hamiltonian.ao2mo_full(self._np(eri), c1n[:, i0:i1], self._np(c2), self._np(c3),
                                                   self._np(c4)))

    def ao2mo_synth(self, nao, c1, c2, c3, c4, r0=0, r1=None, seed=synth.SEED):
        """Partial sum over r in [r0,r1) of the streamed transform (pairs s <= r and their mirrors),
        layout of nbx_ao2mo_synth, by einsum on the dense synthetic tensor."""
        self._count("ao2mo_synth")
        r1 = nao if r1 is None else r1
        eri = synth.eri_dense(nao, seed)
        c1n, c2n, c3n, c4n = (self._np(c) for c in (c1, c2, c3, c4))
        out = np.zeros((c1n.shape[1], c2n.shape[1], c3n.shape[1], c4n.shape[1]))
        for r in range(r0, r1):
            x = np.einsum("pi,qj,pqs->ijs", c1n, c2n, eri[:, :, r, : r + 1])  # half transform, s <= r
            out += np.einsum("ijs,k,sl->ijkl", x, c3n[r], c4n[: r + 1])
            out += np.einsum("ijs,sk,l->ijkl", x[:, :, :r], c3n[:r], c4n[r])
        return self.asarray(out)

    def ao2mo_synth_pair(self, nao, c1, c2, c3, c4, c5, c6, r0=0, r1=None, seed=synth.SEED):
        return (self.ao2mo_synth(nao, c1, c2, c3, c4, r0, r1, seed), self.ao2mo_synth(nao, c1, c2, c5, c6, r0, r1, seed))

    def chem_to_phys(self, x):
        return self.asarray(self._np(x).transpose(0, 2, 3, 1))

    def threshold_scale(self, x, tol, scale):
        xn = self._np(x)
        xn[np.abs(xn) < tol] = 0.0
        xn *= scale
        return x

    def spinorb_scatter(self, one_body, two_body, tol, h2_scale):
        h1, h2 = hamiltonian.spinorb_from_spatial(self._np(one_body), self._np(two_body), tol)
        return self.asarray(h1), self.asarray(h2 * h2_scale)


class _Done:
    """Handle of a 'queued' cycle's scalars (the HIP backend's _PendingScalars, already complete here)."""

    def __init__(self, vals, extra=None, dtail=None):
        self._vals = np.asarray(vals, dtype=np.float64)
        self._extra = None if extra is None else np.asarray(extra, dtype=np.int64)
        self._dtail = None if dtail is None else np.asarray(dtail, dtype=np.float64)

    def get_dtail(self):
        return self._dtail

    def get(self):
        return self._vals.copy()

    def get_extra(self):
        return self._extra


class OracleLookaheadBackend(OracleBackend):
    """The checker backend WITH the look-ahead interface of HipBackend (``huz_cycle_state`` / ``huz_cycle``,
    ``purify``, ``geig_refine``, ``density_occ``, ``huz_cycle_scalars_async``, ``async_to_host``): numpy
    restatements of what nbx_huz_cycle / nbx_huz_cycle_jk + _post queue (csrc/scf_cycle.hip), step for step, with
    the status words the device solvers would report (purification: steps; cold guarded solve: sweeps; accepted
    refinement: 1000 + iterations).  CPU tests drive the product's one-call-per-cycle loop through it -- the
    purified -> cold -> guarded -> tracked schedule, the DIIS ring bookkeeping the loop does itself, the
    one-cycle-late convergence test, and (gloo, world 2) the all-reduce between the two halves of a cycle."""

    use_sym = True
    PURIFY_STEPS = 30

    def density_occ(self, c, nocc):
        cn = self._np(c)
        return self.asarray(np.stack([cn[x][:, : int(nocc[x])] @ cn[x][:, : int(nocc[x])].T for x in range(cn.shape[0])]))

    def async_to_host(self, d_vals):
        return _Done(self._np(d_vals).reshape(-1))

    def huz_cycle_scalars_async(self, hcore, vemb, vhf, hz, dm, dm_old, extra=None, dts=None):
        vals = self.huz_cycle_scalars(hcore, vemb, vhf, hz, dm, dm_old)
        return _Done(vals, None if extra is None else self._np(extra).reshape(-1))

    def eigh(self, a, check=False, v0=None, refine_iters=3):
        w, v = super().eigh(a, check=check, v0=v0)
        batch = 1 if a.dim() == 2 else a.shape[0]
        self.last_eigh_status_d = torch.full((batch,), 1 if v0 is None else 1001, dtype=torch.int32)
        return w, v

    def geig_refine(self, fock, ovlp_b, c0, refine_iters=1):
        import scipy.linalg

        self._count("geig_refine")
        f, s = self._np(fock), self._np(ovlp_b)
        res = [scipy.linalg.eigh(f[x], s[x]) for x in range(f.shape[0])]
        self.last_eigh_status_d = torch.full((f.shape[0],), 1001, dtype=torch.int32)
        return self.asarray(np.stack([r[0] for r in res])), self.asarray(np.stack([r[1] for r in res]))

    def purify(self, f, nocc, max_iter=0):
        self._count("purify")
        fn = self._np(f if f.dim() == 3 else f.reshape(1, *f.shape))
        nocc = [int(nocc)] * 2 if np.isscalar(nocc) else [int(x) for x in nocc]
        ps = []
        for x in range(fn.shape[0]):
            _, v = np.linalg.eigh(fn[x])
            k = nocc[min(x, len(nocc) - 1)]
            ps.append(v[:, :k] @ v[:, :k].T)
        return self.asarray(np.stack(ps)), torch.full((fn.shape[0],), self.PURIFY_STEPS, dtype=torch.int32)

    def huz_cycle_state(self, nao, nelec, packed, hv, ds, s_b, x, dts, diis_space=6, eri=None, p0=0, p1=None):
        class Holder:
            pass

        h = Holder()
        n = int(nao)
        h.n, h.nelec = n, (int(nelec[0]), int(nelec[1]))
        h.eri, h.p0, h.p1 = eri, int(p0), n if p1 is None else int(p1)
        h.hv, h.ds, h.s_b, h.x = hv, ds, s_b, x
        h.space = diis_space
        h.xs, h.es = np.zeros((diis_space, 2 * n * n)), np.zeros((diis_space, 2 * n * n))
        h.H = np.zeros((diis_space + 1, diis_space + 1))
        h.H[0, 1:] = h.H[1:, 0] = 1
        h.xprev = None
        h.jk = None
        h.sets = [{"c": self.empty((2, n, n)), "v": self.empty((2, n, n)), "w": self.empty((2, n)),
                   "dm": self.empty((2, n, n)), "hz": self.empty((2, n, n)),
                   "status": torch.zeros(2, dtype=torch.int32)} for _ in range(3)]
        return h

    def huz_cycle(self, h, dm_in, c_in, out, tracked, refine_iters, diis_mode, diis_slot, diis_nd, dts_ready,
                  reduce=None):
        from nbed_amd.scf.diis import diis_coefficients

        self._count("huz_cycle")
        mode = int(tracked)
        n = h.n
        # ---- nbx_huz_cycle_jk, the all-reduce, nbx_fock_uhf
        h.jk = self.jk_sym(h.eri, dm_in, h.p0, h.p1)
        if reduce is not None:
            reduce(h.jk)
        fock, vhf = self.fock_uhf(h.hv, None, h.jk)
        fds = self._np(fock) @ self._np(h.ds)
        hz = -(fds + np.swapaxes(fds, -1, -2))
        fock2 = self._np(fock) + hz
        # ---- DIIS: pyscf.lib.diis.DIIS.update with the ring bookkeeping done by the caller
        f_use = fock2
        if diis_mode == 1:
            h.xprev = fock2.reshape(-1).copy()
        elif diis_mode == 2:
            flat = fock2.reshape(-1)
            h.xs[diis_slot] = flat
            h.es[diis_slot] = flat - h.xprev
            row = h.es[:diis_nd] @ h.es[diis_slot]
            h.H[diis_slot + 1, 1: diis_nd + 1] = row
            h.H[1: diis_nd + 1, diis_slot + 1] = row
            coef = diis_coefficients(h.H[: diis_nd + 1, : diis_nd + 1])
            h.xprev = coef[1:] @ h.xs[:diis_nd]
            f_use = h.xprev.reshape(2, n, n)
        xn = self._np(h.x)
        if mode == 2:  # purified: no orbitals, X F X left in out["v"]
            fo = xn @ f_use @ xn
            out["v"].copy_(torch.from_numpy(fo))
            p, st = self.purify(self.asarray(fo), h.nelec, refine_iters)
            dm = xn @ self._np(p) @ xn
            status = self._np(st).astype(np.int64)
        else:
            if mode == 1:
                w, c = self.geig_refine(self.asarray(f_use), h.s_b, c_in, refine_iters)
                status = np.array([1001, 1001])
            else:
                w, v = OracleBackend.eigh(self, self.asarray(xn @ f_use @ xn))
                out["v"].copy_(v)
                c = self.asarray(xn @ self._np(v))
                status = np.array([1, 1] if c_in is None else [1001, 1001])
            out["w"].copy_(w)
            out["c"].copy_(c)
            dm = self._np(self.density_occ(c, h.nelec))
        out["dm"].copy_(torch.from_numpy(np.ascontiguousarray(dm)))
        out["hz"].copy_(torch.from_numpy(hz))
        vals = self.huz_cycle_scalars(h.hv, None, vhf, out["hz"], out["dm"], dm_in)
        return _Done(vals, status)

    # ---- the mu-shift cycle (csrc/scf_cycle.hip: nbx_mu_cycle / _solve / _fock / _fock_post), step for step
    def mu_cycle_state(self, nao, nelec, packed, h1e, s_b, x, eri=None, p0=0, p1=None):
        h = self.huz_cycle_state(nao, nelec, packed, h1e, None, s_b, x, None, diis_space=8, eri=eri, p0=p0, p1=p1)
        n = h.n
        for out in h.sets:
            out["fock"], out["vhf"] = out.pop("hz"), self.empty((2, n, n))
        return h

    def _mu_fock_tail(self, h, dm, dm_old, c, out, reduce):
        h.jk = self.jk_sym(h.eri, dm, h.p0, h.p1)
        if reduce is not None:
            reduce(h.jk)
        fock, vhf = self.fock_uhf(h.hv, None, h.jk)
        out["fock"].copy_(fock)
        out["vhf"].copy_(vhf)
        gsq = None
        if c is not None:
            cn, fn = self._np(c), self._np(fock)
            gsq = []
            for x in range(2):
                fmo = cn[x].T @ fn[x] @ cn[x]
                k = h.nelec[x]
                gsq.append(float(np.sum(fmo[k:, :k] ** 2)))
        vals = self.huz_cycle_scalars(h.hv, None, vhf, self.zeros((2, h.n, h.n)), dm, dm_old)
        return vals, gsq

    def mu_cycle_fock(self, h, dm, out, reduce=None):
        self._count("mu_cycle_fock")
        vals, _ = self._mu_fock_tail(h, dm, dm, None, out, reduce)
        return _Done(vals)

    def mu_cycle(self, h, dm_in, fock_in, c_in, out, tracked, refine_iters, diis_on, diis_slot, diis_nd,
                 want_grad=True, reduce=None):
        from nbed_amd.scf.diis import diis_coefficients

        self._count("mu_cycle")
        n = h.n
        f_use = self._np(fock_in)
        if diis_on:  # pyscf.scf.diis.CDIIS: error F D S - S D F, lib.diis ring position with the caller
            s = self._np(h.s_b)[0]
            sdf = s @ self._np(dm_in) @ f_use
            err = np.swapaxes(sdf, -1, -2) - sdf
            h.xs[diis_slot] = f_use.reshape(-1)
            h.es[diis_slot] = err.reshape(-1)
            row = h.es[:diis_nd] @ h.es[diis_slot]
            h.H[diis_slot + 1, 1: diis_nd + 1] = row
            h.H[1: diis_nd + 1, diis_slot + 1] = row
            coef = diis_coefficients(h.H[: diis_nd + 1, : diis_nd + 1])
            f_use = (coef[1:] @ h.xs[:diis_nd]).reshape(2, n, n)
        xn = self._np(h.x)
        if tracked:
            w, c = self.geig_refine(self.asarray(f_use), h.s_b, c_in, refine_iters)
            status = np.array([1001, 1001])
        else:
            w, v = OracleBackend.eigh(self, self.asarray(xn @ f_use @ xn))
            out["v"].copy_(v)
            c = self.asarray(xn @ self._np(v))
            status = np.array([1, 1] if c_in is None else [1001, 1001])
        out["w"].copy_(w)
        out["c"].copy_(c)
        out["dm"].copy_(self.density_occ(c, h.nelec))
        vals, gsq = self._mu_fock_tail(h, out["dm"], dm_in, out["c"] if want_grad else None, out, reduce)
        return _Done(vals, status, gsq)
