"""DIIS extrapolation with device-resident vectors.

``DIIS`` mirrors ``pyscf.lib.diis.DIIS`` as the reference uses it
(nbed/scf/huzinaga_scf.py:130,164: ``adiis.update(fock)`` with no explicit error vector):
space 6, min_space 1, error = x - x_prev_returned, first call only stores x.  On the HIP
backend the whole step -- error vector, Pulay row, the (<= 7 x 7) solve with PySCF's rule of
dropping modes whose |eigenvalue| < 1e-14 when there are any, and the extrapolation -- is one
``nbx_diis_update`` call that leaves nothing for the host to wait on.  A backend without it
(the tests' checker backend) gets the vectors' dot products (``dots``), solves the system on
the host and calls ``lincomb``.

``CDIIS`` is PySCF's SCF default (``scf.diis.CDIIS``: error S D F - F D S, space 8), used
by the mu-shift path's ``kernel()`` (nbed/driver.py:533).
"""

from __future__ import annotations

import numpy as np
import scipy.linalg


def diis_coefficients(h: np.ndarray) -> np.ndarray:
    """Solve H c = (1, 0, ...) as pyscf.lib.diis.DIIS.extrapolate does."""
    g = np.zeros(h.shape[0])
    g[0] = 1
    w, v = scipy.linalg.eigh(h)
    if np.any(abs(w) < 1e-14):
        idx = abs(w) > 1e-14
        return np.dot(v[:, idx] * (1.0 / w[idx]), np.dot(v[:, idx].T.conj(), g))
    return np.linalg.solve(h, g)


class DIIS:
    def __init__(self, backend, space: int = 6, min_space: int = 1):
        self.be = backend
        self.space = space
        self.min_space = min_space
        self._head = 0
        self._nd = 0
        self._xprev = None
        self._xs = None  # (space, n) device
        self._es = None
        self._H = np.zeros((space + 1, space + 1))
        self._H[0, 1:] = self._H[1:, 0] = 1
        self._Hd = None  # device copy of H (HIP backend)
        self._coef = None

    def get_num_vec(self) -> int:
        return self._nd

    def reserve(self, n: int) -> None:
        """Allocate the ring for vectors of n elements now (and upload the empty Pulay matrix):
        done lazily, the upload would make the first extrapolating cycle wait for the queue to drain."""
        be = self.be
        if self._xs is None:
            self._xs = be.empty((self.space, n))
            self._es = be.empty((self.space, n))
        if hasattr(be, "diis_update") and self.min_space <= 1 and self._Hd is None:
            self._Hd = be.asarray(self._H)
            self._coef = be.diis_coef_buffer(self.space)

    def update(self, x):
        be = self.be
        flat = x.reshape(-1)
        if self._xprev is None:
            # no error vector yet: remember x as the "previous returned" vector
            self._xprev = be.copy(flat)
            return x
        if self._xs is None:
            self.reserve(flat.numel())
        if self._head >= self.space:
            self._head = 0
        slot = self._head
        if hasattr(be, "diis_update") and self.min_space <= 1:
            if self._Hd is None:
                self._Hd = be.asarray(self._H)
                self._coef = be.diis_coef_buffer(self.space)
            self._head += 1
            self._nd = min(self._nd + 1, self.space)
            be.diis_update(self.space, slot, self._nd, flat, self._xprev, self._xs, self._es, self._Hd, self._coef)
            # xprev now holds the extrapolated vector (valid until the next update overwrites it)
            return self._xprev.reshape(x.shape)
        be.axpby(1.0, flat, 0.0, self._xs[slot])
        be.axpby(1.0, flat, 0.0, self._es[slot])
        be.axpby(-1.0, self._xprev, 1.0, self._es[slot])  # e = x - x_prev
        self._head += 1
        self._nd = min(self._nd + 1, self.space)
        nd = self._nd
        if nd < self.min_space:
            return x
        row = be.dots(self._es[slot], self._es[:nd])
        for i in range(nd):
            self._H[self._head, i + 1] = row[i]
            self._H[i + 1, self._head] = row[i]
        c = diis_coefficients(self._H[: nd + 1, : nd + 1])
        xnew = be.lincomb(c[1:], self._xs[:nd])
        self._xprev = xnew
        return xnew.reshape(x.shape)


class CDIIS:
    def __init__(self, backend, s_d, space: int = 8):
        self.be = backend
        self.s_d = s_d
        self.space = space
        self._fs = []
        self._es = []
        self._dev = None  # device-resident ring (HIP backend)
        self._head = 0
        self._nd = 0

    def update(self, dm_d, fock_d):
        be = self.be
        sdf = be.gemm(be.gemm(self.s_d, dm_d), fock_d)  # S D F  (batched over spin)
        err = be.transpose(sdf)
        be.axpby(-1.0, sdf, 1.0, err)  # (SDF)^T - SDF = FDS - SDF
        if hasattr(be, "diis_update_err"):
            # one nbx_diis_update_err call: ring slot, Pulay row, solve and extrapolation on the device
            flat, eflat = fock_d.reshape(-1), err.reshape(-1)
            if self._dev is None:
                h = np.zeros((self.space + 1, self.space + 1))
                h[0, 1:] = h[1:, 0] = 1
                self._dev = {
                    "xs": be.empty((self.space, flat.numel())), "es": be.empty((self.space, flat.numel())),
                    "h": be.asarray(h), "coef": be.diis_coef_buffer(self.space), "out": be.empty(flat.numel()),
                }
            slot = self._head % self.space
            self._head += 1
            self._nd = min(self._nd + 1, self.space)
            d = self._dev
            be.diis_update_err(self.space, slot, self._nd, flat, eflat, d["out"], d["xs"], d["es"], d["h"], d["coef"])
            return d["out"].reshape(fock_d.shape)  # valid until the next update
        self._fs.append(be.copy(fock_d).reshape(-1))
        self._es.append(err.reshape(-1))
        if len(self._fs) > self.space:
            self._fs.pop(0)
            self._es.pop(0)
        nd = len(self._fs)
        h = np.zeros((nd + 1, nd + 1))
        h[0, 1:] = h[1:, 0] = 1
        estack = be.torch.stack(self._es) if hasattr(be, "torch") else np.stack(self._es)
        for i in range(nd):
            row = be.dots(self._es[i], estack[: i + 1])
            for j in range(i + 1):
                h[i + 1, j + 1] = h[j + 1, i + 1] = row[j]
        c = diis_coefficients(h)
        fstack = be.torch.stack(self._fs) if hasattr(be, "torch") else np.stack(self._fs)
        return be.lincomb(c[1:], fstack).reshape(fock_d.shape)
