"""Exact diagonalisation of a small second-quantised Hamiltonian (consumer of the path).

The reference computes the embedded FCI energy with PySCF's ``fci.FCI`` on the embedded SCF object
(nbed/driver.py:1044-1102) and checks that the Hamiltonian ``HamiltonianBuilder.build()`` returns has the
same ground state (tests/test_builder.py:55-120).  For active spaces of a few orbitals -- the
reference's own examples: 5-6 spatial orbitals -- the same number follows from diagonalising

    H = constant + sum_pq h1[p,q] a+_p a_q + sum_pqrs h2[p,q,r,s] a+_p a+_q a_r a_s

(the ``(constant, h1, h2)`` of ``build()``, spin orbitals interleaved alpha/beta, the 1/2 already in
h2) in the determinants with the embedded molecule's (n_alpha, n_beta).  Host code, dense, brute
force: meant for <= 16 spin orbitals.
"""

from __future__ import annotations

import itertools

import numpy as np

MAX_SPIN_ORBITALS = 16


class FCIResult:
    """Duck-typed stand-in for the solver object the reference reads ``.e_tot`` from."""

    def __init__(self, e_tot, energies, ci, dets, converged=True):
        self.e_tot = float(e_tot)
        self.energies = energies
        self.ci = ci
        self.determinants = dets
        self.converged = converged


def _apply(det: int, creators, annihilators):
    """a+_{c0} a+_{c1} ... a_{a0} a_{a1} ... applied to the bit string ``det`` (rightmost acts first).
    Returns (sign, new_det) or (0, 0)."""
    sign = 1
    for a in reversed(annihilators):
        if not (det >> a) & 1:
            return 0, 0
        if bin(det & ((1 << a) - 1)).count("1") & 1:
            sign = -sign
        det &= ~(1 << a)
    for c in reversed(creators):
        if (det >> c) & 1:
            return 0, 0
        if bin(det & ((1 << c) - 1)).count("1") & 1:
            sign = -sign
        det |= 1 << c
    return sign, det


def ground_state(constant: float, h1: np.ndarray, h2: np.ndarray, nelec: tuple[int, int], nroots: int = 1) -> FCIResult:
    """Lowest eigenpair(s) of the Hamiltonian in the (n_alpha, n_beta) sector; alpha spin orbitals are
    the even indices (nbed/ham_builder.py:180-210)."""
    nq = h1.shape[0]
    if nq > MAX_SPIN_ORBITALS:
        raise ValueError(f"exact diagonalisation is limited to {MAX_SPIN_ORBITALS} spin orbitals (got {nq})")
    n = nq // 2
    na, nb = int(nelec[0]), int(nelec[1])
    dets = []
    for occ_a in itertools.combinations(range(n), na):
        for occ_b in itertools.combinations(range(n), nb):
            d = 0
            for p in occ_a:
                d |= 1 << (2 * p)
            for p in occ_b:
                d |= 1 << (2 * p + 1)
            dets.append(d)
    index = {d: i for i, d in enumerate(dets)}
    dim = len(dets)
    ham = np.zeros((dim, dim))
    one = [(p, q, h1[p, q]) for p, q in zip(*np.nonzero(h1))]
    two = [(p, q, r, s, h2[p, q, r, s]) for p, q, r, s in zip(*np.nonzero(h2))]
    for j, d in enumerate(dets):
        for p, q, v in one:
            sg, nd = _apply(d, (p,), (q,))
            if sg:
                ham[index[nd], j] += sg * v
        for p, q, r, s, v in two:
            sg, nd = _apply(d, (p, q), (r, s))
            if sg:
                i = index.get(nd)
                if i is not None:
                    ham[i, j] += sg * v
    ham = 0.5 * (ham + ham.T)
    w, c = np.linalg.eigh(ham)
    return FCIResult(w[0] + constant, w[:nroots] + constant, c[:, :nroots], dets)
