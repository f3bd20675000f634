"""Oracle: active-space Hamiltonian (one-body, four-index transform, spin-orbital scatter).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows nbed/ham_builder.py:53-254.
``ao2mo.kernel`` + ``ao2mo.restore(1, ...)`` (PySCF, not in the reference
repository) are restated through their defining identity
(ij|kl) = sum_pqrs C1_pi C2_qj C3_rk C4_sl (pq|rs).
"""

from __future__ import annotations

import numpy as np

EQ_TOLERANCE = 1e-8  # openfermion.config.EQ_TOLERANCE (openfermion 1.7.1), ham_builder.py:8


def ao2mo_full(eri, c1, c2, c3, c4):
    """Dense (n1,n2,n3,n4) chemist-order MO integrals by four quarter transforms."""
    x = np.tensordot(c1.T, eri, axes=(1, 0))  # i q r s
    x = np.tensordot(c2.T, x, axes=(1, 1)).transpose(1, 0, 2, 3)  # i j r s
    x = np.tensordot(x, c3, axes=(2, 0)).transpose(0, 1, 3, 2)  # i j k s
    x = np.tensordot(x, c4, axes=(3, 0))  # i j k l
    return x


def one_body_integrals(mo_coeff, hcore):
    """ham_builder.py:53-96 (unrestricted branch; 2-D hcore is duplicated)."""
    if hcore.ndim == 2:
        hcore = np.array([hcore, hcore])
    ha = mo_coeff[0].T @ hcore[0] @ mo_coeff[0]
    hb = mo_coeff[1].T @ hcore[1] @ mo_coeff[1]
    return np.array([ha, hb])


def two_body_integrals(mo_coeff, eri):
    """ham_builder.py:98-156: blocks aaaa,bbbb,aabb,bbaa in physicist order.

    T[p,q,r,s] = (p s | q r) i.e. ``eri_mo.transpose(0, 2, 3, 1)`` (:133).
    """
    ca, cb = mo_coeff[0], mo_coeff[1]
    if ca.shape[1] != cb.shape[1]:
        raise ValueError("Must localize the same number of alpha and beta orbitals.")
    spin_options = [(ca, ca, ca, ca), (cb, cb, cb, cb), (ca, ca, cb, cb), (cb, cb, ca, ca)]
    out = []
    for cs in spin_options:
        mo = ao2mo_full(eri, *cs)
        out.append(np.asarray(mo.transpose(0, 2, 3, 1), order="C"))
    return np.stack(out, axis=0)


def spinorb_from_spatial(one_body, two_body, tol=EQ_TOLERANCE):
    """ham_builder.py:158-216, vectorised (same element placement, same truncation)."""
    n = one_body.shape[-1]
    nq = 2 * n
    h1 = np.zeros((nq, nq))
    h2 = np.zeros((nq, nq, nq, nq))
    h1[0::2, 0::2] = one_body[0]
    h1[1::2, 1::2] = one_body[1]
    h2[0::2, 0::2, 0::2, 0::2] = two_body[0]
    h2[1::2, 1::2, 1::2, 1::2] = two_body[1]
    h2[0::2, 1::2, 1::2, 0::2] = two_body[2]
    h2[1::2, 0::2, 0::2, 1::2] = two_body[3]
    h1[np.absolute(h1) < tol] = 0.0
    h2[np.absolute(h2) < tol] = 0.0
    return h1, h2


def build(mo_coeff, hcore, eri, constant_e_shift=0.0):
    """HamiltonianBuilder.build() (ham_builder.py:218-254)."""
    ob = one_body_integrals(mo_coeff, hcore)
    tb = two_body_integrals(mo_coeff, eri)
    h1, h2 = spinorb_from_spatial(ob, tb)
    return constant_e_shift, h1, 0.5 * h2
