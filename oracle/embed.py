"""Oracle: driver-level arithmetic of the embedding path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows nbed/driver.py:
``_env_projector`` (:433-449), ``_mu_embed`` (:500-538), ``_huzinaga_embed``
(:540-632), ``_delete_spin_environment`` (:715-791), ``post_embed`` (:925-1041).
"""

from __future__ import annotations

import numpy as np

from .hamiltonian import build as build_hamiltonian
from .huzinaga import energy_elec, huzinaga_scf


def env_projector(s_mat, dm_enviro):
    """P_sigma = S D_env,sigma S (driver.py:433-449)."""
    if dm_enviro.ndim == 2:
        return s_mat @ dm_enviro @ s_mat
    return np.array([s_mat @ dm_enviro[0] @ s_mat, s_mat @ dm_enviro[1] @ s_mat])


def mu_v_emb(mu, projector, embedding_potential):
    """v_emb = mu * P + V_emb (driver.py:518)."""
    return (mu * projector) + embedding_potential


def delete_spin_environment(projector_type, n_env_mo, mo_coeff, mo_energy, mo_occ, env_proj):
    """driver.py:715-791.  ``projector_type`` is "huzinaga" or "mu"."""
    if projector_type == "huzinaga":
        overlap = np.einsum("ij, ki -> i", mo_coeff.swapaxes(-1, -2), env_proj @ mo_coeff)
        overlap_by_size = overlap.argsort()[::-1]
        frozen = list(overlap_by_size[:n_env_mo])
    elif projector_type == "mu":
        shift = mo_coeff.shape[-1] - n_env_mo
        frozen = [i for i in range(shift, mo_coeff.shape[-1])]
    else:
        raise ValueError(projector_type)
    keep = [i for i in range(mo_coeff.shape[-1]) if i not in frozen]
    return mo_coeff[:, keep], mo_energy[keep], mo_occ[keep]


def delete_environment(projector_type, mo_coeff, mo_energy, mo_occ, enviro_mo_inds, env_proj):
    """driver.py:634-713 (unrestricted branch)."""
    n_env = len(set(enviro_mo_inds[0]).union(enviro_mo_inds[1]))
    a = delete_spin_environment(projector_type, n_env, mo_coeff[0], mo_energy[0], mo_occ[0], env_proj[0])
    b = delete_spin_environment(projector_type, n_env, mo_coeff[1], mo_energy[1], mo_occ[1], env_proj[1])
    return np.array([a[0], b[0]]), np.array([a[1], b[1]]), np.array([a[2], b[2]])


def mu_embed(scf, s_mat, dm_enviro, embedding_potential, mu):
    """driver.py:500-538: patch hcore/energy_elec and run the SCF kernel."""
    v_emb = mu_v_emb(mu, env_projector(s_mat, dm_enviro), embedding_potential)
    hcore_std = scf.get_hcore
    scf.energy_elec = lambda *args: energy_elec(scf, *args)
    scf.get_hcore = lambda *args: hcore_std(*args) + v_emb
    scf.kernel()
    return scf, v_emb


def huzinaga_embed(scf, embedding_potential, dm_enviro, dm_initial_guess=None, use_DIIS=True):
    """driver.py:540-632 (no localised virtuals: PAO is disabled, :819-820)."""
    c, e, dm, huz, conv = huzinaga_scf(
        scf, embedding_potential, dm_enviro, dm_conv_tol=1e-6,
        dm_initial_guess=dm_initial_guess, use_DIIS=use_DIIS,
    )
    hcore_std = scf.get_hcore()
    v_emb = huz + embedding_potential
    scf.get_hcore = lambda *args: hcore_std + v_emb
    scf.energy_elec = lambda *args: energy_elec(scf, *args)
    scf.mo_occ = scf.get_occ(e, c)
    scf.mo_coeff = c
    scf.mo_energy = e
    scf.e_tot = scf.energy_tot(dm=dm)
    scf.converged = conv
    return scf, v_emb


def post_embed(scf, v_emb, projector_type, dm_active, enviro_mo_inds, env_proj,
               e_env, two_e_cross, e_nuc, eri=None):
    """driver.py:925-1041 with virtual localisation disabled and no CCSD/FCI."""
    result = {"v_emb": v_emb, "mo_energies_emb_pre_del": scf.mo_energy}
    scf = scf.copy()
    scf.mo_coeff, scf.mo_energy, scf.mo_occ = delete_environment(
        projector_type, scf.mo_coeff, scf.mo_energy, scf.mo_occ, enviro_mo_inds, env_proj
    )
    result["scf"] = scf
    result["mo_energies_emb_post_del"] = scf.mo_energy
    result["correction"] = np.einsum("ij,ij", v_emb[0], dm_active[0])
    result["beta_correction"] = np.einsum("ij,ij", v_emb[1], dm_active[1])
    result["e_rhf"] = (
        scf.e_tot + e_env + two_e_cross - result["correction"] - result["beta_correction"]
    )
    result["classical_energy"] = (
        e_env + two_e_cross + e_nuc - result["correction"] - result["beta_correction"]
    )
    result["hf_emb"] = scf.e_tot - e_nuc
    if eri is not None:
        result["second_quantised"] = build_hamiltonian(
            scf.mo_coeff, scf.get_hcore(), eri, result["classical_energy"]
        )
    return result
