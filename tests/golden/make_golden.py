"""Generate golden input/output vectors by running the REFERENCE's own functions.

Run in the build container only (the reference does not travel):

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

How the reference is imported
-----------------------------
``import nbed`` needs ``pyscf`` and ``openfermion``, which are not installed and
cannot be fetched (no network).  The path's arithmetic that lives in the
reference's OWN files is pure numpy/scipy, so this script registers inert,
name-only stand-ins for those two packages in ``sys.modules`` (empty classes
used solely by ``isinstance`` checks, and the constant EQ_TOLERANCE = 1e-8) and
then imports /root/reference/nbed unchanged.  No reference source is copied:
the functions are called in place and only their numeric inputs/outputs are
stored.

Two stand-ins carry arithmetic because the reference calls into PySCF there;
fixtures depending on them are tagged in ``MANIFEST.json``:

* ``pyscf.lib.diis.DIIS``   -> oracle.pyscf_like.DIIS   (used by huzinaga_scf with use_DIIS=True)
* ``pyscf.ao2mo.kernel/restore`` -> dense einsum identity (used by HamiltonianBuilder._two_body_integrals)

The duck-typed SCF object handed to the reference (``get_ovlp/get_hcore/get_veff/
get_occ/make_rdm1``) is oracle.pyscf_like.ToyUHF/ToyRHF over synthetic dense
S, hcore, (pq|rs) from oracle.synth.
"""

from __future__ import annotations

import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.dont_write_bytecode = True

from oracle import synth  # noqa: E402
from oracle.hamiltonian import ao2mo_full  # noqa: E402
from oracle.pyscf_like import DIIS as OracleDIIS  # noqa: E402
from oracle.pyscf_like import ToyMol, ToyRHF, ToyUHF  # noqa: E402

REFERENCE = "/root/reference"


# --------------------------------------------------------------------------- stand-ins
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class StreamObject:
    pass


class StubRHF(StreamObject):
    pass


class StubUHF(StreamObject):
    pass


class StubRKS(StreamObject):
    pass


class StubUKS(StreamObject):
    pass


def _ao2mo_kernel(mol, coeffs):
    """Stand-in with arithmetic: dense four-index transform of mol._toy_eri."""
    if isinstance(coeffs, np.ndarray) and coeffs.ndim == 2:
        coeffs = (coeffs,) * 4
    return ao2mo_full(mol._toy_eri, *coeffs)


def _ao2mo_restore(sym, eri, n):
    assert sym == 1
    return np.asarray(eri).reshape(n, n, n, n)


def install_standins():
    pyscf = _mod("pyscf")
    lib = _mod("pyscf.lib", StreamObject=StreamObject)
    pyscf.lib = lib
    lib.diis = _mod("pyscf.lib.diis", DIIS=OracleDIIS)
    lib.misc = _mod("pyscf.lib.misc")
    dft = _mod("pyscf.dft", RKS=StubRKS, UKS=StubUKS, KohnShamDFT=object, ROKS=object)
    pyscf.dft = dft
    dft.rks = _mod("pyscf.dft.rks", RKS=StubRKS)
    dft.uks = _mod("pyscf.dft.uks", UKS=StubUKS)
    scf = _mod("pyscf.scf", RHF=StubRHF, UHF=StubUHF, ROHF=object)
    pyscf.scf = scf
    scf.rhf = _mod("pyscf.scf.rhf", RHF=StubRHF)
    scf.uhf = _mod("pyscf.scf.uhf", UHF=StubUHF)
    scf.hf = _mod("pyscf.scf.hf", RHF=StubRHF)
    gto = _mod("pyscf.gto", Mole=object)
    pyscf.gto = gto
    gto.mole = _mod("pyscf.gto.mole", Mole=object)
    lo = _mod("pyscf.lo")
    pyscf.lo = lo
    lo.vvo = _mod("pyscf.lo.vvo")
    pyscf.ao2mo = _mod("pyscf.ao2mo", kernel=_ao2mo_kernel, restore=_ao2mo_restore)
    pyscf.cc = _mod("pyscf.cc", CCSD=object)
    pyscf.fci = _mod("pyscf.fci", FCI=object)
    pyscf.qmmm = _mod("pyscf.qmmm")
    of = _mod("openfermion")
    of.config = _mod("openfermion.config", EQ_TOLERANCE=1e-8)
    of.chem = _mod("openfermion.chem")
    of.chem.pubchem = _mod("openfermion.chem.pubchem", geometry_from_pubchem=None)


# SCF objects the reference accepts through its isinstance checks
class RefUHF(StubUHF, ToyUHF):
    pass


class RefRHF(StubRHF, ToyRHF):
    pass


class TaggedArray(np.ndarray):
    """ndarray carrying .ecoul / .exc, as PySCF's tagged Kohn-Sham veff does."""


class RefUKS(StubUKS, ToyUHF):
    """Toy unrestricted Kohn-Sham object for the reference's KS branch (huzinaga_scf.py:36-62,
    176-180): a hybrid whose only exchange-correlation is a fraction ``hyb`` of exact exchange,
    veff[x] = J - hyb K[x], ecoul = 1/2 tr(Dtot J), exc = -hyb/2 sum_x tr(D[x] K[x])."""

    hyb = 0.2

    def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0):
        dm = self.make_rdm1() if dm is None else np.asarray(dm)
        vj, vk = self.get_jk(mol, dm)
        v = (vj[0] + vj[1] - self.hyb * vk).view(TaggedArray)
        v.ecoul = 0.5 * float(np.einsum("ij,ji->", vj[0] + vj[1], dm[0] + dm[1]))
        v.exc = -0.5 * self.hyb * float(np.einsum("xij,xji->", vk, dm))
        return v


def canon_sign(c):
    """Flip each column so that its largest-|component| is positive (gauge fix)."""
    c = np.array(c, copy=True)
    idx = np.argmax(np.abs(c), axis=-2)
    sign = np.sign(np.take_along_axis(c, idx[..., None, :], axis=-2))
    sign[sign == 0] = 1
    return c * sign


def make_uhf(n, nocc, n_env, max_cycle, conv_tol, ao_slices=None):
    pr = synth.problem(n, nocc, n_env)
    eri = synth.eri_dense(n)
    mol = ToyMol(n, pr["nelec"], ao_slices=ao_slices, e_nuc=1.25)
    mol._toy_eri = eri
    mf = RefUHF(mol, pr["S"], pr["hcore"], eri)
    mf.max_cycle = max_cycle
    mf.conv_tol = conv_tol
    return mf, pr, eri


def main():
    install_standins()
    tmp = tempfile.mkdtemp(prefix="nbed_golden_")
    os.chdir(tmp)  # the reference writes ./.nbed.log at import (nbed/utils.py:32)
    sys.path.insert(0, REFERENCE)
    import nbed  # noqa: F401
    from nbed.config import ProjectorTypes
    from nbed.driver import NbedDriver
    from nbed.ham_builder import HamiltonianBuilder
    from nbed.localizers import ConcentricLocalizer, SPADELocalizer
    from nbed.localizers.system import LocalizedSystem
    from nbed.scf import energy_elec
    from nbed.scf.huzinaga_scf import get_huzinaga_operator, huzinaga_scf

    manifest = {}

    def save(name, note, standin_arithmetic, **arrays):
        np.savez_compressed(HERE / f"{name}.npz", **arrays)
        manifest[name] = {"reference": note, "standin_arithmetic": standin_arithmetic,
                          "keys": sorted(arrays)}

    # ------------------------------------------------------------------ 1. Huzinaga operator
    n = 9
    f3 = np.stack([synth.sym_matrix(synth.STREAM_MISC, n), synth.sym_matrix(synth.STREAM_MISC + 1, n)])
    ds = np.stack([synth.general_matrix(synth.STREAM_MISC + 2, n, n),
                   synth.general_matrix(synth.STREAM_MISC + 3, n, n)]) * 0.3
    dv = np.stack([synth.general_matrix(synth.STREAM_MISC + 4, n, n),
                   synth.general_matrix(synth.STREAM_MISC + 5, n, n)]) * 0.2
    save("huzinaga_operator", "nbed/scf/huzinaga_scf.py:65-90 get_huzinaga_operator", [],
         fock=f3, dm_occ_S=ds, dm_virt_S=dv,
         out_3d=get_huzinaga_operator(f3, ds, dv),
         out_3d_novirt=get_huzinaga_operator(f3, ds, np.zeros_like(ds)),
         out_2d=get_huzinaga_operator(f3[0], ds[0], dv[0]))

    # ------------------------------------------------------------------ 2. Huzinaga SCF loop
    for tag, n, nocc, n_env, diis, cyc in [
        ("uhf_n12_nodiis", 12, (4, 4), 1, False, 8),
        ("uhf_n12_diis", 12, (4, 4), 1, True, 60),
        ("uhf_n24_diis_open", 24, (6, 5), 2, True, 60),
        ("uhf_n24_nodiis_open", 24, (6, 5), 2, False, 60),
    ]:
        mf, pr, eri = make_uhf(n, nocc, n_env, cyc, 1e-9)
        c, e, d, hz, conv = huzinaga_scf(mf, pr["V_emb"], pr["D_env"], use_DIIS=diis)
        save(f"huzinaga_scf_{tag}", "nbed/scf/huzinaga_scf.py:93-206 huzinaga_scf (UHF branch)",
             ["pyscf.lib.diis.DIIS"] if diis else [],
             nao=n, nelec=np.array(pr["nelec"]), n_env=n_env, max_cycle=cyc, conv_tol=1e-9,
             use_DIIS=diis, S=pr["S"], hcore=pr["hcore"], V_emb=pr["V_emb"], D_env=pr["D_env"],
             mo_coeff=c, mo_coeff_canon=canon_sign(c), mo_energy=e, dm=d, huz_op=hz, conv=conv)

    # restricted (2-D) flavour, as tests/test_scf.py:78-99 exercises
    n = 12
    pr = synth.problem(n, (4, 4), 1)
    eri = synth.eri_dense(n)
    mol = ToyMol(n, pr["nelec"])
    rhf = RefRHF(mol, pr["S"], pr["hcore"], eri)
    rhf.max_cycle, rhf.conv_tol = 60, 1e-9
    c, e, d, hz, conv = huzinaga_scf(rhf, pr["V_emb"][0], 2 * pr["D_env"][0], use_DIIS=True)
    save("huzinaga_scf_rhf_n12_diis", "nbed/scf/huzinaga_scf.py:93-206 huzinaga_scf (RHF, 2-D branch)",
         ["pyscf.lib.diis.DIIS"],
         nao=n, nelec=np.array(pr["nelec"]), max_cycle=60, conv_tol=1e-9, S=pr["S"], hcore=pr["hcore"],
         V_emb=pr["V_emb"][0], D_env=2 * pr["D_env"][0], mo_coeff_canon=canon_sign(c), mo_energy=e,
         dm=d, huz_op=hz, conv=conv)

    # Kohn-Sham branch (:176-180 -> calculate_ks_energy :36-62: a second get_veff per cycle)
    from nbed.scf.huzinaga_scf import calculate_ks_energy

    for tag, n, nocc, n_env, diis, cyc in [("uks_n12_diis", 12, (4, 4), 1, True, 60),
                                           ("uks_n24_nodiis_open", 24, (6, 5), 2, False, 80)]:
        pr = synth.problem(n, nocc, n_env)
        eri = synth.eri_dense(n)
        ks = RefUKS(ToyMol(n, pr["nelec"], e_nuc=1.25), pr["S"], pr["hcore"], eri)
        ks.max_cycle, ks.conv_tol = cyc, 1e-9
        c, e, d, hz, conv = huzinaga_scf(ks, pr["V_emb"], pr["D_env"], use_DIIS=diis)
        e_ks = calculate_ks_energy(ks, pr["V_emb"], d, hz)
        save(f"huzinaga_scf_{tag}", "nbed/scf/huzinaga_scf.py:93-206 huzinaga_scf (UKS branch :176-180) + "
             ":36-62 calculate_ks_energy; toy hybrid functional = 0.2 exact exchange",
             ["pyscf.lib.diis.DIIS"] if diis else [],
             nao=n, nelec=np.array(pr["nelec"]), n_env=n_env, max_cycle=cyc, conv_tol=1e-9, hyb=RefUKS.hyb,
             use_DIIS=diis, S=pr["S"], hcore=pr["hcore"], V_emb=pr["V_emb"], D_env=pr["D_env"],
             mo_coeff_canon=canon_sign(c), mo_energy=e, dm=d, huz_op=hz, conv=conv, e_ks=np.asarray(e_ks))

    # ------------------------------------------------------------------ 3. energy_elec
    mf, pr, eri = make_uhf(12, (4, 4), 1, 10, 1e-9)
    dm = synth.problem(12, (5, 4), 0)  # any symmetric density-like input
    _, cfull = synth.lowdin_orthonormal(pr["S"], pr["hcore"])
    dm = np.stack([cfull[:, :5] @ cfull[:, :5].T, cfull[:, 1:5] @ cfull[:, 1:5].T])
    h3 = pr["hcore"] + pr["V_emb"]
    ee = energy_elec(mf, dm, h3, None)
    save("energy_elec", "nbed/scf/embedded_hcore_funcs.py:11-46 energy_elec", [],
         nao=12, S=pr["S"], hcore3=h3, dm=dm, e_elec=ee[0], e_coul=ee[1],
         e1=mf.scf_summary["e1"], e2=mf.scf_summary["e2"])

    # ------------------------------------------------------------------ 4. SPADE
    def spade_case(tag, n, nocc, n_act_aos, overwrite=None, occ_override=None):
        pr = synth.problem(n, nocc, 0)
        h_b = pr["hcore"] + 0.05 * synth.sym_matrix(synth.STREAM_MISC + 7, n)
        _, ca = synth.lowdin_orthonormal(pr["S"], pr["hcore"])
        _, cb = synth.lowdin_orthonormal(pr["S"], h_b)
        mo_coeff = np.stack([ca, cb])
        mo_occ = np.zeros((2, n))
        mo_occ[0, : nocc[0]] = 1
        mo_occ[1, : nocc[1]] = 1
        mol = ToyMol(n, nocc, ao_slices=[[0, 1, 0, n_act_aos], [1, 2, n_act_aos, n]])
        mf = RefUHF(mol, pr["S"], pr["hcore"], None)
        mf.mo_coeff, mf.mo_occ = mo_coeff, mo_occ
        loc = SPADELocalizer(mf, 1, n_mo_overwrite=overwrite)
        ls = loc.localize()
        save(f"spade_{tag}", "nbed/localizers/occupied/spade.py:57-147 + base.py:64-140 localize()", [],
             nao=n, n_act_aos=n_act_aos, S=pr["S"], mo_coeff=mo_coeff, mo_occ=mo_occ,
             overwrite=np.array([-1 if o is None else o for o in (overwrite or (None, None))]),
             active_mo_inds=np.asarray(ls.active_mo_inds), enviro_mo_inds=np.asarray(ls.enviro_mo_inds),
             c_active=ls.c_active, c_enviro=ls.c_enviro, c_loc_occ=ls.c_loc_occ,
             dm_active=ls.dm_active, dm_enviro=ls.dm_enviro, dm_loc_occ=ls.dm_loc_occ,
             sigma_a=loc.enviro_selection_condition[0], sigma_b=loc.enviro_selection_condition[1])

    spade_case("n16_closed", 16, (5, 5), 6)
    spade_case("n16_overwrite", 16, (5, 5), 6, overwrite=(2, 2))
    spade_case("n20_wide", 20, (8, 8), 5)  # fewer active AOs than occupied MOs
    # Open shell: alpha/beta index arrays have different lengths, and with numpy >= 1.24
    # (pyproject.toml:16 pins numpy > 2) ``np.array([alpha_inds, beta_inds])`` at
    # base.py:98-100 raises ValueError before the consistency re-run (:107-130) is reached.
    try:
        spade_case("n16_open", 16, (6, 5), 6)
        raised = ""
    except ValueError as err:
        raised = f"ValueError: {err}"
    save("spade_n16_open_raises", "base.py:98-100 ragged np.array on open-shell occupations", [],
         nocc=np.array([6, 5]), nao=16, n_act_aos=6, raised=raised)

    # restricted SPADE (2-D), base.py:75-85
    n, nocc, n_act_aos = 14, 4, 5
    pr = synth.problem(n, (nocc, nocc), 0)
    _, c = synth.lowdin_orthonormal(pr["S"], pr["hcore"])
    occ = np.zeros(n)
    occ[:nocc] = 2
    mol = ToyMol(n, (nocc, nocc), ao_slices=[[0, 1, 0, n_act_aos], [1, 2, n_act_aos, n]])
    mf = RefRHF(mol, pr["S"], pr["hcore"], None)
    mf.mo_coeff, mf.mo_occ = c, occ
    loc = SPADELocalizer(mf, 1)
    ls = loc.localize()
    save("spade_n14_restricted", "spade.py:57-147 + base.py:75-85 (restricted)", [],
         nao=n, n_act_aos=n_act_aos, S=pr["S"], mo_coeff=c, mo_occ=occ,
         active_mo_inds=ls.active_mo_inds, enviro_mo_inds=ls.enviro_mo_inds,
         dm_active=ls.dm_active, dm_enviro=ls.dm_enviro, dm_loc_occ=ls.dm_loc_occ,
         sigma_a=loc.enviro_selection_condition[0])

    # ------------------------------------------------------------------ 5. env projector, mu v_emb, deletion
    n, nocc, n_env = 14, (5, 5), 2
    pr = synth.problem(n, nocc, n_env)
    drv = NbedDriver.__new__(NbedDriver)

    class _KS:
        def get_ovlp(self_inner):
            return pr["S"]

    drv.__dict__["_global_ks"] = _KS()
    c_env = np.stack([pr["C_env"], pr["C_env"]])
    _, cfull = synth.lowdin_orthonormal(pr["S"], pr["hcore"])
    c_act = np.stack([cfull[:, n_env:nocc[0]], cfull[:, n_env:nocc[1]]])
    c_loc = np.stack([cfull[:, : nocc[0]], cfull[:, : nocc[1]]])
    drv.localized_system = LocalizedSystem(
        np.array([np.arange(3), np.arange(3)]), np.array([np.arange(3, 5), np.arange(3, 5)]),
        c_act, c_env, c_loc)
    proj = drv._env_projector
    save("env_projector", "nbed/driver.py:433-449 _env_projector; :518 v_emb", [],
         S=pr["S"], dm_enviro=drv.localized_system.dm_enviro, projector=proj, mu=1e6,
         V_emb=pr["V_emb"], v_emb=(1e6 * proj) + pr["V_emb"])

    # deletion rules on a Huzinaga-converged toy SCF
    mf, pr, eri = make_uhf(n, nocc, n_env, 60, 1e-9)
    c, e, d, hz, conv = huzinaga_scf(mf, pr["V_emb"], pr["D_env"], use_DIIS=True)
    occ = mf.get_occ(e, c)
    drv.config = types.SimpleNamespace()
    outs = {}
    for ptype, key in [(ProjectorTypes.HUZ, "huz"), (ProjectorTypes.MU, "mu")]:
        for x in range(2):
            r = drv._delete_spin_environment(ptype, n_env, c[x], e[x], occ[x], proj[x])
            outs[f"{key}_coeff_{x}"], outs[f"{key}_energy_{x}"], outs[f"{key}_occ_{x}"] = r
    save("delete_environment", "nbed/driver.py:715-791 _delete_spin_environment", ["pyscf.lib.diis.DIIS"],
         n_env=n_env, mo_coeff=c, mo_energy=e, mo_occ=occ, projector=proj, **outs)

    # ------------------------------------------------------------------ 6. concentric localisation
    n, n_act = 18, 7
    pr = synth.problem(n, (4, 4), 0)
    _, cfull = synth.lowdin_orthonormal(pr["S"], pr["hcore"])
    n_mo = 15  # environment already deleted: (N, n) coefficients
    mo = cfull[:, :n_mo]
    occ = np.zeros(n_mo)
    occ[:4] = 1
    fock = pr["hcore"] + 0.1 * synth.sym_matrix(synth.STREAM_MISC + 9, n)
    cl = ConcentricLocalizer.__new__(ConcentricLocalizer)
    for max_shells in (4, 1):
        cl.max_shells = max_shells
        cl.projected_overlap = pr["S"][:n_act, :n_act]
        cl.overlap_two_basis = pr["S"][:n_act, :]
        cl.n_act_proj_aos = n_act
        out_c, shells, svals = cl._localize_virtual_spin(occ, mo, fock)
        arrays = {f"sigma_{i}": s for i, s in enumerate(svals)}
        save(f"concentric_n18_shells{max_shells}",
             "nbed/localizers/virtual/concentric.py:123-262 _localize_virtual_spin", [],
             S=pr["S"], n_act_proj_aos=n_act, max_shells=max_shells, occ=occ, mo_coeff=mo, fock=fock,
             out_coeff=out_c, shells=np.array(shells), n_sigma=len(svals), **arrays)

    # ------------------------------------------------------------------ 7. spin-orbital scatter + builder
    nmo = 4
    ob = np.stack([synth.sym_matrix(synth.STREAM_MISC + 10, nmo), synth.sym_matrix(synth.STREAM_MISC + 11, nmo)])
    tb = synth.val(synth.STREAM_MISC + 12, np.arange(4 * nmo**4)).reshape(4, nmo, nmo, nmo, nmo)
    tb = tb * (10.0 ** (-9 * (np.arange(4 * nmo**4).reshape(tb.shape) % 3 == 0)))  # exercise the 1e-8 cut
    hb = HamiltonianBuilder.__new__(HamiltonianBuilder)
    h1, h2 = hb._spinorb_from_spatial(ob, tb)
    save("spinorb_from_spatial", "nbed/ham_builder.py:158-216 _spinorb_from_spatial", [],
         one_body=ob, two_body=tb, h1=h1, h2=h2)

    mf, pr, eri = make_uhf(10, (4, 4), 1, 60, 1e-9)
    c, e, d, hz, conv = huzinaga_scf(mf, pr["V_emb"], pr["D_env"], use_DIIS=True)
    mf.mo_coeff = c[:, :, :7]
    mf.mo_occ = mf.get_occ(e, c)[:, :7]
    h3 = pr["hcore"] + hz + pr["V_emb"]
    mf.get_hcore = lambda *a: h3
    builder = HamiltonianBuilder(mf, constant_e_shift=-3.25)
    const, b1, b2 = builder.build()
    save("ham_build_n10", "nbed/ham_builder.py:218-254 build() (one-body :53-96, two-body :98-156)",
         ["pyscf.ao2mo.kernel", "pyscf.ao2mo.restore", "pyscf.lib.diis.DIIS"],
         nao=10, mo_coeff=mf.mo_coeff, hcore3=h3, const=const, one_body=builder._one_body_integrals,
         two_body=builder._two_body_integrals, h1=b1, h2=b2)

    # ------------------------------------------------------------------ 8. post_embed arithmetic (both projectors)
    from nbed.config import VirtualLocalizerTypes

    n, nocc, n_env = 12, (5, 5), 2
    mf, pr, eri = make_uhf(n, nocc, n_env, 80, 1e-10)
    _, cfull = synth.lowdin_orthonormal(pr["S"], pr["hcore"])
    c_env = np.stack([pr["C_env"], pr["C_env"]])
    c_act = np.stack([cfull[:, n_env:nocc[0]], cfull[:, n_env:nocc[1]]])
    c_loc = np.stack([cfull[:, : nocc[0]], cfull[:, : nocc[1]]])
    ls = LocalizedSystem(np.array([np.arange(3), np.arange(3)]),
                         np.array([np.arange(3, 5), np.arange(3, 5)]), c_act, c_env, c_loc)
    drv = NbedDriver.__new__(NbedDriver)

    class _KS2:
        def get_ovlp(self_inner):
            return pr["S"]

    drv.__dict__["_global_ks"] = _KS2()
    drv.localized_system = ls
    drv.config = types.SimpleNamespace(
        virtual_localization=VirtualLocalizerTypes.DISABLE, run_ccsd_emb=False, run_fci_emb=False,
        run_dft_in_dft=False, n_active_atoms=1, max_shells=4, mu_level_shift=1e6)
    drv.e_env, drv.two_e_cross, drv.e_nuc = -3.5, 0.75, 1.25
    emb_scf, v_emb = drv._huzinaga_embed(mf, pr["V_emb"], ls, None)
    res = drv.post_embed(emb_scf, v_emb, ProjectorTypes.HUZ)
    const, b1, b2 = res["second_quantised"]
    save("post_embed_huzinaga_n12",
         "nbed/driver.py:540-632 _huzinaga_embed + :925-1041 post_embed (CL disabled)",
         ["pyscf.ao2mo.kernel", "pyscf.ao2mo.restore", "pyscf.lib.diis.DIIS"],
         nao=n, nelec=np.array(pr["nelec"]), n_env=n_env, S=pr["S"], hcore=pr["hcore"], V_emb=pr["V_emb"],
         c_active=c_act, c_enviro=c_env, c_loc_occ=c_loc,
         e_env=drv.e_env, two_e_cross=drv.two_e_cross, e_nuc=drv.e_nuc, max_cycle=80, conv_tol=1e-10,
         v_emb=v_emb, e_tot=emb_scf.e_tot, converged=emb_scf.converged,
         mo_energies_pre=res["mo_energies_emb_pre_del"], mo_energies_post=res["mo_energies_emb_post_del"],
         mo_coeff_post_canon=canon_sign(res["scf"].mo_coeff), mo_occ_post=res["scf"].mo_occ,
         correction=res["correction"], beta_correction=res["beta_correction"], e_rhf=res["e_rhf"],
         classical_energy=res["classical_energy"], hf_emb=res["hf_emb"], const=const,
         h1_abs=np.abs(b1), h2_abs=np.abs(b2))

    # ------------------------------------------------------------------ 9. DFT-in-DFT (consumer of the path)
    from nbed.driver import dft_in_dft

    n, nocc, n_env = 12, (5, 5), 2
    pr = synth.problem(n, nocc, n_env)
    eri = synth.eri_dense(n)
    _, cfull = synth.lowdin_orthonormal(pr["S"], pr["hcore"])
    c_env = np.stack([pr["C_env"], pr["C_env"]])
    c_act = np.stack([cfull[:, n_env:nocc[0]], cfull[:, n_env:nocc[1]]])
    c_loc = np.stack([cfull[:, : nocc[0]], cfull[:, : nocc[1]]])
    ls = LocalizedSystem(np.array([np.arange(3), np.arange(3)]),
                         np.array([np.arange(3, 5), np.arange(3, 5)]), c_act, c_env, c_loc)
    for tag, ptype in (("huzinaga", ProjectorTypes.HUZ), ("mu", ProjectorTypes.MU)):
        drv = NbedDriver.__new__(NbedDriver)

        class _GlobalKS:
            xc = "toy-hybrid"

            def get_ovlp(self_inner):
                return pr["S"]

            def energy_nuc(self_inner):
                return 1.25

        drv.__dict__["_global_ks"] = _GlobalKS()
        drv.localized_system = ls
        drv.config = types.SimpleNamespace(mu_level_shift=1e6)
        drv.e_env, drv.two_e_cross, drv.e_nuc = -3.5, 0.75, 1.25
        drv.embedding_potential = pr["V_emb"]

        def _local_ks(xc, _pr=pr, _eri=eri):
            ks = RefUKS(ToyMol(n, _pr["nelec"], e_nuc=1.25), _pr["S"], _pr["hcore"], _eri)
            ks.max_cycle, ks.conv_tol, ks.xc = 80, 1e-10, xc
            return ks

        drv._init_local_ks = _local_ks
        if ptype is ProjectorTypes.MU:
            continue  # the mu path runs PySCF's kernel(): no executable reference here
        res = dft_in_dft(drv, ptype)
        save(f"dft_in_dft_{tag}_n12", "nbed/driver.py:1138-1231 dft_in_dft (Huzinaga projector; toy hybrid functional "
             "= 0.2 exact exchange; local KS object through _huzinaga_embed's KS branch)", ["pyscf.lib.diis.DIIS"],
             nao=n, nelec=np.array(pr["nelec"]), n_env=n_env, hyb=RefUKS.hyb, S=pr["S"], hcore=pr["hcore"],
             V_emb=pr["V_emb"], c_active=c_act, c_enviro=c_env, c_loc_occ=c_loc, e_env=drv.e_env,
             two_e_cross=drv.two_e_cross, e_nuc=drv.e_nuc, max_cycle=80, conv_tol=1e-10,
             v_emb_dft=res["v_emb_dft"], dft_correction=res["dft_correction"],
             dft_correction_beta=res["dft_correction_beta"], e_dft_in_dft=res["e_dft_in_dft"], emb_dft=res["emb_dft"],
             mo_energy_post=res["scf_dft"].mo_energy, mo_occ_post=res["scf_dft"].mo_occ,
             mo_coeff_post_canon=canon_sign(res["scf_dft"].mo_coeff), converged=res["scf_dft"].converged)

    # ------------------------------------------------------------------ 10. ACE of SPADE
    from nbed.localizers.ace import ACELocalizer

    n, nocc, n_act_aos = 16, (6, 6), 6
    path = []
    coeffs, occs = [], []
    for k in range(3):  # three "geometries": the same overlap, smoothly changing core Hamiltonians
        pr_k = synth.problem(n, nocc, 0)
        h_k = pr_k["hcore"] + 0.04 * k * synth.sym_matrix(synth.STREAM_MISC + 30, n)
        _, ca = synth.lowdin_orthonormal(pr_k["S"], h_k)
        _, cb = synth.lowdin_orthonormal(pr_k["S"], h_k + 0.002 * synth.sym_matrix(synth.STREAM_MISC + 7, n))
        mo_occ = np.zeros((2, n))
        mo_occ[:, : nocc[0]] = 1
        mf = RefUHF(ToyMol(n, nocc, ao_slices=[[0, 1, 0, n_act_aos], [1, 2, n_act_aos, n]]), pr_k["S"], h_k, None)
        mf.mo_coeff, mf.mo_occ = np.stack([ca, cb]), mo_occ
        path.append(mf)
        coeffs.append(mf.mo_coeff)
        occs.append(mo_occ)
    ace = ACELocalizer(path, 1)
    n_ab = ace.localize_path()
    sig_sets = [np.array([0.99, 0.97, 0.9, 0.35, 0.2, 0.05]), np.array([0.98, 0.95, 0.8, 0.4, 0.1, 0.02]),
                np.array([0.99, 0.9, 0.85, 0.3, 0.25, 0.01])]
    save("ace_of_spade_n16", "nbed/localizers/ace.py:54-131 ACELocalizer.localize_path / localize_spin", [],
         nao=n, n_act_aos=n_act_aos, S=path[0].get_ovlp(), mo_coeff=np.stack(coeffs), mo_occ=np.stack(occs),
         n_mo=np.array(n_ab), sigma_sets=np.stack(sig_sets), n_mo_from_sets=ACELocalizer.localize_spin(ace, sig_sets))

    # entries written by other generators (make_water_integrals.py) are kept
    mpath = HERE / "MANIFEST.json"
    if mpath.exists():
        for name, meta in json.loads(mpath.read_text()).items():
            if name not in manifest and (HERE / f"{name}.npz").exists():
                manifest[name] = meta
    mpath.write_text(json.dumps(manifest, indent=1, sort_keys=True) + "\n")
    print("wrote", len(manifest), "fixtures to", HERE)


if __name__ == "__main__":
    main()
