"""NbedDriver: orchestration of the projection-based embedding, GPU hot path underneath.

Drop-in for nbed/driver.py's ``NbedDriver``: ``NbedDriver(config)``, ``.embed()``,
``.post_embed()``, the attributes ``embed`` sets (``e_nuc, localized_system, e_act, e_env,
two_e_cross, embedding_potential, mu, huzinaga, embedded_scf, classical_energy``) and the
result-dict keys (``scf, v_emb, mo_energies_emb_pre_del, mo_energies_emb_post_del, correction,
beta_correction, cl, e_rhf, classical_energy, hf_emb, second_quantised``).

What runs where
---------------
* The embedded-SCF hot path -- projector S D_env S, mu-shifted or Huzinaga SCF (J/K, projector
  products, eigensolves), environment deletion scores, concentric localisation, correction
  traces, the active-space Hamiltonian -- runs on the GPU through ``nbed_amd``'s own classes.
* What the reference delegates to PySCF OUTSIDE that path -- building the molecule and its AO
  integrals, the global B3LYP Kohn-Sham calculation that produces the densities and the
  embedding potential, CCSD / FCI / DFT-in-DFT reference energies -- comes from a *provider*
  object.  ``PySCFProvider`` (default) uses a local PySCF installation for exactly those
  pieces; tests use a synthetic provider.  Without PySCF the default provider raises
  ``NbedDriverError`` naming what is missing: there is no silent fallback.

Reference quirks kept on purpose: the environment-selection score of
the Huzinaga path is ``colsum(C) * colsum(P C)`` (driver.py:746-756, an einsum that is NOT
diag(C^T P C)); the mu path deletes the LAST n_env orbitals by index (:758-766); n_env is the
size of the union of the alpha and beta environment index sets (:671-675); the ``savefile``
branch binds a boolean (:918) and is not reproduced.
"""

from __future__ import annotations

import logging
from functools import cached_property
from typing import Optional

import numpy as np

from .backend import get_backend
from .config import NbedConfig, OccupiedLocalizerTypes, ProjectorTypes, VirtualLocalizerTypes
from .exceptions import NbedDriverError
from .ham_builder import HamiltonianBuilder
from .localizers import ConcentricLocalizer, LocalizedSystem, SPADELocalizer
from .localizers.occupied.unsupported import BOYSLocalizer, IBOLocalizer, PMLocalizer
from .scf import energy_elec
from .scf.huzinaga_scf import huzinaga_scf

logger = logging.getLogger(__name__)


class PySCFProvider:
    """The out-of-path pieces from a local PySCF: molecule/integrals and the global UKS."""

    def __init__(self):
        try:
            import pyscf  # noqa: F401
        except ImportError as err:
            raise NbedDriverError(
                "NbedDriver needs a provider for the molecule integrals and the global Kohn-Sham "
                "calculation (outside the GPU hot path). The default provider uses PySCF, which is not "
                "installed; pass provider=... (see INTEGRATION.md)."
            ) from err

    def build_mol(self, config: NbedConfig):
        from pyscf import gto

        return gto.Mole(atom=config.geometry[2:], basis=config.basis, charge=config.charge, unit=config.unit,
                        spin=config.spin).build()

    def global_ks(self, config: NbedConfig, run_qmmm: bool = False):
        from pyscf import dft, qmmm

        ks = dft.UKS(self.build_mol(config))
        ks.conv_tol = config.convergence
        ks.xc = config.xc_functional
        ks.max_memory = config.max_ram_memory
        ks.max_cycle = config.max_dft_cycles
        ks.verbose = 1
        if run_qmmm:
            ks = qmmm.mm_charge(ks, config.mm_coords, config.mm_charges, config.mm_radii)
        ks.kernel()
        return ks

    def local_hf(self, config: NbedConfig, embedded_mol, backend=None, run_qmmm: bool = False):
        """GPU-backed UHF object over the embedded molecule's AO integrals.  With ``run_qmmm`` the
        core Hamiltonian carries the point-charge potential and ``energy_nuc()`` the MM-nuclear term,
        both taken from the ``qmmm.mm_charge``-wrapped PySCF object the reference builds
        (nbed/driver.py:246-253)."""
        from pyscf import qmmm
        from pyscf import scf as pyscf_scf

        from .scf import GpuUHF

        s = embedded_mol.intor("int1e_ovlp")
        eri = embedded_mol.intor("int2e", aosym="s1")
        if run_qmmm:
            wrapped = qmmm.mm_charge(pyscf_scf.UHF(embedded_mol), config.mm_coords, config.mm_charges,
                                     config.mm_radii)
            h, e_nuc = wrapped.get_hcore(), float(wrapped.energy_nuc())
        else:
            h, e_nuc = pyscf_scf.hf.get_hcore(embedded_mol), None
        local = GpuUHF(embedded_mol, s, h, eri, backend=backend)
        if e_nuc is not None:
            local.energy_nuc = lambda *args: e_nuc  # nuclear repulsion + nuclei-MM charges
        return local


    def local_ks(self, config: NbedConfig, embedded_mol, xc_functional: str, backend=None):
        """Embedded Kohn-Sham object (driver.py:289-313): PySCF's own ``dft.UKS`` -- the B3LYP
        quadrature is PySCF's -- which ``huzinaga_scf`` / ``_mu_embed`` drive through the protocol path
        (projector products, DIIS and eigensolves on the GPU; ``get_veff`` in PySCF)."""
        from pyscf import dft

        return dft.UKS(embedded_mol)

    def init_guess(self, config: NbedConfig, local_scf):
        """Starting density of the mu-shift SCF: PySCF's default 'minao' guess for the embedded
        molecule (electron count already overwritten), as scf.UHF.kernel() would take it."""
        from pyscf import scf as pyscf_scf

        return np.asarray(pyscf_scf.uhf.init_guess_by_minao(local_scf.mol))


class _TaggedVeff(np.ndarray):
    """ndarray carrying .ecoul / .exc like PySCF's tagged Kohn-Sham veff."""


class BuiltinHFProvider:
    """The out-of-path pieces without PySCF, for what ``nbed_amd.integrals`` and ``nbed_amd.xc``
    cover: molecules of H, C, N, O (F) in STO-3G, 6-31G, 6-31G* or cc-pVDZ with ``xc_functional`` = 'b3lyp' (the reference's
    default workflow: global B3LYP Kohn-Sham, nbed/driver.py:155-191), 'lda' or 'hf' (exact exchange:
    HF-in-HF embedding; PySCF's ``dft.UKS`` accepts the same strings).  Integrals come from the
    host-side McMurchie-Davidson engine, exchange-correlation from the host-side quadrature
    (``XCProvider``), Coulomb and exact exchange from libnbx (``GpuUKS`` / ``GpuUHF``)."""

    def __init__(self, backend=None, xc_grid: tuple[int, int] | None = None):
        """``xc_grid``: (radial points per heavy atom, polar angles) of the exchange-correlation quadrature, or
        ("lebedev", level) for the Treutler-Ahlrichs x pruned-Lebedev construction PySCF documents as its default
        (``nbed_amd.xc.build_grid``; None = that construction at its default level 3, which needs SciPy >= 1.15 for
        ``scipy.integrate.lebedev_rule`` -- pass (n_rad, n_theta) for the product grid on an older SciPy).  Matrix elements of
        v_xc[D_act] between VIRTUAL orbitals converge much more slowly -- the active density of a SPADE partition
        has near-nodal surfaces where the GGA potential is singular -- and need a finer grid."""
        self._be = backend
        self._cache = {}
        self.xc_grid = xc_grid

    @staticmethod
    def supports(config: NbedConfig) -> bool:
        from . import integrals

        from . import xc as xcmod

        no_mm = None in [config.mm_charges, config.mm_coords, config.mm_radii]
        return (str(config.xc_functional).lower().replace(" ", "") in xcmod.HYBRID_FRACTION and no_mm
                and integrals.supports(config.geometry, str(config.basis)))

    def _integrals(self, config: NbedConfig):
        from . import integrals

        key = (config.geometry, str(config.basis).lower(), str(config.unit))
        if key not in self._cache:
            self._cache[key] = integrals.molecule_integrals(config.geometry, str(config.basis), str(config.unit))
        return self._cache[key]

    def _eri_kwargs(self, config: NbedConfig, backend) -> dict:
        """``eri=`` / ``eri_packed=`` of the SCF objects of one molecule: the tensor goes to the device
        once and the packed J/K copy is made once (a 148-function molecule: 3.8 GB up the bus, 0.97 GB
        packed), however many mean-field objects a run builds (global KS, global HF, embedded HF per
        projector, DFT-in-DFT)."""
        from .backend import get_backend

        be = backend if backend is not None else (self._be if self._be is not None else get_backend())
        key = ("eri_device", id(be), config.geometry, str(config.basis).lower(), str(config.unit))
        if key not in self._cache:
            self._cache[key] = {"eri": be.asarray(self._integrals(config)["eri"]), "donor": None}
        slot = self._cache[key]
        out = {"eri": slot["eri"], "backend": be}
        if slot["donor"] is not None:
            out["eri_packed"] = slot["donor"].eri_packed_device()
        return out

    def _adopt(self, config: NbedConfig, obj):
        """Remember the first SCF object of a molecule as the owner of the packed copies."""
        key = ("eri_device", id(obj.be), config.geometry, str(config.basis).lower(), str(config.unit))
        slot = self._cache.get(key)
        if slot is not None:
            if slot["donor"] is None:
                slot["donor"] = obj
            obj._eri_shared = slot  # the rs-packed copy of the four-index transform is shared through it
        return obj

    def build_mol(self, config: NbedConfig):
        from .scf import Mole

        ints = self._integrals(config)
        nelectron = ints["nelectron"] - int(config.charge)
        spin = int(config.spin)
        if (nelectron + spin) % 2:
            raise NbedDriverError(f"{nelectron} electrons are incompatible with spin {spin}")
        nelec = ((nelectron + spin) // 2, (nelectron - spin) // 2)
        return Mole(ints["nao"], nelec, ao_slices=ints["ao_slices"], e_nuc=ints["e_nuc"], atom=config.geometry,
                    basis=config.basis, charge=config.charge)

    def global_hf(self, config: NbedConfig):
        """Converged global UHF (nbed/driver.py:106-124), on libnbx."""
        from .scf import GpuUHF

        ints = self._integrals(config)
        hf = self._adopt(config, GpuUHF(self.build_mol(config), ints["S"], ints["hcore"],
                                        **self._eri_kwargs(config, self._be)))
        hf.conv_tol = config.convergence
        hf.max_cycle = config.max_hf_cycles
        hf.kernel()
        return hf

    def _xc_provider(self, config: NbedConfig, xc_functional: str):
        """Quadrature of the semi-local part of ``xc_functional`` for this molecule (None for 'hf')."""
        from . import integrals
        from . import xc as xcmod

        name = str(xc_functional).lower().replace(" ", "")
        if name == "hf":
            return None
        key = ("xc", config.geometry, str(config.basis).lower(), str(config.unit), name)
        if key not in self._cache:
            atoms = integrals.parse_geometry(config.geometry, str(config.unit))
            if self.xc_grid is None:
                grid = {}
            elif isinstance(self.xc_grid[0], str):  # ("lebedev", level): Treutler-Ahlrichs x pruned Lebedev
                grid = {"scheme": str(self.xc_grid[0]), "level": int(self.xc_grid[1])}
            else:  # (n_rad, n_theta): the product grid
                grid = {"scheme": "product", "n_rad": int(self.xc_grid[0]), "n_theta": int(self.xc_grid[1])}
            self._cache[key] = xcmod.XCProvider(atoms, integrals.Basis(atoms, str(config.basis)), name, **grid)
        return self._cache[key]

    def _uks(self, config: NbedConfig, mol, xc_functional: str, backend):
        from . import xc as xcmod
        from .scf import GpuUKS

        ints = self._integrals(config)
        return self._adopt(config, GpuUKS(mol, ints["S"], ints["hcore"], xc=str(xc_functional),
                                          hyb=xcmod.hybrid_fraction(xc_functional),
                                          xc_provider=self._xc_provider(config, xc_functional),
                                          **self._eri_kwargs(config, backend)))

    def global_ks(self, config: NbedConfig, run_qmmm: bool = False):
        from .scf import GpuUHF

        if run_qmmm or not self.supports(config):
            raise NbedDriverError("BuiltinHFProvider covers xc_functional in ('b3lyp', 'lda', 'hf'), H/C/N/O in "
                                  "STO-3G, 6-31G(*), cc-pVDZ, no QM/MM")
        if str(config.xc_functional).lower() != "hf":
            ks = self._uks(config, self.build_mol(config), config.xc_functional, self._be)
            ks.conv_tol = config.convergence
            ks.max_cycle = config.max_dft_cycles
            ks.kernel()
            return ks
        ints = self._integrals(config)

        class GlobalHF(GpuUHF):
            """UHF presented through the Kohn-Sham protocol: veff = J - K with ecoul = 1/2 tr(D J)
            and exc = -1/2 sum_x tr(D_x K_x), as PySCF tags its UKS veff."""

            xc = "hf"

            def get_veff(self, mol=None, dm=None, dm_last=0, vhf_last=0, hermi=1):
                dm = self.make_rdm1() if dm is None else np.asarray(dm)
                dm3 = np.array((dm * 0.5, dm * 0.5)) if dm.ndim == 2 else dm
                vj, vk = self.get_jk(mol, dm3)
                v = np.asarray(vj[0] + vj[1] - vk).view(_TaggedVeff)
                v.ecoul = 0.5 * float(np.einsum("ij,ji->", vj[0] + vj[1], dm3[0] + dm3[1]))
                v.exc = -0.5 * float(np.einsum("xij,xji->", vk, dm3))
                return v

        ks = self._adopt(config, GlobalHF(self.build_mol(config), ints["S"], ints["hcore"],
                                          **self._eri_kwargs(config, self._be)))
        ks.conv_tol = config.convergence
        ks.max_cycle = config.max_dft_cycles
        ks.kernel()
        return ks

    def local_ks(self, config: NbedConfig, embedded_mol, xc_functional: str, backend=None):
        """Embedded Kohn-Sham object of ``xc_functional`` (DFT-in-DFT, nbed/driver.py:289-313)."""
        return self._uks(config, embedded_mol, xc_functional, backend)

    def local_hf(self, config: NbedConfig, embedded_mol, backend=None, run_qmmm: bool = False):
        from .scf import GpuUHF

        if run_qmmm:
            raise NbedDriverError("BuiltinHFProvider has no QM/MM point-charge integrals")
        ints = self._integrals(config)
        return self._adopt(config, GpuUHF(embedded_mol, ints["S"], ints["hcore"], **self._eri_kwargs(config, backend)))


class NbedDriver:
    """Run projection-based embedding and produce the embedded active-space Hamiltonian."""

    def __init__(self, config: NbedConfig, provider=None, backend=None, hamiltonian_format: str = "dense"):
        """``hamiltonian_format``: "dense" -- ``result["second_quantised"]`` is the reference's
        ``(constant, h1 (2n,2n), h2 (2n,2n,2n,2n))``; "spatial" -- a ``SpatialHamiltonian`` (three unique
        spin blocks, 16/3 times smaller; ``.to_dense()`` gives the former)."""
        if hamiltonian_format not in ("dense", "spatial"):
            raise NbedDriverError(f"unknown hamiltonian_format {hamiltonian_format!r}")
        self.hamiltonian_format = hamiltonian_format
        self.config = config
        self._provider = provider
        self._be = backend
        self.localized_system: LocalizedSystem
        self.two_e_cross = None
        self.mu: dict = None
        self.huzinaga: dict = None
        self.active_geometry = f"{config.n_active_atoms}\n\n" + "\n".join(
            config.geometry.splitlines()[2 : 2 + config.n_active_atoms]
        )
        self._restricted_scf = False  # the reference is always unrestricted (driver.py:69-78)
        self.run_qmmm = None not in [config.mm_charges, config.mm_coords, config.mm_radii]

    # ------------------------------------------------------------------ plumbing
    @property
    def be(self):
        if self._be is None:
            self._be = get_backend()
        return self._be

    @property
    def provider(self):
        if self._provider is None:
            try:
                self._provider = PySCFProvider()
            except NbedDriverError:
                # no PySCF: the built-in integrals cover HF-in-HF on small s/p molecules in STO-3G
                if not BuiltinHFProvider.supports(self.config):
                    raise
                self._provider = BuiltinHFProvider(self._be)
        return self._provider

    def _build_mol(self):
        return self.provider.build_mol(self.config)

    @cached_property
    def _global_ks(self):
        """Converged global (cheap) Kohn-Sham object: source of densities and potentials."""
        ks = self.provider.global_ks(self.config, self.run_qmmm) if self.run_qmmm else self.provider.global_ks(self.config)
        if getattr(ks, "converged", True) is not True:
            logger.warning("(cheap) global DFT calculation has NOT converged!")
        return ks

    def _unsupported(self, what: str):
        raise NotImplementedError(
            f"{what} is outside the MI355X hot path (SURVEY.md section 2, component 4); run it with PySCF on the "
            "embedded SCF object returned in result['scf']."
        )

    @cached_property
    def _global_hf(self):
        if hasattr(self.provider, "global_hf"):
            return self.provider.global_hf(self.config)
        self._unsupported("The global Hartree-Fock reference calculation")

    def _ccsd_of(self, scf_obj, frozen=None):
        """CCSD of an SCF object: PySCF's solver where there is one, else -- for small orbital spaces -- the
        spin-orbital equations over the Hamiltonian ``HamiltonianBuilder`` makes of it (``nbed_amd.ccsd``)."""
        try:
            return run_emb_ccsd(scf_obj, frozen, self.config.convergence, self.config.max_ram_memory)[0]
        except NbedDriverError:
            from . import ccsd

            n = np.asarray(scf_obj.mo_coeff).shape[-1]
            if frozen is not None or 2 * n > ccsd.MAX_SPIN_ORBITALS:
                raise
            const, h1, h2 = HamiltonianBuilder(scf_obj, scf_obj.energy_nuc(), backend=self.be).build()
            # the reference determinant from the object's own occupations (alpha on the even spin-orbital indices):
            # after the environment is deleted / the virtuals are re-ordered the occupied MOs need not lead
            mo_occ = np.asarray(scf_obj.mo_occ)
            if mo_occ.ndim == 1:
                mo_occ = np.array((mo_occ > 0, mo_occ > 1), dtype=float)  # (alpha holds the singly occupied orbitals)
            occupied = ([2 * int(i) for i in np.flatnonzero(mo_occ[0] > 0)]
                        + [2 * int(i) + 1 for i in np.flatnonzero(mo_occ[1] > 0)])
            na, nb = scf_obj.mol.nelec
            if (int(np.sum(mo_occ[0] > 0)), int(np.sum(mo_occ[1] > 0))) != (int(na), int(nb)):
                raise NbedDriverError(f"mo_occ holds {int(np.sum(mo_occ[0] > 0))} + {int(np.sum(mo_occ[1] > 0))} occupied "
                                      f"orbitals, the molecule {na} + {nb} electrons")
            return ccsd.solve(const, h1, h2, occupied, conv_tol=min(self.config.convergence, 1e-8))

    @cached_property
    def _global_ccsd(self):
        """Global CCSD reference (nbed/driver.py:122-136)."""
        return self._ccsd_of(self._global_hf)

    @cached_property
    def _global_fci(self):
        """Global FCI reference (nbed/driver.py:139-153): PySCF's solver on the global HF object, or for
        small molecules the exact diagonalisation of its full-space Hamiltonian (``nbed_amd.fci``)."""
        hf = self._global_hf
        try:
            return run_emb_fci(hf, None, self.config.convergence, self.config.max_ram_memory)
        except NbedDriverError:
            from . import fci

            if 2 * np.asarray(hf.mo_coeff).shape[-1] > fci.MAX_SPIN_ORBITALS:
                raise
            const, h1, h2 = HamiltonianBuilder(hf, hf.energy_nuc(), backend=self.be).build()
            return fci.ground_state(const, h1, h2, hf.mol.nelec)

    # ------------------------------------------------------------------ localisation
    def _localize(self) -> LocalizedSystem:
        match self.config.localization:
            case OccupiedLocalizerTypes.SPADE:
                localizer = SPADELocalizer(self._global_ks, self.config.n_active_atoms,
                                           max_shells=self.config.max_shells, n_mo_overwrite=self.n_mo_overwrite,
                                           backend=self.be)
            case OccupiedLocalizerTypes.BOYS:
                localizer = BOYSLocalizer(self._global_ks, self.config.n_active_atoms)
            case OccupiedLocalizerTypes.IBO:
                localizer = IBOLocalizer(self._global_ks, self.config.n_active_atoms)
            case OccupiedLocalizerTypes.PM:
                localizer = PMLocalizer(self._global_ks, self.config.n_active_atoms)
        self.localizer = localizer
        return localizer.localize()

    # ------------------------------------------------------------------ embedded SCF objects
    def _init_embedded_mol(self):
        """Molecule whose electron count is overwritten with the active one (driver.py:262-287)."""
        mol = self._build_mol()
        inds = np.asarray(self.localized_system.active_mo_inds)
        if inds.ndim == 1:
            n = len(inds)
            mol.nelectron, mol.nelec, mol.spin = 2 * n, (n, n), 0
        else:
            na, nb = len(inds[0]), len(inds[1])
            mol.nelectron, mol.nelec, mol.spin = na + nb, (na, nb), na - nb
        self._electron = mol.nelectron
        return mol

    def _init_local_hf(self):
        """Embedded HF object for the active subsystem (driver.py:230-260).  Under QM/MM the
        provider must put the point-charge field into the local hcore and energy_nuc as well
        (the reference wraps the local object in qmmm.mm_charge, :246-253): a provider that cannot
        is refused, never silently run without the field."""
        if self.run_qmmm:
            import inspect

            if "run_qmmm" not in inspect.signature(self.provider.local_hf).parameters:
                raise NbedDriverError(
                    "QM/MM was requested (mm_coords / mm_charges / mm_radii) but this provider's local_hf() "
                    "cannot add the point-charge field to the embedded SCF object")
            local_hf = self.provider.local_hf(self.config, self._init_embedded_mol(), backend=self.be, run_qmmm=True)
        else:
            local_hf = self.provider.local_hf(self.config, self._init_embedded_mol(), backend=self.be)
        local_hf.max_memory = self.config.max_ram_memory
        local_hf.conv_tol = self.config.convergence
        local_hf.max_cycle = self.config.max_hf_cycles
        local_hf.verbose = 1
        return local_hf

    def _init_local_ks(self, xc_functional: str):
        """Embedded Kohn-Sham object for the active subsystem (driver.py:289-313): the provider's
        ``local_ks(config, embedded_mol, xc, backend=)``."""
        if not hasattr(self.provider, "local_ks"):
            self._unsupported("An embedded Kohn-Sham object (DFT-in-DFT) with this provider")
        local_ks = self.provider.local_ks(self.config, self._init_embedded_mol(), xc_functional, backend=self.be)
        local_ks.max_memory = self.config.max_ram_memory
        local_ks.conv_tol = self.config.convergence
        local_ks.xc = xc_functional
        local_ks.verbose = 1
        return local_ks

    # ------------------------------------------------------------------ subsystem DFT (inputs)
    def _subsystem_dft(self, global_ks, localized_system):
        """Energies of the active / environment densities and their two-electron cross term
        (driver.py:315-431).  The Kohn-Sham potential itself comes from ``global_ks``."""
        be = self.be

        def ks_components(dm):
            two_e = global_ks.get_veff(dm=dm)
            j_mat = global_ks.get_j(dm=dm)
            dm_tot = dm[0] + dm[1] if dm.ndim == 3 else dm
            e = float(be.trace_prod(be.asarray(np.asarray(global_ks.get_hcore())), be.asarray(dm_tot)))
            return e + two_e.ecoul + two_e.exc, two_e, np.asarray(j_mat)

        dm_act, dm_env = localized_system.dm_active, localized_system.dm_enviro
        e_act, two_e_act, j_act = ks_components(dm_act)
        e_env, two_e_env, j_env = ks_components(dm_env)

        total_dm = dm_act + dm_env
        if dm_act.ndim == 3:
            total_dm = total_dm[0] + total_dm[1]
        e_xc_total = global_ks.get_veff(dm=total_dm).exc

        def dot(a, b):  # einsum("ij,ij")
            return float(be.dots(be.asarray(a).reshape(-1), be.asarray(b).reshape(1, -1))[0])

        if dm_act.ndim == 2:
            j_cross = 0.5 * (dot(dm_act, j_env) + dot(dm_env, j_act))
        else:
            j_cross = 0.5 * sum(
                dot(dm_act[x], j_env[y]) + dot(dm_env[x], j_act[y]) for x in range(2) for y in range(2)
            )
        k_cross = 0.0  # the projection makes the kinetic cross term vanish (driver.py:415)
        xc_cross = e_xc_total - two_e_act.exc - two_e_env.exc
        return e_act, e_env, j_cross + k_cross + xc_cross

    @cached_property
    def _env_projector(self) -> np.ndarray:
        """P = S D_env S per spin (driver.py:433-449)."""
        be = self.be
        s = be.asarray(np.asarray(self._global_ks.get_ovlp()))
        d = be.asarray(np.asarray(self.localized_system.dm_enviro))
        return be.to_host(be.gemm(be.gemm(s, d), s))

    # ------------------------------------------------------------------ projectors
    def _mu_embed(self, localized_scf, embedding_potential):
        """mu-shift projector: v_emb = mu P + V_emb added to hcore, then the SCF kernel
        (driver.py:500-538)."""
        v_emb = (self.config.mu_level_shift * self._env_projector) + embedding_potential
        if v_emb.ndim == 3:
            localized_scf.energy_elec = lambda *args: energy_elec(localized_scf, *args)
        hcore_std = localized_scf.get_hcore
        localized_scf.get_hcore = lambda *args: hcore_std(*args) + v_emb
        # PySCF's kernel() starts from the 'minao' guess (nbed/driver.py:533 -> scf.hf.kernel); a provider
        # that can make one supplies it, otherwise the core-Hamiltonian guess of GpuUHF is used
        dm0 = None
        if hasattr(self.provider, "init_guess"):
            dm0 = self.provider.init_guess(self.config, localized_scf)
        localized_scf.kernel(dm0) if dm0 is not None else localized_scf.kernel()
        if not localized_scf.converged:
            logger.warning("mu-shift embedded SCF has NOT converged in %s cycles.", localized_scf.max_cycle)
        logger.info(f"Embedded scf energy MU_SHIFT: {localized_scf.e_tot}, converged: {localized_scf.converged}")
        return localized_scf, v_emb

    def _huzinaga_embed(self, active_scf, embedding_potential, localized_system, dmat_initial_guess=None):
        """Huzinaga projector: own SCF loop, results written back to the SCF object
        (driver.py:540-632)."""
        if localized_system.c_loc_virt is not None:
            c_virt = np.asarray(localized_system.c_loc_virt)
            virtual_projector = c_virt @ np.swapaxes(c_virt, -1, -2)
            dm_environment_virtual = np.identity(c_virt.shape[-2]) - localized_system.dm_loc_occ - virtual_projector
        else:
            dm_environment_virtual = None

        c_emb, e_emb, dm_emb, huz_op, conv = huzinaga_scf(
            active_scf, embedding_potential, localized_system.dm_enviro,
            dm_environment_virtual=dm_environment_virtual, dm_conv_tol=1e-6, dm_initial_guess=dmat_initial_guess,
        )
        hcore_std = active_scf.get_hcore()
        v_emb = huz_op + embedding_potential
        active_scf.get_hcore = lambda *args: hcore_std + v_emb
        if np.asarray(localized_system.c_active).ndim == 3:
            active_scf.energy_elec = lambda *args: energy_elec(active_scf, *args)
        active_scf.mo_occ = active_scf.get_occ(e_emb, c_emb)
        if localized_system.c_loc_virt is not None:
            occ_any = np.sum(active_scf.mo_occ, axis=0)
            active_scf.mo_coeff = np.concatenate(
                (c_emb[..., occ_any > 0], c_emb[..., occ_any == 0][: localized_system.c_loc_virt.shape[-1]]), axis=2
            )
            active_scf.mo_occ = active_scf.mo_occ[: active_scf.mo_coeff.shape[-1]]
        else:
            active_scf.mo_coeff = c_emb
        active_scf.mo_energy = e_emb
        active_scf.e_tot = active_scf.energy_tot(dm=dm_emb)
        active_scf.converged = conv
        logger.info(f"Embedded scf energy HUZINAGA: {active_scf.e_tot}")
        return active_scf, v_emb

    # ------------------------------------------------------------------ environment deletion
    def _delete_environment(self, projector, scf, localized_system, env_projector):
        c_env = np.asarray(localized_system.c_enviro)
        if c_env.ndim == 2:
            n_env = c_env.shape[-1]
            scf.mo_coeff, scf.mo_energy, scf.mo_occ = self._delete_spin_environment(
                projector, n_env, scf.mo_coeff, scf.mo_energy, scf.mo_occ, env_projector)
        else:
            n_env = len(set(localized_system.enviro_mo_inds[0]).union(localized_system.enviro_mo_inds[1]))
            a = self._delete_spin_environment(projector, n_env, scf.mo_coeff[0], scf.mo_energy[0], scf.mo_occ[0],
                                              env_projector[0])
            b = self._delete_spin_environment(projector, n_env, scf.mo_coeff[1], scf.mo_energy[1], scf.mo_occ[1],
                                              env_projector[1])
            scf.mo_coeff = np.array([a[0], b[0]])
            scf.mo_energy = np.array([a[1], b[1]])
            scf.mo_occ = np.array([a[2], b[2]])
        return scf

    def _delete_spin_environment(self, projector, n_env_mo, mo_coeff, mo_energy, mo_occ, environment_projector):
        """Drop the environment orbitals of one spin (driver.py:715-791)."""
        mo_coeff = np.asarray(mo_coeff)
        n_mo = mo_coeff.shape[-1]
        match projector:
            case ProjectorTypes.HUZ:
                # einsum("ij, ki -> i", C^T, P C) = colsum(C)_i * colsum(P C)_i
                be = self.be
                c_d = be.asarray(mo_coeff)
                pc = be.gemm(be.asarray(np.asarray(environment_projector)), c_d)
                ones = be.asarray(np.ones((1, mo_coeff.shape[0])))
                score = be.to_host(be.gemm(ones, c_d))[0] * be.to_host(be.gemm(ones, pc))[0]
                frozen = list(score.argsort()[::-1][:n_env_mo])
            case ProjectorTypes.MU:
                frozen = list(range(n_mo - n_env_mo, n_mo))
        keep = [i for i in range(n_mo) if i not in frozen]
        logger.info(f"Orbital indices for embedded system: {keep}")
        logger.info(f"Orbital indices removed from embedded system: {frozen}")
        return mo_coeff[:, keep], np.asarray(mo_energy)[keep], np.asarray(mo_occ)[keep]

    def _dft_in_dft(self, projection_method):
        """driver.py:793-806."""
        return dft_in_dft(self, projection_method)

    def _run_emb_ccsd(self, emb_scf, frozen=None):
        """driver.py:451-474 -> ``(ccsd, e_corr)``.  Without PySCF, small active spaces go through
        ``nbed_amd.ccsd`` on the active-space Hamiltonian of this same embedded object."""
        cc = self._ccsd_of(emb_scf, frozen)
        return cc, cc.e_corr

    def _run_emb_fci(self, emb_scf, frozen=None):
        """driver.py:476-498.  Without PySCF, small active spaces are diagonalised exactly from the
        active-space Hamiltonian of this same embedded object (``nbed_amd.fci``)."""
        try:
            return run_emb_fci(emb_scf, frozen, self.config.convergence, self.config.max_ram_memory)
        except NbedDriverError:
            from . import fci

            n = np.asarray(emb_scf.mo_coeff).shape[-1]
            if frozen is not None or 2 * n > fci.MAX_SPIN_ORBITALS:
                raise
            # Hamiltonian with the nuclear repulsion as its constant: its ground state is what PySCF's
            # FCI object reports as e_tot (tests/test_builder.py:55-120)
            const, h1, h2 = HamiltonianBuilder(emb_scf, emb_scf.energy_nuc(), backend=self.be).build()
            return fci.ground_state(const, h1, h2, emb_scf.mol.nelec)

    # ------------------------------------------------------------------ the embedding
    def embed(self, init_huzinaga_rhf_with_mu: bool = False,
              n_mo_overwrite: tuple[int | None, int | None] = (None, None)) -> None:
        """Run the embedded SCF calculation(s) (driver.py:808-923)."""
        cfg = self.config
        if cfg.virtual_localization is VirtualLocalizerTypes.PROJECTED_AO:
            raise NotImplementedError("PAO not yet fully implemented.")

        self.e_nuc = self._global_ks.energy_nuc()
        if n_mo_overwrite is not None and n_mo_overwrite != (None, None):
            self.n_mo_overwrite = n_mo_overwrite
        else:
            self.n_mo_overwrite = cfg.n_mo_overwrite

        self.localized_system = self._localize()
        self.e_act, self.e_env, self.two_e_cross = self._subsystem_dft(self._global_ks, self.localized_system)

        total_dm = self.localized_system.dm_active + self.localized_system.dm_enviro
        g_act_and_env = self._global_ks.get_veff(dm=total_dm)
        g_act = self._global_ks.get_veff(dm=self.localized_system.dm_active)
        embedding_potential = np.asarray(g_act_and_env) - np.asarray(g_act)
        self.embedding_potential = embedding_potential
        logger.info(f"DFT potential average {np.mean(embedding_potential)}.")

        if cfg.projector in [ProjectorTypes.MU, ProjectorTypes.BOTH] or init_huzinaga_rhf_with_mu:
            local_hf = self._init_local_hf()
            embedded_scf, v_emb = self._mu_embed(local_hf, embedding_potential)
            self.mu = self.post_embed(embedded_scf, v_emb, ProjectorTypes.MU)

        if cfg.projector in [ProjectorTypes.HUZ, ProjectorTypes.BOTH]:
            local_hf = self._init_local_hf()
            dmat_initial_guess: Optional[np.ndarray] = (
                self.mu["scf"].make_rdm1() if init_huzinaga_rhf_with_mu else None
            )
            embedded_scf, v_emb = self._huzinaga_embed(local_hf, embedding_potential, self.localized_system,
                                                       dmat_initial_guess)
            self.huzinaga = self.post_embed(embedded_scf, v_emb, ProjectorTypes.HUZ)

        match cfg.projector:
            case ProjectorTypes.MU:
                self.embedded_scf = self.mu["scf"]
                self.classical_energy = self.mu["classical_energy"]
            case ProjectorTypes.HUZ:
                self.embedded_scf = self.huzinaga["scf"]
                self.classical_energy = self.huzinaga["classical_energy"]
            case ProjectorTypes.BOTH:
                logger.warning("Outputting both mu and huzinaga embedding results as tuple.")
                self.embedded_scf = (self.mu["scf"], self.huzinaga["scf"])
                self.classical_energy = (self.mu["classical_energy"], self.huzinaga["classical_energy"])
        logger.info("Embedding complete.")

    def post_embed(self, embedded_scf, v_emb, projector: ProjectorTypes) -> dict:
        """Projector-dependent components of the embedding (driver.py:925-1041)."""
        cfg = self.config
        be = self.be
        result = {"scf": embedded_scf.copy(), "v_emb": v_emb}
        result["mo_energies_emb_pre_del"] = result["scf"].mo_energy
        result["scf"] = self._delete_environment(projector, result["scf"], self.localized_system, self._env_projector)
        result["mo_energies_emb_post_del"] = result["scf"].mo_energy

        def dot(a, b):  # einsum("ij,ij")
            return float(be.dots(be.asarray(np.asarray(a)).reshape(-1), be.asarray(np.asarray(b)).reshape(1, -1))[0])

        dm_active = self.localized_system.dm_active
        if dm_active.ndim == 2:
            result["correction"] = dot(result["v_emb"], dm_active)
            result["beta_correction"] = 0
        else:
            result["correction"] = dot(result["v_emb"][0], dm_active[0])
            result["beta_correction"] = dot(result["v_emb"][1], dm_active[1])

        match cfg.virtual_localization:
            case VirtualLocalizerTypes.CONCENTRIC:
                result["cl"] = ConcentricLocalizer(result["scf"], cfg.n_active_atoms, max_shells=cfg.max_shells,
                                                   backend=be)
                result["scf"] = result["cl"].localize_virtual()
            case VirtualLocalizerTypes.DISABLE:
                logger.debug("Not performing virtual localization.")

        corr = result["correction"] + result["beta_correction"]
        result["e_rhf"] = result["scf"].e_tot + self.e_env + self.two_e_cross - corr
        result["classical_energy"] = self.e_env + self.two_e_cross + self.e_nuc - corr

        if cfg.run_ccsd_emb is True:
            ccsd_emb, _ = self._run_emb_ccsd(result["scf"])
            result["e_ccsd"] = ccsd_emb.e_tot + self.e_env + self.two_e_cross - corr
            result["ccsd_emb"] = ccsd_emb.e_tot - self.e_nuc
        if cfg.run_fci_emb is True:
            fci_emb = self._run_emb_fci(result["scf"])
            result["e_fci"] = fci_emb.e_tot + self.e_env + self.two_e_cross - corr
            result["fci_emb"] = fci_emb.e_tot - self.e_nuc
        result["hf_emb"] = result["scf"].e_tot - self.e_nuc
        if cfg.run_dft_in_dft is True:
            result.update(self._dft_in_dft(projector))

        builder = HamiltonianBuilder(result["scf"], result["classical_energy"], backend=be)
        spatial = getattr(self, "hamiltonian_format", "dense") == "spatial"
        result["second_quantised"] = builder.build_spatial() if spatial else builder.build()
        return result


# ---------------------------------------------------------------------- consumers of the embedded SCF object
def _as_pyscf_uhf(emb_scf):
    """A PySCF ``scf.UHF`` carrying the embedded object's molecule, orbitals and patched hcore, for
    the PySCF post-HF solvers (they index PySCF-internal attributes a GpuUHF does not have)."""
    try:
        from pyscf import scf as pyscf_scf
    except ImportError as err:
        raise NbedDriverError(
            "embedded CCSD / FCI are PySCF's cc.CCSD / fci.FCI run on the embedded SCF object "
            "(nbed/driver.py:1044-1135); PySCF is not installed. The active-space Hamiltonian in "
            "result['second_quantised'] is what those solvers diagonalise.") from err
    if isinstance(emb_scf, pyscf_scf.hf.SCF):
        return emb_scf
    mol = emb_scf.mol
    if not hasattr(mol, "intor"):
        raise NbedDriverError("embedded CCSD / FCI need an SCF object over a PySCF molecule (PySCFProvider)")
    mf = pyscf_scf.UHF(mol)
    hcore = np.asarray(emb_scf.get_hcore())
    mf.get_hcore = lambda *args: hcore
    mf.mo_coeff, mf.mo_occ, mf.mo_energy = emb_scf.mo_coeff, emb_scf.mo_occ, emb_scf.mo_energy
    mf.e_tot, mf.converged = emb_scf.e_tot, emb_scf.converged
    if "energy_nuc" in vars(emb_scf):
        mf.energy_nuc = emb_scf.energy_nuc
    return mf


def run_emb_fci(emb_pyscf_scf_rhf, frozen: Optional[list] = None, convergence: Optional[float] = 1e-6,
                max_ram_memory: Optional[int] = 4000):
    """FCI on the embedded SCF object (nbed/driver.py:1044-1102): PySCF's ``fci.FCI`` (``mcscf.CASSCF``
    with ``frozen``), with the one-electron part taken from the 3-D embedded hcore.  Outside the MI355X hot
    path: delegated to a local PySCF, refused with ``NbedDriverError`` without one."""
    mf = _as_pyscf_uhf(emb_pyscf_scf_rhf)
    from pyscf import fci

    if frozen is None:
        fci_scf = fci.FCI(mf)
    else:
        from pyscf import mcscf

        fci_scf = mcscf.CASSCF(mf, mf.mol.nelec, mf.mol.nao - len(frozen))
        fci_scf.sort_mo([i + 1 for i in range(mf.mol.nao) if i not in frozen])
    fci_scf.conv_tol = convergence
    fci_scf.max_memory = max_ram_memory
    fci_scf.verbose = 1
    hcore = np.asarray(mf.get_hcore())
    if hcore.ndim == 3:
        mo = mf.mo_coeff
        fci_scf.kernel(h1e=[mo[0].T @ hcore[0] @ mo[0], mo[1].T @ hcore[1] @ mo[1]])
    else:
        fci_scf.kernel()
    logger.info(f"FCI embedding energy: {fci_scf.e_tot}")
    return fci_scf


def run_emb_ccsd(emb_pyscf_scf_rhf, frozen: Optional[list] = None, convergence: float = 1e-6,
                 max_ram_memory: int = 4000):
    """CCSD on the embedded SCF object (nbed/driver.py:1105-1135) -> ``(ccsd, e_ccsd_corr)``;
    PySCF's ``cc.CCSD``, delegated like ``run_emb_fci``."""
    mf = _as_pyscf_uhf(emb_pyscf_scf_rhf)
    from pyscf import cc

    ccsd = cc.CCSD(mf, frozen=frozen)
    ccsd.conv_tol = convergence
    ccsd.max_memory = max_ram_memory
    ccsd.verbose = 2
    e_ccsd_corr, _, _ = ccsd.kernel()
    logger.info(f"Embedded CCSD energy: {e_ccsd_corr}")
    return ccsd, e_ccsd_corr


def dft_in_dft(driver: "NbedDriver", projection_method: ProjectorTypes) -> dict:
    """Energy of DFT-in-DFT embedding (nbed/driver.py:1138-1231): the embedded SCF is repeated
    with a Kohn-Sham object of the GLOBAL functional in place of Hartree-Fock; with an exact
    embedding it reproduces the global Kohn-Sham energy (tests/test_driver.py:83-88).  Runs on the
    GPU path: the provider's ``local_ks`` object (``GpuUKS``) goes through the same
    ``_mu_embed`` / ``_huzinaga_embed`` (KS branch of ``huzinaga_scf``) as the Hartree-Fock one."""
    be = driver.be
    result = {}
    e_nuc = driver._global_ks.energy_nuc()
    local_ks = driver._init_local_ks(driver._global_ks.xc)
    hcore_std = np.asarray(local_ks.get_hcore())
    match projection_method:
        case ProjectorTypes.MU:
            result["scf_dft"], result["v_emb_dft"] = driver._mu_embed(local_ks, driver.embedding_potential)
        case ProjectorTypes.HUZ:
            result["scf_dft"], result["v_emb_dft"] = driver._huzinaga_embed(
                local_ks, driver.embedding_potential, driver.localized_system)
    result["scf_dft"] = driver._delete_environment(projection_method, result["scf_dft"], driver.localized_system,
                                                   driver._env_projector)

    def dot(a, b):  # einsum("ij,ij")
        return float(be.dots(be.asarray(np.asarray(a)).reshape(-1), be.asarray(np.asarray(b)).reshape(1, -1))[0])

    dm_active = driver.localized_system.dm_active
    if dm_active.ndim == 2:
        y_emb = result["scf_dft"].make_rdm1()
        result["dft_correction"] = dot(result["v_emb_dft"], y_emb - dm_active)
        result["dft_correction_beta"] = 0
        veff = result["scf_dft"].get_veff(dm=y_emb)
        rks_e_elec = veff.exc + veff.ecoul + dot(hcore_std, y_emb)
    else:
        y_a, y_b = result["scf_dft"].make_rdm1()
        result["dft_correction"] = dot(result["v_emb_dft"][0], y_a - dm_active[0])
        result["dft_correction_beta"] = dot(result["v_emb_dft"][1], y_b - dm_active[1])
        veff = result["scf_dft"].get_veff(dm=np.array([y_a, y_b]))
        rks_e_elec = veff.exc + veff.ecoul + dot(hcore_std, y_a) + dot(hcore_std, y_b)
    result["e_dft_in_dft"] = (rks_e_elec + driver.e_env + driver.two_e_cross + result["dft_correction"]
                              + result["dft_correction_beta"] + e_nuc)
    result["emb_dft"] = rks_e_elec
    return result
