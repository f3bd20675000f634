"""End-to-end NbedDriver run of a REAL molecule on the GPU, nothing injected: integrals from libnbx's
host engine (nbx_host_eri), exchange-correlation quadrature on the device, the embedding hot path on
libnbx.  Default: BASELINE.json configs[2], octane / 6-31G*, 4 active atoms, SPADE + concentric
localization (148 AOs -- the size the bench's synthetic workload stands in for).

    python tools/e2e_molecule.py [octane|water] [basis] [n_active_atoms]
Environment: E2E_PROJECTOR (both), E2E_FORMAT (spatial), E2E_GRID ("96,28" radial, theta points)."""
import os
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from molecules import octane_xyz  # noqa: E402
from nbed_amd import NbedConfig, nbed  # noqa: E402
from nbed_amd import xc as xcmod  # noqa: E402
from nbed_amd.backend import HipBackend  # noqa: E402
from nbed_amd.driver import BuiltinHFProvider  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "octane"
basis = sys.argv[2] if len(sys.argv) > 2 else "6-31g*"
nact = int(sys.argv[3]) if len(sys.argv) > 3 else 4
geom = octane_xyz() if name == "octane" else "3\n\nO   0.0000  0.000  0.115\nH   0.0000  0.754  -0.459\nH   0.0000  -0.754  -0.459"
n_rad, n_theta = (int(x) for x in os.environ.get("E2E_GRID", "96,28").split(","))

be = HipBackend()
prov = BuiltinHFProvider(be)
_init = xcmod.XCProvider.__init__
xcmod.XCProvider.__init__ = lambda self, atoms, bs, xc, **kw: _init(self, atoms, bs, xc, n_rad=n_rad, n_theta=n_theta)

cfg = NbedConfig(geometry=geom, n_active_atoms=nact, basis=basis, xc_functional="b3lyp", convergence=1e-8,
                 max_hf_cycles=100, max_dft_cycles=100, projector=os.environ.get("E2E_PROJECTOR", "both"),
                 localization="spade", virtual_localization="cl", max_shells=4)
t0 = time.perf_counter()
ints = prov._integrals(cfg)
t1 = time.perf_counter()
print(f"{name}/{basis}: {ints['nao']} AOs, {ints['nelectron']} electrons; integrals (host, all cores) {t1 - t0:.2f} s", flush=True)
grid = prov._xc_provider(cfg, "b3lyp")
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"quadrature grid: {grid.points.shape[0]} points on {grid.device}, {t2 - t1:.2f} s", flush=True)
fmt = os.environ.get("E2E_FORMAT", "spatial")
import nbed_amd.driver as drv_mod  # noqa: E402

_huz = drv_mod.huzinaga_scf
_cyc = be.huz_cycle
_count = {"cycles": 0}


def _counted_cycle(*a, **k):
    _count["cycles"] += 1
    return _cyc(*a, **k)


def _timed_huz(*a, **k):
    be.huz_cycle = _counted_cycle
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = _huz(*a, **k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    n = _count["cycles"]
    print(f"huzinaga_scf: {dt * 1e3:.1f} ms wall, {n} fused cycles queued" + (f" ({n / dt:.0f} cycles/s)" if n else ""), flush=True)
    return out


drv_mod.huzinaga_scf = _timed_huz
import cProfile  # noqa: E402
import pstats  # noqa: E402

pr = cProfile.Profile()
pr.enable()
drv = nbed(cfg, provider=prov, backend=be, hamiltonian_format=fmt)
torch.cuda.synchronize()
pr.disable()
if os.environ.get("E2E_PROFILE", "1") != "0":
    pstats.Stats(pr).sort_stats("cumulative").print_stats(int(os.environ.get("E2E_NSTATS", "45")))
t3 = time.perf_counter()
print(f"NbedDriver.embed(): {t3 - t2:.2f} s  (whole run {t3 - t0:.2f} s)", flush=True)
ks = drv._global_ks
print(f"global B3LYP e_tot {ks.e_tot:.10f}  converged {ks.converged} in {ks.cycles} cycles; electrons on the grid {grid.nelec_last:.8f}")
for pname in ("mu", "huzinaga"):
    res = getattr(drv, pname)
    if res is None:
        continue
    sq = res["second_quantised"]
    size = f"{sq.nbytes / 1e9:.2f} GB spatial" if fmt == "spatial" else str(sq[2].shape)
    print(f"{pname}: e_rhf {res['e_rhf']:.10f} classical {res['classical_energy']:.10f} converged {bool(res['scf'].converged)} "
          f"cycles {getattr(res['scf'], 'cycles', None)}; n_mo {res['scf'].mo_coeff.shape[-1]}; hamiltonian {size}")
ls = drv.localized_system
print("active occupied (alpha, beta):", [len(x) for x in ls.active_mo_inds] if hasattr(ls, "active_mo_inds") else None)
