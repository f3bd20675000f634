"""Why does the warm-start refinement (csrc/eigh_refine.hip) accept or reject a cycle's matrix?  Re-plays its
iteration on the host with the DEVICE GEMMs (same rounding noise in S = V^T A V) for every eigensolve of a
step-by-step GpuUHF.kernel() run and prints omega, max|E|, the largest residual numerator and cluster coupling per
iteration: ``python tools/refine_diag.py [N_AO] [mu] [pairwise 0/1]``."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["NBED_CYCLE_CALL"] = "0"
import torch  # noqa: E402

from nbed_amd import synth  # noqa: E402
from nbed_amd.backend import HipBackend  # noqa: E402
from nbed_amd.scf import GpuUHF, Mole  # noqa: E402

be = HipBackend(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 148
mu = float(sys.argv[2]) if len(sys.argv) > 2 else 1e6
PAIR = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nocc, nenv = (33, 20) if N == 148 else (N // 6, N // 12)
pr = synth.problem(be, N, (nocc, nocc), nenv)
S, h = np.asarray(pr["S"]), np.asarray(pr["hcore"])
h3 = h[None] + mu * (S @ np.asarray(pr["D_env"]) @ S) + np.asarray(pr["V_emb"])
eri = be.synth_eri(N)
real_eigh = be.eigh
count = [0]


def replay(a, v0, iters=6):
    x = v0[0:1].clone()
    a1 = a[0:1]
    na = float(torch.linalg.norm(a1))
    out = []
    for it in range(iters):
        y = be.gemm(a1, x)
        sm = be.to_host(be.gemm(x, y, "T", "N"))[0]
        g = be.to_host(be.gemm(x, x, "T", "N"))[0]
        lam = np.diag(sm) / np.diag(g)
        r = np.eye(N) - g
        offm = sm - np.diag(np.diag(sm))
        d = lam[None, :] - lam[:, None]
        num = sm + lam[None, :] * r
        if PAIR:
            num_t = sm + lam[:, None] * r
            mn = np.minimum(np.abs(num), np.abs(num_t))
            np.fill_diagonal(mn, 0)
            om = 2 * (np.linalg.norm(mn) + np.maximum(np.abs(lam)[None, :], np.abs(lam)[:, None]) * np.linalg.norm(r))
        else:
            om = 2 * (np.linalg.norm(offm) + na * np.linalg.norm(r))
        far = np.abs(d) > om
        np.fill_diagonal(far, False)
        e = np.where(far, num / np.where(d == 0, 1, d), r / 2)
        nm = np.abs(np.where(far, num, 0))
        cm = np.abs(np.where(far, 0, sm))
        np.fill_diagonal(cm, 0)
        ncl = int((~far).sum() - N)
        out.append(f"om={np.max(om):.1e} e={np.abs(e).max():.1e} n={nm.max():.1e} c={cm.max():.1e} ncl={ncl} R={np.linalg.norm(r):.0e}")
        if np.abs(e).max() < 3e-8 and cm.max() <= 1e-14 * na:
            break
        x = be.gemm(x, be.asarray((np.eye(N) + e)[None]))
    return na, out


def eigh(a, check=False, v0=None, refine_iters=3):
    if v0 is not None:
        na, out = replay(a, v0)
        print(f"solve {count[0]}: ||A||={na:.2e} noise_thr={8 * 1.11e-16 * np.sqrt(N) * na:.1e} iters={len(out)}\n   " + "\n   ".join(out))
    count[0] += 1
    w, v = real_eigh(a, check=True, v0=v0, refine_iters=6)
    ah, vh, wh = be.to_host(a), be.to_host(v), be.to_host(w)
    msg = []
    for x in range(ah.shape[0]):
        res = ah[x] @ vh[x] - vh[x] * wh[x][None, :]
        col = np.abs(res).max(axis=0)
        wr = np.linalg.eigvalsh(ah[x])
        j = int(np.argmax(col))
        msg.append(f"spin {x}: max resid {col.max():.1e} (col {j}, w={wh[x][j]:.6g}), valence cols {col[:N - nenv].max():.1e}, "
                   f"orth {np.abs(vh[x].T @ vh[x] - np.eye(N)).max():.1e}, |w - lapack| {np.abs(wh[x] - wr).max():.1e}")
    print("   device status:", be.last_eigh_sweeps, "; ".join(msg))
    return w, v


be.eigh = eigh
mf = GpuUHF(Mole(N, pr["nelec"]), S, h, eri, backend=be)
mf.get_hcore = lambda *a: h3
mf.conv_tol, mf.max_cycle = 1e-10, 30
e = mf.kernel()
print(e, mf.cycles, mf.converged)
