"""BASELINE configs[3] in FULL on one GPU: the three spin blocks (aa|aa), (aa|bb), (bb|bb) of the active-space tensor at
N_AO = 2000, n_act = 128 from generated (pq|rs) -- every r, not a sample -- and one full generated-integral J/K build.
Checks that use the whole r-sum (p, q and r, s are treated differently by the transform, so these are not built in):
(ij|kl) = (kl|ij) on the same-spin blocks, and the contraction identity sum_kl (ij|kl) X_kl = (C^T J[C X C^T] C)_ij
against the J of the independent streaming J/K kernel (nbx_jk_synth_sym).  Writes a JSON record.

    python tools/n2000_full.py [out.json] [r-slabs per call]
"""
import json
import sys
import time

import numpy as np
import torch

from nbed_amd import synth
from nbed_amd.backend import HipBackend

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/n2000_full.json"
step = int(sys.argv[2]) if len(sys.argv) > 2 else 100
N, n = 2000, 128
be = HipBackend()
ca = be.asarray(np.ascontiguousarray(synth.sym_matrix(7, N)[:, :n]))
cb = be.asarray(np.ascontiguousarray(synth.sym_matrix(6, N)[:, :n]))
be.ao2mo_synth_pair(N, ca, ca, ca, ca, cb, cb, r0=1000, r1=1001)  # first touch of the workspaces
torch.cuda.synchronize()

aa = torch.zeros((n, n, n, n), dtype=torch.float64, device=be.device)
ab = torch.zeros_like(aa)
bb = torch.zeros_like(aa)
t0 = time.perf_counter()
for r0 in range(0, N, step):
    r1 = min(N, r0 + step)
    p_aa, p_ab = be.ao2mo_synth_pair(N, ca, ca, ca, ca, cb, cb, r0=r0, r1=r1)
    aa += p_aa
    ab += p_ab
    bb += be.ao2mo_synth(N, cb, cb, cb, cb, r0=r0, r1=r1)
    torch.cuda.synchronize()
    print(f"r < {r1}: {time.perf_counter() - t0:.1f} s", flush=True)
wall = time.perf_counter() - t0


def flops_block(N, n):
    return 2.0 * n * N**4 + 2.0 * n**2 * N**3 + 2.0 * n**3 * N**2 + 2.0 * n**4 * N


def sym_defects(t, same_spin):
    s = float(t.abs().max())
    d = {"ij_ji": float((t - t.transpose(0, 1)).abs().max()) / s, "kl_lk": float((t - t.transpose(2, 3)).abs().max()) / s}
    if same_spin:
        d["ijkl_klij"] = float((t - t.permute(2, 3, 0, 1)).abs().max()) / s
    return d


rec = {"workload": f"N_AO={N}, n_act={n}: (aa|aa), (aa|bb), (bb|bb) from generated (pq|rs), every r in [0, {N}), one GPU, "
                   f"{step} r-slabs per call (partial tensors summed on the device)",
       "wall_s_three_blocks": wall,
       "reference_count_tflops": 3 * flops_block(N, n) / wall / 1e12,
       "scale_max_abs": {"aa": float(aa.abs().max()), "ab": float(ab.abs().max()), "bb": float(bb.abs().max())},
       "symmetry_defects_relative": {"aa": sym_defects(aa, True), "ab": sym_defects(ab, False), "bb": sym_defects(bb, True)}}
ab_ba = float((ab.permute(2, 3, 0, 1) - ab).abs().max())  # (aa|bb) is NOT symmetric under the pair swap: a control
rec["control_ab_pair_swap_is_not_a_symmetry"] = ab_ba / float(ab.abs().max())

# contraction identity against the independent J/K kernel on generated integrals (full build: every row p)
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(n, n, generator=g, dtype=torch.float64)
x = (x + x.T).to(be.device)
t1 = time.perf_counter()
jrows = 250
for name, c_left, tensor, c_right in (("aa", ca, aa, ca), ("ab", ca, ab, cb), ("bb", cb, bb, cb)):
    dmx = (c_right @ x @ c_right.T).contiguous()
    jfull = torch.zeros((N, N), dtype=torch.float64, device=be.device)
    tj = time.perf_counter()
    for p0 in range(0, N, jrows):
        jfull += be.jk_synth_sym(N, dmx.reshape(1, N, N), p0, min(N, p0 + jrows))[0]
        torch.cuda.synchronize()
    dtj = time.perf_counter() - tj
    lhs = torch.einsum("ijkl,kl->ij", tensor, x)
    rhs = c_left.T @ jfull @ c_left
    err = float((lhs - rhs).abs().max())
    rec.setdefault("contraction_identity", {})[name] = {"max_abs_diff": err, "scale": float(rhs.abs().max()),
                                                        "relative": err / float(rhs.abs().max()), "full_j_build_s": dtj}
    print(f"identity {name}: {err:.3e} of {float(rhs.abs().max()):.3e}; J build {dtj:.1f} s", flush=True)
rec["note"] = ("wall includes the host loop and the device adds of the partial tensors; the contraction identity compares "
               "einsum('ijkl,kl->ij') of the transformed block with C^T J[C X C^T] C, J from nbx_jk_synth_sym on the same "
               "generated integrals (one density: J only)")
with open(out_path, "w") as fh:
    json.dump(rec, fh, indent=1)
print(json.dumps(rec))
