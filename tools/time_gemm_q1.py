"""The quarter-1 product of the active-space transform as a bare GEMM: C (n x cols) = A^T (n x K) . B (K x cols) with
n = 128 and K = 148 (octane / 6-31G*) or 2000, timed with events on the stream.  NBX_LIB / NBX_GEMM_DMA=0 choose the
build and the MFMA form:   python tools/time_gemm_q1.py [K] [cols] [reps]"""
import os
import sys

import numpy as np
import torch

from nbed_amd.backend import HipBackend

be = HipBackend()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 148
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 148 * 148 * 74
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n = 128
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.rand((K, n), dtype=torch.float64, device="cuda", generator=g) - 0.5
b = torch.rand((K, cols), dtype=torch.float64, device="cuda", generator=g) - 0.5
c = be.empty((n, cols))
be.gemm(a, b, ta="T", out=c)
torch.cuda.synchronize()
ref = (a[:, :8].T @ b[:, :4096]).cpu().numpy()
if os.environ.get("NBX_GEMM_NOCHECK") != "1":  # (measurement builds compute something else on purpose)
    np.testing.assert_allclose(c[:8, :4096].cpu().numpy(), ref, rtol=0, atol=1e-12)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    be.gemm(a, b, ta="T", out=c)
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"K={K} cols={cols}: {ms:.3f} ms  {2.0 * n * K * cols / ms / 1e9:.2f} TFLOP/s  "
      f"(read+write {(K * cols + n * cols) * 8 / ms / 1e9:.2f} TB/s)")
