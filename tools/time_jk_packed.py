"""nbx_jk_packed against nbx_jk_dense_sym (itself parity-tested against the oracle): max
difference at several N / ndm / slabs, and the timing of both at the benchmark size."""
import sys
import time

import torch

from nbed_amd.backend import HipBackend


def main():
    be = HipBackend()
    torch.manual_seed(7)
    sizes = [int(a) for a in sys.argv[1:]] or [24, 72, 128, 148, 192, 256]
    for n in sizes:
        if not be.jk_packed_supported(n):
            print(f"N={n}: not covered by nbx_jk_packed", flush=True)
            continue
        eri = be.synth_eri(n)
        packed = be.eri_pack(eri, n)
        for ndm in (2, 1):
            dm = torch.randn(ndm, n, n, dtype=torch.float64, device=be.device)
            dm = dm + dm.transpose(1, 2)
            ref = be.jk_sym(eri, dm).clone()
            got = be.jk_packed(packed, dm).clone()
            err = (got - ref).abs().max().item()
            scale = ref.abs().max().item()
            # two slabs add up
            h = n // 3
            a = be.jk_packed(be.eri_pack(eri[:h], n, 0, h), dm, 0, h).clone()
            b = be.jk_packed(be.eri_pack(eri[h:], n, h, n), dm, h, n).clone()
            err2 = (a + b - ref).abs().max().item()
            print(f"N={n} ndm={ndm}: max|packed - sym| = {err:.3e} (scale {scale:.2e}); slabs {err2:.3e}", flush=True)
        dm = torch.randn(2, n, n, dtype=torch.float64, device=be.device)
        dm = dm + dm.transpose(1, 2)
        for name, fn in (("sym", lambda: be.jk_sym(eri, dm)), ("packed", lambda: be.jk_packed(packed, dm))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            nbytes = packed.numel() * 8 if name == "packed" else 4 * n ** 3 * (n + 1)
            print(f"  {name}: {dt * 1e3:.3f} ms/call, {nbytes / dt / 1e12:.2f} TB/s on the bytes it reads", flush=True)
        del eri, packed
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
