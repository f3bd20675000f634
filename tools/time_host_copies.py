import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
n = 2 * 1024**3  # 16 GB of doubles
a = torch.rand(n, dtype=torch.float64, device='cuda')
torch.cuda.synchronize()
t = time.perf_counter(); h = a.cpu().numpy(); dt = time.perf_counter() - t
print(f"torch .cpu(): {n*8/dt/1e9:.1f} GB/s ({dt:.2f} s)", flush=True)
del h
def pipelined(a, chunk=32*1024*1024, nbuf=3, threads=4):
    out = np.empty(a.numel(), dtype=np.float64)
    bufs = [torch.empty(chunk, dtype=torch.float64, pin_memory=True) for _ in range(nbuf)]
    evs = [torch.cuda.Event() for _ in range(nbuf)]
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    pool = ThreadPoolExecutor(threads)
    nch = (a.numel() + chunk - 1) // chunk
    pending = [None] * nbuf
    def drain(k):
        b = k % nbuf
        lo = k * chunk; hi = min(a.numel(), lo + chunk)
        evs[b].synchronize()
        src = bufs[b].numpy()[: hi - lo]
        step = (hi - lo + threads - 1) // threads
        return [pool.submit(np.copyto, out[lo + i*step: min(hi, lo + (i+1)*step)], src[i*step: min(hi-lo, (i+1)*step)]) for i in range(threads)]
    for k in range(nch + nbuf - 1):
        if k < nch:
            b = k % nbuf
            if pending[b] is not None:
                for f in pending[b]: f.result()
            lo = k * chunk; hi = min(a.numel(), lo + chunk)
            with torch.cuda.stream(stream):
                bufs[b][: hi - lo].copy_(a[lo:hi], non_blocking=True)
                evs[b].record()
        j = k - (nbuf - 1)
        if j >= 0:
            pending[j % nbuf] = drain(j)
    for p in pending:
        if p is not None:
            for f in p: f.result()
    pool.shutdown()
    return out
for th in (2, 4, 8):
    t = time.perf_counter(); h = pipelined(a, threads=th); dt = time.perf_counter() - t
    ok = float(h[12345]) == float(a[12345]) and float(h[-7]) == float(a[-7])
    print(f"pipelined threads={th}: {n*8/dt/1e9:.1f} GB/s ({dt:.2f} s) ok={ok}", flush=True)
    del h
from nbed_amd.backend import HipBackend
be = HipBackend()
hh = np.random.rand(n // 2)
t = time.perf_counter(); d = torch.from_numpy(hh).to('cuda'); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"torch .to(cuda): {hh.nbytes/dt/1e9:.1f} GB/s")
t = time.perf_counter(); d2 = be.asarray(hh); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"pipelined upload: {hh.nbytes/dt/1e9:.1f} GB/s ok={bool((d2 == d).all())}")
