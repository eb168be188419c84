"""Does a decode projection start faster when its weights are already on the chip?  Times gemm_rows (M = 32 rows, the QKV / FFN
shapes of the decode step) launched back to back with (cold) a different copy of the weights per launch, 700 MB apart, so
neither L2 nor the 256 MiB Infinity Cache holds them, and (warm) the same copy every launch.  The difference bounds what a
prefetch of the NEXT kernel's weights (issued from inside the previous kernel) could buy.
Host-side launch rate hides kernel time here (eager launches): run one mode per process under rocprofv3 --kernel-trace --stats
and compare the kernels' own durations:  python tools/warm_probe.py cold|warm"""
import ctypes as C
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from markushgrapher_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=400, warm=40):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M = 32
for name, N, K, epi in [("qkv", 3072, 1024, 3), ("wi", 4096, 1024, 2), ("o", 1024, 1024, 3), ("wo-shaped", 1024, 4096, 3)]:
    wbytes = N * K * 2
    ncopy = int(900e6 // wbytes)
    W = torch.randint(-3000, 3000, (ncopy, wbytes // 2), dtype=torch.int16, device=dev)
    X = torch.randint(-3000, 3000, (M * K,), dtype=torch.int16, device=dev)
    out = torch.empty((M * N,), dtype=torch.int16, device=dev)
    res = {}
    modes = (("cold", ncopy), ("two copies", 2), ("warm", 1))
    if len(sys.argv) > 1:      # one mode per process: kernel durations then come from `rocprofv3 --kernel-trace` of that process
        modes = tuple(m for m in modes if m[0].split()[0] == sys.argv[1])
    for mode, nc in modes:
        def f(i):
            lib.mgk_gemm(st(), 1, epi, P(X), P(W[i % nc]), M, N, K, None, N, None, P(out))
        res[mode] = timeit(f)
    print(f"gemm_rows {name:10s} N={N:5d} K={K:5d}: " + "  ".join(f"{k} {v:6.2f} us" for k, v in res.items()), flush=True)
