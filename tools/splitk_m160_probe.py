"""Lead for the beam-5 decode projections: the split-K form (grid = feature tiles x K slices, fp32 partial slabs) at
M = 160 live rows against the complete-sum kernels' times in the beam profile (FFN-wo 24.9 us, QKV 15.9 us).  Kernel time only
(host launch rate hides it otherwise): run under rocprofv3 --kernel-trace --stats.   python tools/splitk_m160_probe.py"""
import ctypes as C
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from markushgrapher_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 160
for name, N, K in [("wo", 1024, 4096), ("qkv", 3072, 1024), ("wi", 4096, 2048)]:
    wbytes = N * K * 2
    ncopy = int(700e6 // wbytes)
    W = torch.randint(-3000, 3000, (ncopy, wbytes // 2), dtype=torch.int16, device=dev)
    X = torch.randint(-3000, 3000, (M * K,), dtype=torch.int16, device=dev)
    for KS in (2, 4, 8, 16):
        if KS > K // 64:
            continue
        Pb = torch.empty((KS, M, N), dtype=torch.float32, device=dev)
        for i in range(40):
            lib.mgk_gemm_splitk(st(), P(X), P(W[i % ncopy]), P(Pb), M, N, K, N, C.c_size_t(M * N), KS)
        torch.cuda.synchronize()
        print(name, N, K, "KS", KS, "grid", N // 32 * KS, flush=True)
