"""Where do the ~8.7 us of the 32-row FFN-wo residual projection go?  Launches the phase-stamped copy of the kernel on
HBM-cold weights and prints, per phase, the mean / p95 over all waves of the shader-clock time since that wave's own start."""
import ctypes as C
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from tools import _toolslib as _lib  # noqa: E402  (tools build: trace kernels / what-if variants)

lib = _lib.load()
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
NAMES = ["wave start", "weight loads issued", "row scales ready", "MFMAs done (loads landed)", "LDS exchange + barrier",
         "epilogue stores issued (finishing wave)", "wave end"]


def main():
    N, K, M = 1024, 4096, 32
    ncopy = 48
    W = torch.randint(-3000, 3000, (ncopy, N * K), dtype=torch.int16, device=dev)
    X = torch.randint(-3000, 3000, (M * K,), dtype=torch.int16, device=dev)
    h = torch.zeros((M, N), dtype=torch.float32, device=dev)
    gain = torch.ones((N,), dtype=torch.float32, device=dev)
    xpk = torch.empty((M * N,), dtype=torch.int16, device=dev)
    part = torch.ones((M * (N // 8),), dtype=torch.float32, device=dev)
    rsp = torch.ones((M * (N // 8),), dtype=torch.float32, device=dev)
    nblk, nw = N // 8, 16
    trace = torch.zeros((nblk * nw * 8,), dtype=torch.int64, device=dev)
    acc = []
    for i in range(40):
        lib.mgk_gemm_resid_trace(st(), P(X), P(W[i % ncopy]), P(h), P(gain), P(xpk), P(part), N, K, P(rsp), P(trace))
        torch.cuda.synchronize()
        if i >= 8:
            t = trace.cpu().numpy().reshape(nblk, nw, 8).astype(np.float64)
            acc.append(t - t[:, :, 0:1])          # per-wave deltas (the counters of different XCDs have different bases)
    a = np.stack(acc)            # [rep][blk][wave][phase] in shader clocks
    freq_ghz = 0.1               # s_memtime ticks at 100 MHz on gfx9 (constant), convert below if it looks like that
    for k, n in enumerate(NAMES):
        if k == 5:
            v = a[:, :, 0, k]      # finishing wave = wave 0 (one row tile)
        else:
            v = a[..., k]
        print(f"{k} {n:42s} mean {v.mean():9.1f}  p95 {np.percentile(v, 95):9.1f}  max {v.max(axis=tuple(range(1, v.ndim))).mean():9.1f} ticks")


if __name__ == "__main__":
    main()
