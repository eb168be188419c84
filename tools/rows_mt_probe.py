"""Times the residual projection of the decode step over several row tiles: one-workgroup form against the K-slab form
(gemm_rows_resid_mt_kernel), alone on the GPU.   python tools/rows_mt_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from markushgrapher_amd import _lib  # noqa: E402

if __name__ == "__main__":
    lib = _lib.load()
    lib.mgk_gemm_resid_mt.argtypes = [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M, N, K in ((160, 1024, 4096), (160, 1024, 1024), (64, 1024, 4096), (256, 1024, 4096)):
        Mp = (M + 31) // 32 * 32
        X = torch.randint(-2000, 2000, (Mp * K,), dtype=torch.int16, device="cuda")
        W = torch.randint(-2000, 2000, (N * K,), dtype=torch.int16, device="cuda")
        h = torch.zeros(Mp * N, device="cuda"); g = torch.ones(N, device="cuda")
        xp = torch.zeros(Mp * N, dtype=torch.int16, device="cuda"); part = torch.zeros(Mp * (N // 8), device="cuda")
        kpart = torch.zeros(16 * Mp * N, device="cuda"); ticket = torch.zeros(N // 32, dtype=torch.int32, device="cuda")
        out = []
        for mode in (0, 1, 2):
            lib.mgk_set_rows_mt(mode)
            def run():
                lib.mgk_gemm_resid_mt(st, X.data_ptr(), W.data_ptr(), h.data_ptr(), g.data_ptr(), C.c_float(1.0), xp.data_ptr(), part.data_ptr(), M, N, K,
                                      None, 0, C.c_float(0), C.c_float(0), 8, kpart.data_ptr(), ticket.data_ptr())
            for _ in range(20):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(400):
                run()
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 400 * 1e3)
        lib.mgk_set_rows_mt(0)
        print(f"M={M} N={N} K={K}: one-workgroup form {out[0]:.2f} us, K-slab form {out[1]:.2f} us, K-slab in two launches {out[2]:.2f} us per projection (back-to-back, weights {N * K * 2 / 1e6:.1f} MB)")
