"""Probe of candidate decode-GEMM shapes under rocprofv3 (kernel durations, not host-side timing): every case is
launched 100 times over rotating weight copies (HBM-cold); read the result with tools/rocpd_stats.py --by-grid.
   cd /tmp && rocprofv3 --kernel-trace -d $OUT -o probe -- python tools/kprobe.py
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from markushgrapher_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
EPI_F32_STORE, EPI_F32_RESID, EPI_PK_RELU, EPI_PK = 0, 1, 2, 3


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def main():
    M = 32
    cases = [("wi      rows relu", 4096, 1024, "rows"), ("wi'     rows relu", 4096, 2048, "rows"), ("xq'     rows pk", 1024, 2048, "rows"),
             ("xq      rows pk", 1024, 1024, "rows"), ("o       resid", 1024, 1024, "resid"), ("o K2048 resid", 1024, 2048, "resid"),
             ("wo2     resid", 1024, 4096, "resid")]
    for name, N, K, kind in cases:
        wbytes = N * K * 2
        ncopy = max(2, min(128, int(600e6 // wbytes)))
        W = torch.randint(-3000, 3000, (ncopy, wbytes // 2), dtype=torch.int16, device=dev)
        X = torch.randint(-3000, 3000, (M * K,), dtype=torch.int16, device=dev)
        out_pk = torch.empty((M * N,), dtype=torch.int16, device=dev)
        h = torch.zeros((M, N), dtype=torch.float32, device=dev)
        gain = torch.ones((N,), dtype=torch.float32, device=dev)
        part = torch.zeros((M * (N // 8),), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        for i in range(100):
            if kind == "rows":
                epi = EPI_PK_RELU if "relu" in name else EPI_PK
                rc = lib.mgk_gemm(st(), 1, epi, P(X), P(W[i % ncopy]), M, N, K, None, 0, None, P(out_pk))
            else:
                rc = lib.mgk_gemm_resid(st(), P(X), P(W[i % ncopy]), P(h), P(gain), C.c_float(1.0), P(out_pk), P(part), M, N, K, None, 0,
                                        C.c_float(0.0), C.c_float(0.0))
            assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        print(name, "grid-id N", N, "K", K)


if __name__ == "__main__":
    main()
