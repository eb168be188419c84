"""Probe of decode-GEMM variants under rocprofv3 (kernel durations, not host-side timing): every case is launched 100
times over rotating weight copies (HBM-cold).  tools/kprobe_report.py prints the per-case averages.
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


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


CASES = [("wo2 resid full", 0), ("wo2 resid no W loads", 1), ("wo2 resid no X loads", 2), ("wo2 resid empty", 3),
         ("wo2 resid no epilogue", 4), ("wo2 resid with rs", 10)]


def main():
    M, N, K = 32, 1024, 4096
    wbytes = N * K * 2
    ncopy = 64
    W = torch.randint(-3000, 3000, (ncopy, wbytes // 2), dtype=torch.int16, device=dev)
    X = torch.randint(-3000, 3000, (M * K,), dtype=torch.int16, device=dev)
    out_pk = torch.empty((M * N,), dtype=torch.int16, device=dev)
    h = torch.zeros((M, N), dtype=torch.float32, device=dev)
    gain = torch.ones((N,), dtype=torch.float32, device=dev)
    part = torch.ones((M * (N // 8),), dtype=torch.float32, device=dev)
    part_in = torch.ones((M * (N // 8),), dtype=torch.float32, device=dev)
    for name, dbg in CASES:
        os.environ["MG_KPROBE_DBG"] = str(dbg if dbg < 10 else 0)
        torch.cuda.synchronize()
        for i in range(100):
            rs = (P(part_in), N // 8, C.c_float(1.0 / N), C.c_float(1e-6)) if dbg == 10 else (None, 0, C.c_float(0.0), C.c_float(0.0))
            rc = lib.mgk_gemm_resid(st(), P(X), P(W[i % ncopy]), P(h), P(gain), C.c_float(1.0), P(out_pk), P(part), M, N, K, *rs)
            assert rc == 0, (name, rc)
        torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
