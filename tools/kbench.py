"""Isolated timing of the decode-step kernels on a real MI355X (HBM-cold: every launch reads a different copy of its
weights / K-V streams so the 256 MiB Infinity Cache cannot serve them).  Prints one line per case:
   name  avg_us  GB/s(algorithmic)
Usage (on the GPU box):  python tools/kbench.py [gemm] [attn] [norm]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from tools import _toolslib as _lib  # noqa: E402  (tools build: trace kernels / what-if variants)

lib = _lib.load()
dev = torch.device("cuda:0")


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=200, warm=20):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def bench_gemm():
    M = 32
    for name, N, K in [("qkv", 3072, 1024), ("o/xq/xo", 1024, 1024), ("wi", 4096, 1024), ("wo2", 1024, 4096), ("lm_head", 33201, 1024)]:
        wbytes = ((N + 31) // 32 * 32) * K * 2
        ncopy = max(2, min(256, int(700e6 // wbytes)))
        W = torch.randint(-3000, 3000, (ncopy, wbytes // 2), dtype=torch.int16, device=dev)
        X = torch.randint(-3000, 3000, (M * K,), dtype=torch.int16, device=dev)
        ks_auto = lib.mgk_splitk_factor(N, K)
        for KS in sorted(set([1, 2, 4, 8, 16, ks_auto])):
            if KS > K // 64:
                continue
            ldp = (N + 31) // 32 * 32
            Pb = torch.empty((KS, 32, ldp), dtype=torch.float32, device=dev)

            def f(i):
                lib.mgk_gemm_splitk(stream(), P(X), P(W[i % ncopy]), P(Pb), M, N, K, ldp, C.c_size_t(32 * ldp), KS)
            us = timeit(f)
            print(f"gemm_splitk {name:8s} N={N:5d} K={K:4d} KS={KS:2d}{'*' if KS == ks_auto else ' '} blocks={(N + 31) // 32 * KS:5d} "
                  f"{us:7.2f} us  {wbytes / us / 1e3:7.1f} GB/s")


def bench_ppexp():
    """What-if variants of the ping-pong GEMM (MG_PP_EXP, tools build, wrong results): where the tile time goes."""
    M = 40960
    xp = os.environ.get("MG_PP_EXP", "0")
    for name, N, K, epi in [("qkv", 3072, 1024, 3), ("o", 1024, 1024, 1), ("wo", 1024, 4096, 1)]:
        X = torch.randint(-3000, 3000, (M * K,), dtype=torch.int16, device=dev)
        W = torch.randint(-3000, 3000, (N * K,), dtype=torch.int16, device=dev)
        out_pk = torch.empty((M * N,), dtype=torch.int16, device=dev)
        out_f = torch.zeros((M, N), dtype=torch.float32, device=dev) if epi == 1 else None
        for variant in (5, 6):
            lib.mgk_gemm_set_variant(variant)

            def f(i):
                lib.mgk_gemm(stream(), 0, epi, P(X), P(W), M, N, K, P(out_f), N, None, P(out_pk))
            us = timeit(f, iters=20, warm=3)
            print(f"pp exp {xp:>2s} {name:4s} variant {variant}: {us:8.1f} us  {2.0 * M * N * K / us / 1e9:7.3f} PFLOP/s", flush=True)
    lib.mgk_gemm_set_variant(3)


def bench_encgemm():
    """Encoder-sized GEMMs (M = 32 images x 1280 positions): 256x128 three-stage kernel (variant 1) vs 256x256 (variant 2)."""
    M = 40960
    for name, N, K, epi in [("qkv", 3072, 1024, 3), ("o", 1024, 1024, 1), ("wi", 4096, 1024, 2), ("wo", 1024, 4096, 1), ("xkv", 2048, 1024, 3)]:
        X = torch.randint(-3000, 3000, (M * K,), dtype=torch.int16, device=dev)
        W = torch.randint(-3000, 3000, (N * K,), dtype=torch.int16, device=dev)
        out_pk = torch.empty((M * N,), dtype=torch.int16, device=dev)
        out_f = torch.zeros((M, N), dtype=torch.float32, device=dev) if epi == 1 else None
        ref = None
        for variant in (4, 5, 6):
            lib.mgk_gemm_set_variant(variant)

            def f(i):
                lib.mgk_gemm(stream(), 0, epi, P(X), P(W), M, N, K, P(out_f), N, None, P(out_pk))
            us = timeit(f, iters=20, warm=3)
            # same bits as the two-stage kernel (same K order per accumulator): one run from a zeroed output each, repeated 3x (race screen)
            same = []
            for rep in range(3 if variant != 4 else 1):
                out_pk.zero_()
                if out_f is not None:
                    out_f.zero_()
                f(0)
                torch.cuda.synchronize()
                got = (out_f.view(torch.int32) if epi == 1 else out_pk).clone()          # bit patterns (random operands overflow to inf / nan)
                if variant == 4:
                    ref = got
                else:
                    same.append(bool(torch.equal(got, ref)))
            print(f"enc gemm {name:4s} M={M} N={N:5d} K={K:4d} variant {variant}: {us:8.1f} us  {2.0 * M * N * K / us / 1e9:7.3f} PFLOP/s"
                  + ("" if variant == 4 else f"  bits equal to variant 4: {same}"), flush=True)
    lib.mgk_gemm_set_variant(3)


def bench_encnorm():
    """The encoder's residual projections as the encoder runs them (EPI_RESID_NORM: tiled fp32 h += X W^T, bf16(h * gain), partial sums
    of squares): two-stage kernel (variant 4) vs the persistent ping-pong kernel (5 / 6); bits of h, the packed output and the partial sums compared."""
    M, d = 40960, 1024
    lib.mgk_gemm_norm.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p] * 4 + \
                                 [C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float]
    for name, K in [("o", 1024), ("wo", 4096)]:
        X = (torch.randn((M * K,), device=dev) * 0.5).to(torch.bfloat16).view(torch.int16)
        W = (torch.randn((d * K,), device=dev) * 0.05).to(torch.bfloat16).view(torch.int16)
        h0 = torch.randn((M * d,), device=dev)
        g = 1 + 0.2 * torch.randn((d,), device=dev)
        h = h0.clone()
        xo = torch.empty((M * d,), dtype=torch.int16, device=dev)
        part = torch.zeros((M, d // 64), dtype=torch.float32, device=dev)
        ref = None
        for variant in (4, 5, 6):
            lib.mgk_gemm_set_variant(variant)

            def f(i):
                lib.mgk_gemm_norm(stream(), 5, P(X), P(W), M, d, K, P(h), P(g), P(xo), P(part), d // 64, None, 0, 0.0, 0.0)
            us = timeit(f, iters=20, warm=3)
            same = []
            for rep in range(3 if variant != 4 else 1):
                h.copy_(h0); xo.zero_(); part.zero_()
                f(0)
                torch.cuda.synchronize()
                got = (h.view(torch.int32).clone(), xo.clone(), part.view(torch.int32).clone())
                if variant == 4:
                    ref = got
                else:
                    same.append(all(bool(torch.equal(a_, b_)) for a_, b_ in zip(got, ref)))
            print(f"enc norm-gemm {name:3s} M={M} N={d} K={K:4d} variant {variant}: {us:8.1f} us  {2.0 * M * d * K / us / 1e9:7.3f} PFLOP/s"
                  + ("" if variant == 4 else f"  bits equal to variant 4: {same}"), flush=True)
    lib.mgk_gemm_set_variant(3)


def bench_attn():
    B, H, cap = 32, 16, 1280
    for name, group, lens in [("cross", 1, 1072), ("cross-warm", 1, 1072), ("cross-full", 1, 1280), ("self t=128", 0, 129), ("self t=400", 0, 401)]:
        is_self = group == 0
        rows = B
        ncopy = (1 if "warm" in name else 5) if not is_self else 24
        caps = cap if not is_self else 512
        Kc = torch.randint(-3000, 3000, (ncopy, rows, H, caps, 64), dtype=torch.int16, device=dev)
        Vc = torch.randint(-3000, 3000, (ncopy, rows, H, caps, 64), dtype=torch.int16, device=dev)
        q = torch.randint(-3000, 3000, (rows, H, 64), dtype=torch.int16, device=dev)
        ctx = torch.empty((rows * H * 64,), dtype=torch.int16, device=dev)
        ln = torch.full((rows,), lens, dtype=torch.int32, device=dev)
        bias = torch.zeros((512, H), dtype=torch.float32, device=dev)

        def f(i):
            c = i % ncopy
            lib.mgk_attention_step(stream(), P(q), P(Kc[c]), P(Vc[c]), P(ctx), rows, H, 1, caps,
                                   None if is_self else P(ln), lens, P(bias) if is_self else None, None, lens - 1)
        us = timeit(f)
        nbytes = rows * H * lens * 64 * 2 * 2
        print(f"attn_step {name:12s} keys={lens:5d} {us:7.2f} us  {nbytes / us / 1e3:7.1f} GB/s")


def bench_norm():
    M, d = 32, 1024
    for KS in (4, 8, 10):
        h = torch.randn((M, d), device=dev)
        Pb = torch.randn((KS, M, d), device=dev)
        g = torch.ones((d,), device=dev)
        xp = torch.empty((M * d,), dtype=torch.int16, device=dev)

        def f(i):
            lib.mgk_add_norm_pack(stream(), P(h), P(Pb), KS, d, C.c_size_t(M * d), P(g), P(xp), M, d, C.c_float(1e-6), C.c_float(1.0))
        print(f"add_norm_pack KS={KS:2d} {timeit(f):7.2f} us")
    Pb = torch.randn((3, M, 4096), device=dev)
    y = torch.empty((M * 4096,), dtype=torch.int16, device=dev)

    def f2(i):
        lib.mgk_relu_pack(stream(), P(Pb), 3, 4096, C.c_size_t(M * 4096), P(y), M, 4096)
    print(f"relu_pack KS=3 {timeit(f2):7.2f} us")


if __name__ == "__main__":
    lib.mgk_gemm_splitk.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_size_t, C.c_int]
    what = sys.argv[1:] or ["gemm", "attn", "norm"]
    if "gemm" in what:
        bench_gemm()
    if "encgemm" in sys.argv:
        bench_encgemm()
    if "encnorm" in sys.argv:
        bench_encnorm()
    if "ppexp" in sys.argv:
        bench_ppexp()
    if "attn" in what:
        bench_attn()
    if "norm" in what:
        bench_norm()


def bench_reorder():
    """configs[2]: KV-cache beam reorder. Physical index_select copy (reference behaviour) vs the ancestor-table form."""
    layers, B, K, H, cap = 24, 32, 5, 16, 512
    rows = B * K
    for t in (32, 128):
        src = torch.randint(-3000, 3000, (layers, 2, rows, H, cap, 64), dtype=torch.int16, device=dev)
        dst = torch.empty_like(src)
        idx = (torch.arange(rows, device=dev, dtype=torch.int32) // K) * K + torch.randint(0, K, (rows,), device=dev, dtype=torch.int32)

        def f(i):
            lib.mg_beam_reorder(stream(), P(src), P(dst), P(idx), layers, rows, H, cap, t)
        us = timeit(f, iters=20, warm=3)
        nbytes = layers * 2 * rows * H * t * 64 * 2 * 2
        print(f"beam_reorder physical t={t:4d}: {us:9.1f} us  {nbytes / us / 1e3:7.1f} GB/s (read+write {nbytes / 1e6:.0f} MB); "
              f"ancestor table moves {t * rows * 4 / 1e3:.0f} KB instead")


if "reorder" in sys.argv[1:]:
    bench_reorder()
