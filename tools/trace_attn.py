"""Phase stamps of the decoder cross-attention kernel (shader clock, per wave, relative to the wave's own start)."""
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
NAMES = ["wave start", "first K/V round issued", "first round consumed", "stream done", "wave merged (shuffles)",
         "LDS exchange + barrier", "wave end"]


def main():
    B, H, cap, lens = 32, 16, 1280, 1090
    ncopy = 5
    Kc = torch.randint(-3000, 3000, (ncopy, B, H, cap, 64), dtype=torch.int16, device=dev)
    Vc = torch.randint(-3000, 3000, (ncopy, B, H, cap, 64), dtype=torch.int16, device=dev)
    q = torch.randint(-3000, 3000, (B, H, 64), dtype=torch.int16, device=dev)
    ctx = torch.empty((B * H * 64,), dtype=torch.int16, device=dev)
    ln = torch.full((B,), lens, dtype=torch.int32, device=dev)
    nblk, nw = B * H, 8
    trace = torch.zeros((nblk * nw * 8,), dtype=torch.int64, device=dev)
    acc = []
    for i in range(30):
        lib.mgk_attention_step_trace(st(), P(q), P(Kc[i % ncopy]), P(Vc[i % ncopy]), P(ctx), B, H, cap, P(ln), P(trace))
        torch.cuda.synchronize()
        if i >= 6:
            t = trace.cpu().numpy().reshape(nblk, nw, 8).astype(np.float64)
            acc.append(t - t[:, :, 0:1])
    a = np.stack(acc)
    for k, n in enumerate(NAMES):
        v = a[:, :, 0, k] if k == 6 else a[..., k]
        print(f"{k} {n:32s} mean {v.mean():9.1f}  p95 {np.percentile(v, 95):9.1f} ticks")


if __name__ == "__main__":
    main()
