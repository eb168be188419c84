"""Timing of the weight-absorbed cross-attention launches alone (k_xattn.hip through mgk_xattn): rows decode rows, each reading its own
image's `keys` attended encoder states.  Per-kernel durations: run under `rocprofv3 --kernel-trace --stats`.
  python tools/xattn_bench.py [rows] [keys] [nsplit] [nstg] [iters] [cap]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from markushgrapher_amd import _lib  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    keys = int(sys.argv[2]) if len(sys.argv) > 2 else 1047
    nsplit = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    nstg = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 50
    H, d = 16, 1024
    cap = int(sys.argv[6]) if len(sys.argv) > 6 else (keys + 63) // 64 * 64
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(1)
    q = (torch.randn(rows, H, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    wkv = (torch.randn(2 * H * 64, d, generator=g) / 32).to(torch.bfloat16).float().to(dev)
    enc = torch.randn(rows, cap, d, generator=g).to(torch.bfloat16).to(dev)
    lens = torch.from_numpy(np.clip(keys + np.random.RandomState(2).randint(-20, 21, rows), 1, cap).astype(np.int32)).to(dev)
    wk = torch.zeros(H * d * 64, dtype=torch.bfloat16, device=dev)
    wv = torch.zeros_like(wk)
    qx = torch.zeros(rows * H * d, dtype=torch.bfloat16, device=dev)
    part = torch.zeros(rows * nsplit * H * d, dtype=torch.float32, device=dev)
    ml = torch.zeros(rows * nsplit * H * 2, dtype=torch.float32, device=dev)
    ctx = torch.zeros(((rows + 31) // 32 * 32) * H * 64, dtype=torch.bfloat16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())

    def call():
        rc = lib.mgk_xattn(st, p(q), p(wkv), p(enc), p(lens), None, rows, H, d, cap, nsplit, nstg, p(wk), p(wv), p(qx), p(part), p(ml), p(ctx))
        assert rc == 0, rc
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        call()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / iters
    moved = float(lens.sum().item()) * d * 2
    print(f"rows {rows} keys {keys} cap {cap} nsplit {nsplit} nstg {nstg}: {dt * 1e6:.1f} us per layer-step (4 launches incl. weight re-ordering), "
          f"stream bytes {moved / 1e6:.1f} MB -> {moved / dt / 1e12:.2f} TB/s if it were the stream alone")


if __name__ == "__main__":
    main()
