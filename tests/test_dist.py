"""world_size-2 gloo test of the multi-GPU path's host logic (sharding + all-gather of token ids)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from markushgrapher_amd.dist import ID_COLS, IdExchange, shard_bounds, sharded_generate


def fake_generate(input_ids, bbox=None, pixel_values=None, attention_mask=None, max_length=8, **kw):
    """deterministic stand-in for model.generate: row -> [0, sum(ids)%97, len, 1, pad...] with ragged lengths"""
    B = input_ids.shape[0]
    n = int(3 + int(input_ids[:, 0].max()) % 3)
    out = torch.zeros((B, n), dtype=torch.int64)
    out[:, 1] = input_ids.sum(1) % 97
    out[:, 2] = (input_ids != 0).sum(1)
    out[:, n - 1] = 1
    return out


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(3)
    batch = {"input_ids": torch.randint(0, 50, (B, 6), generator=g), "bbox": torch.rand((B, 6, 4), generator=g),
             "pixel_values": torch.rand((B, 3, 4, 4), generator=g), "attention_mask": None}
    out = sharded_generate(fake_generate, batch, max_length=8)
    # the bench's own use of the exchange (bench.py step()): 32 rows per rank, [32, 512] int32 + lengths, posted asynchronously
    # for two batches in a row (double buffering) before the first is waited for
    ex = IdExchange(32, torch.device("cpu"))
    mk = lambda step: (torch.arange(32 * (257 - step), dtype=torch.int64).reshape(32, 257 - step) % 33201) + 1000 * rank + step
    h0, h1 = ex.post(mk(0)), ex.post(mk(1))
    cp = lambda pair: (pair[0].numpy().copy(), pair[1].numpy().copy())     # the blocks are valid until their slot is posted again
    r0 = cp(ex.wait(h0))
    r1 = cp(ex.wait(h1))
    r2 = cp(ex.wait(ex.post(mk(2))))      # reuses the first slot
    q.put((rank, out.numpy(), [r0, r1, r2]))
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (1, 5, 32, 33, 256):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi = shard_bounds(n, w, r)
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def test_sharded_generate_world2_gloo():
    for B in (5, 8):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 29500 + (os.getpid() + B) % 1000
        procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = [q.get(timeout=120) for _ in range(2)]
        res = {r: o for r, o, _ in got}
        bench_payload = {r: p for r, _, p in got}
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        g = torch.Generator().manual_seed(3)
        ids = torch.randint(0, 50, (B, 6), generator=g)
        # single-process reference: each shard decoded on its own, padded to max_length
        ref = np.zeros((B, 8), np.int64)
        for r in range(2):
            lo, hi = shard_bounds(B, 2, r)
            o = fake_generate(ids[lo:hi]).numpy()
            ref[lo:hi, :o.shape[1]] = o
        assert np.array_equal(res[0], ref) and np.array_equal(res[1], ref)
        # bench payload: static [world * 32, 512] int32 + [world * 32] int32 on every rank, identical, rows in rank order
        for step in range(3):
            a, la = bench_payload[0][step]
            b, lb = bench_payload[1][step]
            assert a.dtype == np.int32 and a.shape == (64, ID_COLS) and la.shape == (64,) and la.dtype == np.int32
            assert np.array_equal(a, b) and np.array_equal(la, lb)
            t = 257 - step
            assert np.all(la == t)
            for r in range(2):
                want = (np.arange(32 * t, dtype=np.int64).reshape(32, t) % 33201) + 1000 * r + step
                assert np.array_equal(a[32 * r:32 * r + 32, :t], want) and np.all(a[32 * r:32 * r + 32, t:] == 0)


class _FakeMem:
    pass


class _FakeEngine:
    """engine stand-in for the host logic of the batches in flight: clone() gives a further context, generate() is fake_generate"""
    made = 0

    def __init__(self):
        self.mem = _FakeMem()
        _FakeEngine.made += 1
        self.closed = False

    def clone(self):
        return _FakeEngine()

    def close(self):
        self.closed = True

    def generate(self, ids):
        return fake_generate(ids)


def _worker_inflight(rank, world, port, q):
    """bench.py's loop shape at world 2: K batches over n contexts in flight per rank, every batch's ids posted to the exchange by the
    main thread in submission order (double-buffered), results identical on both ranks."""
    from markushgrapher_amd.inflight import InFlight
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fl = InFlight(_FakeEngine(), 3)
    ex = IdExchange(4, torch.device("cpu"))
    K = 7
    batches = [torch.full((4, 6), 10 * rank + k, dtype=torch.int64) + torch.arange(6)[None] for k in range(K)]
    futs = [fl.submit(lambda ctx, b: ctx.generate(b), b) for b in batches]
    handles, gathered = [], []
    for f in futs:
        handles.append(ex.post(f.result()))
        if len(handles) > 1:
            a, l = ex.wait(handles.pop(0))
            gathered.append((a.numpy().copy(), l.numpy().copy()))
    while handles:
        a, l = ex.wait(handles.pop(0))
        gathered.append((a.numpy().copy(), l.numpy().copy()))
    n_ctx = len(fl)
    fl.close()
    q.put((rank, gathered, n_ctx))
    dist.destroy_process_group()


def test_batches_in_flight_with_the_exchange_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 77) % 1000
    procs = [ctx.Process(target=_worker_inflight, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (g, n) for r, g, n in [q.get(timeout=120) for _ in range(2)]}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] == 3
    assert len(got[0][0]) == len(got[1][0]) == 7
    for k in range(7):
        (a0, l0), (a1, l1) = got[0][0][k], got[1][0][k]
        assert np.array_equal(a0, a1) and np.array_equal(l0, l1)            # every rank holds the same gathered block
        for r in range(2):
            want = fake_generate(torch.full((4, 6), 10 * r + k, dtype=torch.int64) + torch.arange(6)[None]).numpy()
            t = want.shape[1]
            assert np.all(l0[4 * r:4 * r + 4] == t)
            assert np.array_equal(a0[4 * r:4 * r + 4, :t], want)           # batch k of rank r, in submission order


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("native", [False, True])
def test_id_exchange_over_rccl_single_rank_group(native):
    """The exchange's collective path on real hardware: a process group of one rank over RCCL (backend "nccl"), the ids of three batches
    posted asynchronously (double-buffered) as all-gathers while a second stream keeps the GPU busy - what every rank of `bench.py --gpus N`
    runs, minus the peers.  native: the gathers through the library's own entry points (mg_dist_create / mg_dist_allgather: ncclAllGather
    called by libmgrapher_hip.so on the exchange's stream), torch.distributed carrying only the unique id."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29500 + (os.getpid() + 311 + (17 if native else 0)) % 1000)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ex = IdExchange(32, dev, pad_token_id=0, always_collective=True, native=native)
        assert ex.collective and ex.world == 1 and (ex._native is not None) == native
        side = torch.cuda.Stream()
        busy = torch.randn(2048, 2048, device=dev)
        mk = lambda step: ((torch.arange(32 * (257 - step), dtype=torch.int64).reshape(32, 257 - step) % 33201) + step).to(dev)
        with torch.cuda.stream(side):
            for _ in range(20):
                busy = busy @ busy * 1e-3
        h0, h1 = ex.post(mk(0)), ex.post(mk(1))
        r0 = [x.cpu().numpy().copy() for x in ex.wait(h0)]
        r1 = [x.cpu().numpy().copy() for x in ex.wait(h1)]
        r2 = [x.cpu().numpy().copy() for x in ex.wait(ex.post(mk(2)))]
        for step, (a, l) in enumerate((r0, r1, r2)):
            t = 257 - step
            assert a.shape == (32, ID_COLS) and a.dtype == np.int32 and np.all(l == t)
            assert np.array_equal(a[:, :t], mk(step).cpu().numpy()) and np.all(a[:, t:] == 0)
        torch.cuda.synchronize()
        ex.close()
    finally:
        dist.destroy_process_group()


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` as the driver invokes it (ONE process, WORLD_SIZE unset) must start its ranks itself.  Here without
    GPUs: MG_BENCH_BACKEND=gloo sends the ranks through the launcher self-test (rendezvous on 127.0.0.1, process group, barrier,
    max-over-ranks reduction, the id exchange's all-gather) instead of the measurement; rank 0 prints one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MG_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["launcher_selftest"] and out["n_gpus"] == 2 and out["exchange_ok"] and out["rows_gathered"] == 64
    # a rank count that disagrees with --gpus is refused (torch.distributed.run with another --nproc-per-node)
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env2, cwd=root, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, timeout=300)
    assert r2.returncode != 0 and "started 3 processes" in (r2.stderr + r2.stdout)
