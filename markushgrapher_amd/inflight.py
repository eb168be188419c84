"""Several batches in flight on one GPU.

The reference's evaluation loop (utils_evaluation.py:269-285) hands the model one batch after the other.  On MI355X one batch
cannot fill the machine while it decodes: five of the six launches of a decoder layer are latency-sized (a few MB of weights on
32 rows), only the cross-attention K/V stream is bandwidth-sized.  `InFlight` keeps `n` batches going at once, each on its own
execution context (`Engine.clone()`: shared weights; own workspace, decode graph, stream and host thread), so one batch's small
launches run in the gaps of another's.  Rows of different batches never meet, so every batch's result is exactly what a call made
alone returns (tests/test_inflight.py); what changes is throughput (profiles/r03_inflight_ab.txt).
"""
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor


_STREAMS = {}


def shared_streams(torch, device, n):
    """The process-wide streams of the batches in flight, `n` of at most InFlight.MAX.  The HIP runtime spreads streams over its
    hardware queues as they are created, and two busy streams that land on one queue run back to back: every user of several
    contexts in a process (InFlight, the pipeline's OCR contexts) takes its streams from this one list instead of creating more."""
    key = str(device)
    lst = _STREAMS.setdefault(key, [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=device))
    return lst[:n]


def plan_calls(k, n_contexts, max_per_call):
    """Cut `k` equal-shaped batches into calls for `n_contexts` execution contexts: every context gets k / n of them (the remainder
    spread), cut into calls of at most `max_per_call` batches of near-equal size, so that all contexts stay busy to the end (20 batches
    on 4 contexts, at most 4 per call: a call of 3 and a call of 2 each).  Returns the batches per call in submission order (the first
    calls of every context, then the second ones ...)."""
    per_ctx = []
    for i in range(n_contexts):
        q = k // n_contexts + (1 if i < k % n_contexts else 0)
        c = -(-q // max_per_call) if q else 0
        per_ctx.append([q // c + (1 if j < q % c else 0) for j in range(c)] if c else [])
    order = []
    for j in range(max((len(x) for x in per_ctx), default=0)):
        order += [x[j] for x in per_ctx if j < len(x)]
    return order


class InFlight:
    """`submit(fn, *args)` runs `fn(ctx, *args)` on the next free context and returns a future; `map(fn, items)` keeps the order.

    `fn` runs on a worker thread with the context's stream current; whatever it returns must be complete on the host's clock
    (Engine.generate ends with a synchronisation of its stream) and must not alias the context's reusable output buffers if it is
    read after the context's next call - copy inside `fn` (`ids.clone()`).
    """

    MAX = 4      # busy hardware queues beyond four are time-sliced by the chip's scheduler: throughput drops below one batch at a time

    def __init__(self, engine, n=4, streams=None, include_source=True, lock=None):
        """include_source=False: every context is a clone (the source engine stays free for its owner's other calls);
        lock: shared with other users of the same single-threaded backend (CPU emulator in the tests)."""
        if n < 1 or n > self.MAX:
            raise ValueError("InFlight: n must be in [1, %d] (more contexts than compute pipes collapse: profiles/r03_inflight_ab.txt)" % self.MAX)
        self._owned_from = 1 if include_source else 0
        self.contexts = ([engine] if include_source else []) + [engine.clone() for _ in range(n - (1 if include_source else 0))]
        torch = getattr(engine.mem, "torch", None)
        if streams is not None:
            self.streams = list(streams)
        elif torch is not None and torch.cuda.is_available():
            self.streams = shared_streams(torch, engine.mem.device, n)
        else:
            self.streams = [None] * n          # CPU emulator backend (tests): no streams; jobs run one at a time
        self._torch = torch
        self._free = queue.SimpleQueue()
        for i in range(n):
            self._free.put(i)
        self._pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix="mg-inflight")
        self._lock = lock if lock is not None else threading.Lock()
        # several contexts on one GPU: every context's cross-attention stream leaves wave slots to the others (mg_set_shared_gpu)
        # (MG_SHARED_GPU=0: leave the contexts' setting alone - A/B runs)
        share = n > 1 and os.environ.get("MG_SHARED_GPU", "1") != "0"
        # only ever switched ON here: with one context or MG_SHARED_GPU=0 whatever the owner had set stays (and so do its captured graphs)
        self._shared_prev = [c.set_shared_gpu(True) for c in self.contexts] if (share and hasattr(engine, "set_shared_gpu")) else None

    def __len__(self):
        return len(self.contexts)

    def _run(self, fn, args):
        i = self._free.get()
        try:
            st = self.streams[i]
            if st is None:
                with self._lock:                # the emulator is single-threaded test infrastructure
                    return fn(self.contexts[i], *args)
            # a new host thread starts on device 0: make the context's GPU current for the raw HIP calls of the library too
            with self._torch.cuda.device(st.device), self._torch.cuda.stream(st):
                try:
                    return fn(self.contexts[i], *args)
                finally:
                    st.synchronize()        # also after an exception: the context's buffers are free for its next job
        finally:
            self._free.put(i)

    def submit(self, fn, *args):
        return self._pool.submit(self._run, fn, args)

    def map(self, fn, items):
        futures = [self.submit(fn, it) for it in items]
        return [f.result() for f in futures]

    def generate_batches(self, batches, max_batches_per_call=4, **gen_kwargs):
        """Greedy `generate` over a list of batches (dicts with input_ids / bbox / attention_mask / pixel_values of EQUAL shapes: the same
        number of rows and the same padded text length) with up to `max_batches_per_call` of them per call (rows side by side: one pass
        over the decoder's weights per step for the call's batches; every image's ids are bit-identical to a call on its batch alone under
        the same cross-attention form - the form is pinned for the invocation) and the calls spread over the contexts (plan_calls).  Returns one host array of ids per batch, in order."""
        import numpy as np
        torch = self._torch
        if gen_kwargs.get("num_beams", 1) != 1:
            max_batches_per_call = 1
        keys = ("input_ids", "bbox", "attention_mask", "pixel_values")
        sizes = plan_calls(len(batches), len(self), max(1, int(max_batches_per_call)))
        rows = int(batches[0]["input_ids"].shape[0]) if batches else 0
        for i, b in enumerate(batches):            # a short last batch (the usual dataloader tail) would be cut at the wrong offsets below
            for k in keys:
                if tuple(b[k].shape) != tuple(batches[0][k].shape):
                    raise ValueError(f"generate_batches: batch {i} has {k} of shape {tuple(b[k].shape)}, batch 0 has {tuple(batches[0][k].shape)}: "
                                     "batches must have equal shapes (pad the tail batch or pass it in a call of its own)")

        def cat(parts):
            if torch is not None and all(torch.is_tensor(p) for p in parts):          # (numpy 2 arrays have a .device too)
                return torch.cat(list(parts), dim=0)
            return np.concatenate([np.asarray(p) for p in parts], axis=0)

        def job(ctx, group):
            args = [cat([b[k] for b in group]) for k in keys]
            out = ctx.generate(*args, **gen_kwargs)[0]
            out = out.cpu().numpy() if hasattr(out, "cpu") else np.array(out, copy=True)
            return [out[j * rows:(j + 1) * rows] for j in range(len(group))]

        # the calls of one invocation may differ in size (3 + 2 batches ...): pin ONE cross-attention form for all of them, so that every batch
        # goes through the same arithmetic whatever call it rides in (Engine.set_cross_absorb: "auto" picks the form by the call's rows)
        pinned = None
        if sizes and hasattr(self.contexts[0], "set_cross_absorb") and gen_kwargs.get("num_beams", 1) == 1:
            form = bool(rows * max(sizes) >= getattr(self.contexts[0], "ABSORB_AUTO_ROWS", 96))
            try:
                pinned = [c.set_cross_absorb(form) for c in self.contexts]
            except Exception:              # (a geometry without the absorbed form)
                pinned = None
        try:
            futures, lo = [], 0
            for nb in sizes:
                futures.append(self.submit(job, batches[lo:lo + nb]))
                lo += nb
            return [ids for f in futures for ids in f.result()]
        finally:
            if pinned is not None:
                for c, prev in zip(self.contexts, pinned):
                    c.set_cross_absorb(prev)

    def close(self):
        self._pool.shutdown(wait=True)
        if self._shared_prev is not None:
            for c, prev in zip(self.contexts[:self._owned_from], self._shared_prev):
                c.set_shared_gpu(prev)              # (the source engine goes back to what its owner had set)
        for c in self.contexts[self._owned_from:]:
            c.close()
        self.contexts = self.contexts[:self._owned_from]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
