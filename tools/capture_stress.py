"""Stress: one context re-captures its decode step on every call (alternating max_length) while another decodes; any HIP error that
the capture provokes in the OTHER thread shows up here (AMD_LOG_LEVEL=1 names the API).  python tools/capture_stress.py [seconds]"""
import os
import sys
import threading
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np


def main():
    import torch
    from markushgrapher_amd.inflight import shared_streams
    from tests.backends import make_engine
    from tests.conftest import load_golden
    from tests.test_oracle_golden import _weights, _inputs
    g = load_golden("g3_trained_tiny.npz")
    shape, sd = _weights(g)
    inp = _inputs(g, shape)
    eng = make_engine("hip", shape, sd)
    ctx = eng.clone()
    sts = shared_streams(torch, eng.mem.device, 2)
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    want = {T: eng.mem.numpy(eng.generate(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"], max_length=T, min_length=T)[0]).copy()
            for T in (8, 9, 12)}
    stop = time.time() + secs
    stats = {"a": 0, "b": 0, "err": []}

    def run(e, st, Ts, key):
        with torch.cuda.device(st.device), torch.cuda.stream(st):
            i = 0
            while time.time() < stop:
                T = Ts[i % len(Ts)]
                i += 1
                try:
                    ids, _, _ = e.generate(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"], max_length=T, min_length=T)
                    st.synchronize()
                    if not np.array_equal(ids.cpu().numpy(), want[T]):
                        stats["err"].append((key, "ids differ"))
                except Exception as ex:
                    stats["err"].append((key, repr(ex)[:300]))
                stats[key] += 1
    th = [threading.Thread(target=run, args=(eng, sts[0], (8, 9), "a")), threading.Thread(target=run, args=(ctx, sts[1], (12,), "b"))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    print("calls:", stats["a"], stats["b"], "errors:", len(stats["err"]))
    for e in stats["err"][:5]:
        print("  ", e)


if __name__ == "__main__":
    main()
