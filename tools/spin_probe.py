"""Probe: does a host thread that waits for the GPU sleep or spin?  stream.synchronize(), an event created with blocking=True (hipEventBlockingSync)
and a poll-and-sleep loop, each behind the same GPU work: wall time against the process CPU time (markushgrapher_amd/csrc/mg_device.h mg_stream_sync).
    python tools/spin_probe.py"""
import torch, time, resource, os
x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
def work():
    for _ in range(600):
        y = x @ x
def cpu():
    r = resource.getrusage(resource.RUSAGE_SELF); return r.ru_utime + r.ru_stime
work(); torch.cuda.synchronize()
for mode in ("stream.synchronize", "blocking event", "event poll+sleep"):
    c0, t0 = cpu(), time.time()
    work()
    if mode == "stream.synchronize":
        torch.cuda.current_stream().synchronize()
    elif mode == "blocking event":
        e = torch.cuda.Event(blocking=True); e.record(); e.synchronize()
    else:
        e = torch.cuda.Event(); e.record()
        while not e.query():
            time.sleep(0.0005)
    print(f"{mode}: wall {time.time()-t0:.2f}s cpu {cpu()-c0:.2f}s")
print("env", {k: v for k, v in os.environ.items() if "HIP" in k or "AMD" in k or "GPU_" in k or "HSA" in k})
