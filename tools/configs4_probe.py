"""Probe: configs[4] loop, sweep of contexts in flight inside the stages / OCR form / queue length.
    python tools/configs4_probe.py"""
import json
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    shape = synth.SHAPES["large"]
    eng = Engine(shape, max_decode_len=512)
    eng.load_state_dict(synth.recipe_state_dict(shape, **synth.BENCH_RECIPE))
    keep = ("pages_per_s", "ocr_s", "host_s", "main_s", "ocr_form")
    cases = (dict(ocr_slots=256), dict(ocr_slots=128, main_inflight=4, ocr_inflight=4),
             dict(ocr_pages=2048, ocr_slots=128, main_inflight=4, ocr_inflight=4), dict(ocr_pages=2048, ocr_slots=256, main_inflight=4, ocr_inflight=4),
             dict(ocr_pages=1024, ocr_slots=128, main_inflight=4, ocr_inflight=4))
    if len(sys.argv) > 1:
        cases = cases[int(sys.argv[1]):int(sys.argv[1]) + 1]
    for kw in cases:
        kw = dict(kw)
        r = bench.configs4_run(eng, 32, 256, ocr_pages=kw.pop("ocr_pages", 512), **kw)
        print(kw, {k: r[k] for k in keep}, r["ocr_strings_as_scripted"], flush=True)


if __name__ == "__main__":
    main()
