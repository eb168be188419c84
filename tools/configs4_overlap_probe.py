"""Probe: the configs[4] loop with its two GPU stages overlapped (pipeline.Configs4Pipeline(overlap_slab=...)) against stage-after-stage.
    python tools/configs4_overlap_probe.py"""
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
    keep = ("pages_per_s", "ocr_s", "host_s", "main_s")
    cases = (dict(ocr_slots=128, main_inflight=4, ocr_inflight=4, main_batch=64),
             dict(ocr_slots=128, main_inflight=4, ocr_inflight=4, main_batch=64, overlap_slab=128),
             dict(ocr_slots=64, main_inflight=4, ocr_inflight=2, main_batch=64, overlap_slab=128),
             dict(ocr_slots=128, main_inflight=2, ocr_inflight=2, main_batch=64, overlap_slab=128),
             dict(ocr_slots=128, main_inflight=3, ocr_inflight=1, main_batch=64, overlap_slab=128),
             dict(ocr_slots=128, main_inflight=4, ocr_inflight=4, main_batch=64, overlap_slab=256))
    if len(sys.argv) > 1:          # indices into the list above, or JSON dicts of configs4_run keywords
        import json
        cases = [cases[int(a)] if a.isdigit() else json.loads(a) for a in sys.argv[1:]]
    for kw in cases:
        r = bench.configs4_run(eng, 32, 256, ocr_pages=512, **kw)
        print(kw, {k: r[k] for k in keep}, r["ocr_strings_as_scripted"], flush=True)


if __name__ == "__main__":
    main()
