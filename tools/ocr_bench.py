"""ChemicalOCR stage alone: prints bench.py's `extra_runs.ocr_stage` object.  python tools/ocr_bench.py [B] [new_tokens]"""
import json
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    print(json.dumps(bench.ocr_stage_run(B, n)))
