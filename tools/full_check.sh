#!/bin/bash
# full check of the tree on the GPU box: GPU tests, default bench -> gpurun_out/<tag>_*   usage: tools/full_check.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-check}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/${tag}_gputests.txt
cat gpurun_out/${tag}_gputests.txt
timeout 1200 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
python - <<PY
import json
for l in open("gpurun_out/${tag}_bench_default.json"):
    l=l.strip()
    if l.startswith("{"):
        d=json.loads(l); p=d["phases"]; o=d["one_batch_in_flight"]
        print("headline %.2f images/s, %.1f ms/step, ids equal %s; solo %.2f images/s enc %.2f ms (%.4f) step %.4f ms roofline %.4f" % (d["value"], d["ms_per_step"], d["config"].get("ids_equal_one_batch_calls"), o["images_per_s"], o["phases"]["encoder_ms"], o["phases"]["enc_mfma_frac"], o["phases"]["decode_step_ms"], o["roofline"]["frac"]))
        print("   whole_job", p["whole_job"])
        for k,v in d["extra_runs"].items():
            print("  ", k, v.get("images_per_s", v.get("pages_per_s")), v.get("ids_equal_batch_calls", ""))
PY
