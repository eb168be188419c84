"""GPU probe: encoder output with and without the row-tile list (MG_ENC_ROW_TILES), row by row."""
import os, sys, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from markushgrapher_amd import synth
    from markushgrapher_amd.engine import Engine
    shape = synth.SHAPES["large"]
    eng = Engine(shape, max_decode_len=64)
    eng.load_state_dict(synth.recipe_state_dict(shape, gain=1.0))
    out = {}
    for name, inp in (("g2", synth.synth_batch(shape, 1, seed=21, fixed_L=64)), ("b4", synth.synth_batch(shape, 4, seed=5, L_min=40, L_max=200))):
        enc, mask = eng.encode(inp["input_ids"], inp["bbox"], inp["attention_mask"], inp["pixel_values"])
        out[name + "_enc"] = enc.cpu().numpy(); out[name + "_mask"] = mask.cpu().numpy()
    np.savez(sys.argv[2], **out)
    sys.exit(0)
res = {}
for v in ("0", "1"):
    f = f"/tmp/tiles_{v}.npz"
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", f], env=dict(os.environ, MG_ENC_ROW_TILES=v), check=True)
    res[v] = dict(np.load(f))
for name in ("g2", "b4"):
    e0, e1, m = res["0"][name + "_enc"], res["1"][name + "_enc"], res["0"][name + "_mask"].astype(bool)
    assert np.array_equal(res["0"][name + "_mask"], res["1"][name + "_mask"])
    for b in range(e0.shape[0]):
        diff = np.nonzero((e0[b] != e1[b]).any(1))[0]
        dv = [int(r) for r in diff if m[b, r]]
        S = m.shape[1]
        dead_tiles = [t for t in range(S // 32) if not m[b, 32 * t:32 * t + 32].any()]
        print(name, "image", b, "S", S, "attended", int(m[b].sum()), "dead tiles", dead_tiles, "rows differing", len(diff), "of them attended", len(dv), dv[:12],
              "max abs diff attended", float(np.abs(e0[b][m[b]] - e1[b][m[b]]).max()))
