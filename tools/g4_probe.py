"""GPU-side probe: error statistics of the HIP path against the G4 fixture (benchmark configuration), used to set the
tolerances written in tests/test_bench_config.py.  python tools/g4_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from markushgrapher_amd import synth
from markushgrapher_amd.engine import Engine

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g4_bench.npz")))
shape = synth.SHAPES["large"]
B, NEW = int(g["batch"]), int(g["new_tokens"])
sd = synth.recipe_state_dict(shape, **dict(zip(("gain", "embed_gain", "ffn_gain", "xq_gain"), [float(v) for v in g["recipe"]])))
eng = Engine(shape, max_decode_len=64)
eng.load_state_dict(sd)
inp = synth.synth_batch(shape, B, seed=int(g["synth_seed"]), return_pages=True)
pix = eng.preprocess(inp["pages_u8"])
args = (inp["input_ids"], inp["bbox"], inp["attention_mask"], pix)
enc, mask = eng.encode(*args)
enc, mask = eng.mem.numpy(enc), eng.mem.numpy(mask)
print("mask equal", np.array_equal(mask, g["enc_mask"].astype(np.uint8)))
rel, pe = [], []
for b in range(B):
    v = mask[b].astype(bool)
    rel.append(abs(np.abs(enc[b][v]).astype(np.float64).sum() - g["enc_abs_sum"][b]) / g["enc_abs_sum"][b])
    pe.append(np.abs(enc[b][g["enc_rows"][b]] - g["enc_probe"][b]))
pe = np.stack(pe)
print("enc abs-sum rel err max %.2e; probe rows max %.4f mean %.5f" % (max(rel), pe.max(), pe.mean()))

def top8_stats(cap, vals, idx, name):
    # cap [B, steps, V]; vals/idx [B, steps, 8]
    at = np.take_along_axis(cap, idx, -1)
    err = np.abs(at - vals)
    print(f"{name}: |logit err| at stock's top-8: max {err.max():.4f} mean {err.mean():.5f} p99 {np.quantile(err, 0.99):.4f}; max|logit| {np.abs(vals).max():.2f}")
    am = cap.argmax(-1)
    margin = vals[..., 0] - vals[..., 1]
    for thr in (0.0, 0.02, 0.05, 0.1, 0.2):
        sel = margin > thr
        print(f"   margin > {thr}: {sel.sum()} cases, argmax equal {(am[sel] == idx[..., 0][sel]).mean():.4f}")
    return err

# free-running greedy with capture
cap = eng.debug_decode_capture(NEW, B, None)
ids, _, top2 = eng.generate(*args, max_length=NEW + 1, min_length=NEW + 1, return_top2=True)
ids = eng.mem.numpy(ids)
same = (ids == g["greedy_ids"])
first_div = [int(np.argmin(same[b])) if not same[b].all() else NEW + 1 for b in range(B)]
print("free-running greedy: rows fully equal", int(same.all(1).sum()), "of", B, "; first divergence per row", first_div)
mg = g["step_top_vals"][..., 0] - g["step_top_vals"][..., 1]
for b in range(B):
    if first_div[b] <= NEW:
        t = first_div[b]
        print(f"   row {b}: diverges at step {t}, stock margin there {mg[b, t - 1]:.4f}")
# forced decoding with stock's ids
cap = eng.debug_decode_capture(NEW, B, g["greedy_ids"])
ids_f, _, _ = eng.generate(*args, max_length=NEW + 1, min_length=NEW + 1)
c = eng.mem.numpy(cap).transpose(1, 0, 2).copy()
eng.debug_decode_capture()
top8_stats(c, g["step_top_vals"], g["step_top_idx"], "forced decode path")
# teacher-forced forward
from oracle.udop_oracle import Oracle
labels = g["labels"]
dec_ids = Oracle.shift_right(labels, shape.decoder_start_token_id, shape.pad_token_id).numpy()
dam = (labels != -100).astype(np.uint8)
logits, _, _ = eng.forward_logits(*args, dec_ids, dam)
lg = eng.mem.numpy(logits)
top8_stats(lg, g["tf_top_vals"], g["tf_top_idx"], "teacher-forced forward")
# beam-5 on the subset and on the whole batch
NB = int(g["beam_rows"])
sub = tuple(a[:NB] for a in args)
bids, bsc, _ = eng.generate(*sub, num_beams=5, max_length=NEW + 1, min_length=NEW + 1)
bids, bsc = eng.mem.numpy(bids), eng.mem.numpy(bsc)
print("beam-5 subset: ids equal rows", [bool(np.array_equal(bids[b], g["beam_ids"][b])) for b in range(NB)], "scores", bsc.tolist(), "stock", g["beam_scores"].tolist())
t0 = time.time()
bids32, bsc32, _ = eng.generate(*args, num_beams=5, max_length=NEW + 1, min_length=NEW + 1)
bids32, bsc32 = eng.mem.numpy(bids32), eng.mem.numpy(bsc32)
print("beam-5 B=32: first rows equal subset", [bool(np.array_equal(bids32[b], bids[b])) for b in range(NB)], "score diff", np.abs(bsc32[:NB] - bsc).max())
