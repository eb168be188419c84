"""For the decode-step kernels of a rocprofv3 kernel trace (rocpd SQLite): duration while an ENCODER kernel of another context (gemm_pp / gemm_xl /
attention_enc) is running somewhere on the GPU against duration while none is, and duration percentiles.  Usage: rocpd_overlap.py results.db"""
import bisect
import re
import sqlite3
import sys

import numpy as np


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, start, end, grid_x, workgroup_x from kernels").fetchall()
    enc = sorted((s, e) for n, s, e, gx, wx in rows if ("gemm_pp_kernel" in n or "gemm_xl_kernel" in n or "attention_enc_kernel" in n))
    starts = [s for s, _ in enc]
    # merged busy intervals of encoder kernels
    merged = []
    for s, e in enc:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    ms = [m[0] for m in merged]

    def enc_overlap(s, e):
        i = bisect.bisect_right(ms, e) - 1
        tot = 0
        while i >= 0 and merged[i][1] > s:
            tot += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
            i -= 1
        return tot

    groups = {}
    for n, s, e, gx, wx in rows:
        if not any(k in n for k in ("attn_step_kernel", "gemm_rows", "greedy_select")):
            continue
        key = re.sub(r"\(.*$", "", n).replace("void ", "").replace("mg::", "")[:70] + f" [grid {gx // max(wx, 1)}]"
        d = e - s
        ov = enc_overlap(s, e) / max(d, 1)
        groups.setdefault(key, []).append((d, ov))
    print("| kernel | calls | avg us | p50 | p90 | p99 | share of launches beside an encoder kernel | avg us beside | avg us not beside |")
    print("|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(groups.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
        d = np.array([x[0] for x in v], dtype=np.float64) / 1e3
        o = np.array([x[1] for x in v])
        a, b = d[o > 0.5], d[o <= 0.5]
        print(f"| {k} | {len(d)} | {d.mean():.1f} | {np.percentile(d, 50):.1f} | {np.percentile(d, 90):.1f} | {np.percentile(d, 99):.1f} | "
              f"{(o > 0.5).mean():.2f} | {a.mean() if len(a) else float('nan'):.1f} | {b.mean() if len(b) else float('nan'):.1f} |")


if __name__ == "__main__":
    main()
