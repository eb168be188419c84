"""Where a decode step's time goes, from a rocprofv3 kernel trace (rocpd SQLite): per queue, the kernels between two token selections
(greedy_select*) are one step; per position in the step: kernel name, average duration, average gap to the previous kernel's end.
Usage: rocpd_step_gaps.py results.db [out.md] [--mark kernel-name-prefix]   (default mark: greedy_select)"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("mg::", "")[:70]


def main():
    mark = "greedy_select"
    if "--mark" in sys.argv:
        i = sys.argv.index("--mark")
        mark = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    key = "queue_id" if "queue_id" in cols else "stream_id"
    rows = db.execute(f"select {key}, start, end, name, grid_x, workgroup_x from kernels order by {key}, start").fetchall()
    per = defaultdict(list)
    for q, s, e, n, gx, wx in rows:
        per[q].append((s, e, short(n), gx // max(wx, 1)))
    out = []
    for q, ks in per.items():
        sel = [i for i, k in enumerate(ks) if k[2].startswith(mark)]
        if len(sel) < 6:
            continue
        steps = [ks[a:b + 1] for a, b in zip(sel[:-1], sel[1:])]          # [select_i, kernels of step i+1 ..., select_{i+1}]
        n0 = max(set(len(s) for s in steps), key=[len(s) for s in steps].count)
        steps = [s for s in steps if len(s) == n0][2:]                     # steady state, same launch sequence
        if not steps:
            continue
        dur = [0.0] * n0; gap = [0.0] * n0
        for s in steps:
            for i in range(1, n0):
                dur[i] += (s[i][1] - s[i][0]) / 1e3
                gap[i] += (s[i][0] - s[i - 1][1]) / 1e3
        ns = len(steps)
        wall = sum((s[-1][1] - s[0][1]) / 1e3 for s in steps) / ns
        tk = sum(dur) / ns; tg = sum(gap) / ns
        out.append(f"## queue {q}: {ns} steps of {n0 - 1} launches; step {wall:.1f} us = {tk:.1f} us inside kernels + {tg:.1f} us between them")
        agg = defaultdict(lambda: [0, 0.0, 0.0])
        for i in range(1, n0):
            k = f"{steps[0][i][2]} [{steps[0][i][3]} wg]"
            a = agg[k]; a[0] += 1; a[1] += dur[i] / ns; a[2] += gap[i] / ns
        out.append("| kernel | launches per step | us per step inside | avg us | us per step waiting before it | avg gap us |")
        out.append("|---|---|---|---|---|---|")
        for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            out.append(f"| {k} | {a[0]} | {a[1]:.1f} | {a[1] / a[0]:.2f} | {a[2]:.1f} | {a[2] / a[0]:.2f} |")
        out.append("")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
