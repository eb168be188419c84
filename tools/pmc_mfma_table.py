"""MFMA-utilisation table from one rocprofv3 --pmc pass (rocpd database):
    SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
clock = SQ_BUSY_CYCLES / 32 shader engines / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock);
wait / issuing = share of SQ_WAVE_CYCLES.   Usage: pmc_mfma_table.py results.db out.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, grid_size, workgroup_size, counter_name, value, duration from counters_collection").fetchall()
agg = {}
for name, grid, wg, cname, val, dur in rows:
    k = (re.sub(r"\(.*$", "", name).replace("void ", "").replace("mg::", "")[:70], grid // max(wg, 1), wg)
    a = agg.setdefault(k, {})
    c = a.setdefault(cname, [0, 0.0, 0.0])
    c[0] += 1; c[1] += val; c[2] += dur
lines = ["| kernel | grid | us | clock GHz | MFMA busy | waiting on s_waitcnt | waiting (any) | issuing |", "|---|---|---|---|---|---|---|---|"]
out = []
for k, a in agg.items():
    need = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY")
    if not all(n in a for n in need):
        continue
    avg = {n: a[n][1] / a[n][0] for n in need}
    us = a["SQ_WAVE_CYCLES"][2] / a["SQ_WAVE_CYCLES"][0] / 1e3
    if us < 5:
        continue
    clock = avg["SQ_BUSY_CYCLES"] / 32 / (us * 1e3)
    mfma = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * us * 1e3 * clock) if clock > 0 else 0.0
    wc = max(avg["SQ_WAVE_CYCLES"], 1.0)
    out.append((us * a["SQ_WAVE_CYCLES"][0], f"| {k[0]} | {k[1]}x{k[2]} | {us:.1f} | {clock:.2f} | {100 * mfma:.1f} % | {100 * avg['SQ_WAIT_INST_ANY'] / wc:.0f} % | "
                                              f"{100 * avg['SQ_WAIT_ANY'] / wc:.0f} % | {100 * avg['SQ_ACTIVE_INST_ANY'] / wc:.0f} % |"))
lines += [l for _, l in sorted(out, key=lambda t: -t[0])]
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))
