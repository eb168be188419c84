"""Per-kernel average of a PMC counter from a rocprofv3 rocpd database (one --pmc pass).
Usage: rocpd_pmc.py results.db [out.md]   — FETCH_SIZE/WRITE_SIZE are in KiB per dispatch."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select kernel_name, grid_size, workgroup_size, counter_name, value, duration from counters_collection").fetchall()
    agg = {}
    for name, grid, wg, cname, val, dur in rows:
        k = (re.sub(r"\(.*$", "", name).replace("void ", "").replace("mg::", "")[:70], grid // max(wg, 1), wg, cname)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += val; a[2] += dur
    lines = ["| kernel | grid x wg | counter | dispatches | avg value | avg us (profiled) |", "|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k[0]} | {k[1]}x{k[2]} | {k[3]} | {a[0]} | {a[1] / a[0]:.1f} | {a[2] / a[0] / 1e3:.2f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
