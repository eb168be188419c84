"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel table:
calls, total / average / min / max duration, share of GPU kernel time.  Usage: rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("mg::", "")
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    by_grid = "--by-grid" in sys.argv
    if by_grid:
        sys.argv.remove("--by-grid")
    rows = db.execute(f"select {namecol}, start, end, grid_x, workgroup_x, vgpr_count, lds_size from kernels").fetchall()
    agg = {}
    for name, s, e, gx, wx, vg, lds in rows:
        key = short(name) + (f" [grid {gx // max(wx, 1)}x{wx} vgpr {vg} lds {lds}]" if by_grid else "")
        a = agg.setdefault(key, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = (max(r[2] for r in rows) - min(r[1] for r in rows)) if rows else 0
    rows = [r[:3] for r in rows]
    lines = [f"# kernel trace summary: {len(rows)} dispatches, {tot/1e6:.2f} ms kernel time, {span/1e6:.2f} ms first-to-last span", "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {a[0]} | {a[1]/1e6:.3f} | {a[1]/a[0]/1e3:.2f} | {a[2]/1e3:.2f} | {a[3]/1e3:.2f} | {100*a[1]/tot:.1f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
