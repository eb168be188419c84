"""Per-queue hand-over gaps of a rocprofv3 kernel trace (rocpd SQLite): time between a kernel's end and the start of the next kernel on
the same queue, as a histogram, plus the share of each queue's span that is gap.  Usage: rocpd_gaps.py results.db [name-filter]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    print("columns:", cols)
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    scol = "stream_id" if "stream_id" in cols else None
    key = qcol or scol
    rows = db.execute(f"select {key}, start, end, name from kernels order by {key}, start").fetchall()
    per = defaultdict(list)
    for q, s, e, n in rows:
        per[q].append((s, e, n))
    edges = [0.5, 1, 2, 4, 8, 16, 32, 64, 128, 256, 1024, 1e9]
    for q, ks in per.items():
        if len(ks) < 1000:
            continue
        hist = [0] * len(edges)
        gsum = [0.0] * len(edges)
        busy = 0
        for (s0, e0, _), (s1, e1, _) in zip(ks, ks[1:]):
            g = (s1 - e0) / 1e3
            for i, ed in enumerate(edges):
                if g < ed:
                    hist[i] += 1
                    gsum[i] += max(g, 0.0)
                    break
        pairs = defaultdict(lambda: [0, 0.0])
        for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
            g = (s1 - e0) / 1e3
            if 8 <= g < 128:
                k = (n0.split("(")[0][-60:], n1.split("(")[0][-60:])
                pairs[k][0] += 1
                pairs[k][1] += g
        for k, (c, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"      {c:>6} x {t/c:6.1f} us  {k[0]}  ->  {k[1]}")
        busy = sum(e - s for s, e, _ in ks) / 1e3
        span = (ks[-1][1] - ks[0][0]) / 1e3
        print(f"queue {q}: {len(ks)} kernels, span {span/1e3:.1f} ms, kernel time {busy/1e3:.1f} ms ({100*busy/span:.0f} %)")
        for ed, h, gs in zip(edges, hist, gsum):
            print(f"   gap < {ed:>6} us: {h:>8} hand-overs, {gs/1e3:9.1f} ms")


def releasers(db_path):
    """For hand-over gaps of 16-128 us: which kernel on ANOTHER queue ended last before the waiting kernel started (within 3 us)?"""
    import bisect
    db = sqlite3.connect(db_path)
    rows = db.execute("select queue_id, start, end, name from kernels order by start").fetchall()
    per = defaultdict(list)
    for q, s, e, n in rows:
        per[q].append((s, e, n))
    ends = sorted((e, q, n, s) for q, s, e, n in rows)
    end_t = [x[0] for x in ends]
    hist = defaultdict(lambda: [0, 0.0])
    none = 0
    for q, ks in per.items():
        if len(ks) < 50000:
            continue
        for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
            g = (s1 - e0) / 1e3
            if not (16 <= g < 128):
                continue
            i = bisect.bisect_right(end_t, s1) - 1
            found = None
            while i >= 0 and s1 - end_t[i] < 3000:
                e, q2, n2, st2 = ends[i]
                if q2 != q:
                    found = (n2.split("(")[0][-50:], (e - st2) / 1e3)
                    break
                i -= 1
            if found is None:
                none += 1
            else:
                hist[found[0]][0] += 1
                hist[found[0]][1] += found[1]
    # what the other queues were running at the middle of each such gap
    starts = sorted((s, e, q, n) for q, s, e, n in rows)
    start_t = [x[0] for x in starts]
    combos = defaultdict(int)
    for q, ks in per.items():
        if len(ks) < 50000:
            continue
        for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
            g = (s1 - e0) / 1e3
            if not (16 <= g < 128):
                continue
            mid = (e0 + s1) // 2
            i = bisect.bisect_right(start_t, mid) - 1
            running = []
            j = i
            while j >= 0 and mid - start_t[j] < 3_000_000:
                s, e, q2, n2 = starts[j]
                if q2 != q and e > mid:
                    nm = n2.split("(")[0]
                    tag = "xattn" if "attn_step_kernel<1, 8, true" in nm else ("sattn" if "attn_step" in nm else ("enc" if ("gemm_xl" in nm or "attention_enc" in nm) else "small"))
                    running.append(tag)
                j -= 1
            combos[tuple(sorted(running))] += 1
    print("other queues' kernels running at the middle of a 16-128 us gap:")
    for k, c in sorted(combos.items(), key=lambda kv: -kv[1])[:14]:
        print(f"  {c:>7}  {k}")
    # does a gap close right after ANOTHER queue's kernel starts (a shared dispatcher working through its queues in turn)?
    near_start = defaultdict(int)
    tot = 0
    for q, ks in per.items():
        if len(ks) < 50000:
            continue
        for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
            g = (s1 - e0) / 1e3
            if not (16 <= g < 128):
                continue
            tot += 1
            i = bisect.bisect_right(start_t, s1) - 1
            best = None
            j = i
            while j >= 0 and s1 - start_t[j] < 8000:
                if starts[j][2] != q:
                    best = (s1 - start_t[j]) / 1e3
                    break
                j -= 1
            k = i + 1
            nxt = None
            while k < len(starts) and start_t[k] - s1 < 8000:
                if starts[k][2] != q:
                    nxt = (start_t[k] - s1) / 1e3
                    break
                k += 1
            near_start["other start within 1 us before" if best is not None and best < 1 else
                       ("other start within 3 us before" if best is not None and best < 3 else
                        ("other start within 8 us before" if best is not None else "none within 8 us before"))] += 1
            near_start["other start within 1 us after" if nxt is not None and nxt < 1 else ("other start 1-8 us after" if nxt is not None else "none within 8 us after")] += 1
    print(f"gap closes relative to other queues' kernel STARTS ({tot} gaps):")
    for k, c in sorted(near_start.items(), key=lambda kv: -kv[1]):
        print(f"  {c:>7}  {k}")
    print("kernel on another queue that ended within 3 us before a 16-128 us gap closed (count, its mean duration):")
    for k, (c, t) in sorted(hist.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"  {c:>7} x {t/c:8.1f} us  {k}")
    print(f"  {none:>7} gaps with no such kernel")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--releasers":
        releasers(sys.argv[1])
        sys.exit(0)
    main()
