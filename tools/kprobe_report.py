import glob, sqlite3, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from kprobe import CASES  # noqa
db = glob.glob(sys.argv[1])[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = c.execute(f"select k.start, k.end, s.kernel_name from {kd} k join {ks} s on k.kernel_id = s.id order by k.start").fetchall()
rows = [r for r in rows if 'gemm_rows' in r[2]]
for i, (n, _) in enumerate(CASES):
    ch = rows[i * 100:(i + 1) * 100]
    d = sorted((r[1] - r[0]) / 1e3 for r in ch)
    gaps = sorted((ch[j + 1][0] - ch[j][1]) / 1e3 for j in range(len(ch) - 1))
    print(f"{n:26s} avg {sum(d)/len(d):6.2f} med {d[50]:6.2f} min {d[0]:6.2f} us   median gap to next {gaps[len(gaps)//2]:5.2f} us")
