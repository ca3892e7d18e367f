"""GPU busy fraction from a rocprofv3 --kernel-trace rocpd database: over the last N dispatches, sum of kernel durations
vs. the span they cover (gaps = launch latency / host starvation)."""
import sqlite3, sys
db = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
con = sqlite3.connect(db)
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cand = [t for t in tabs if 'kernel_dispatch' in t.lower() or t.lower() == 'kernels']
print('tables:', cand[:6])
t = [c for c in cand if 'kernel_dispatch' in c.lower()][0] if any('kernel_dispatch' in c.lower() for c in cand) else cand[0]
cols = [r[1] for r in cur.execute(f'pragma table_info({t})')]
print(t, cols)
sc = 'start' if 'start' in cols else [c for c in cols if 'start' in c.lower()][0]
ec = 'end' if 'end' in cols else [c for c in cols if 'end' in c.lower()][0]
rows = sorted(cur.execute(f'select {sc}, {ec} from {t}').fetchall())
rows = rows[-n:]
busy = sum(e - s for s, e in rows)
span = rows[-1][1] - rows[0][0]
gaps = sorted((rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1))
print(f'last {len(rows)} dispatches: busy {busy/1e6:.2f} ms, span {span/1e6:.2f} ms, busy fraction {busy/span:.3f}')
print(f'gap median {gaps[len(gaps)//2]/1e3:.2f} us, p90 {gaps[int(len(gaps)*0.9)]/1e3:.2f} us, max {gaps[-1]/1e3:.1f} us, sum of positive gaps {sum(g for g in gaps if g > 0)/1e6:.2f} ms')
