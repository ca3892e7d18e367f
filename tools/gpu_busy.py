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

# which kernels wait longest for their launch? (gap before a kernel, keyed by that kernel and by its predecessor)
try:
    names = {}
    for kt in [x for x in tabs if 'kernel_symbol' in x.lower()] + (['kernels'] if 'kernels' in tabs else []):
        kcols = [r[1] for r in cur.execute(f'pragma table_info({kt})')]
        nc = [c for c in ('kernel_name', 'display_name', 'name') if c in kcols]
        ic = [c for c in ('id', 'kernel_id') if c in kcols]
        if nc and ic:
            names = dict(cur.execute(f'select {ic[0]}, {nc[0]} from {kt}').fetchall())
            break
        print(kt, kcols)
    rows2 = sorted(cur.execute(f'select {sc}, {ec}, kernel_id from {t}').fetchall())[-n:]
    import collections
    by_next, by_prev = collections.defaultdict(lambda: [0, 0.0]), collections.defaultdict(lambda: [0, 0.0])
    big = []
    for i in range(len(rows2) - 1):
        gap = rows2[i + 1][0] - rows2[i][1]
        if gap <= 0:
            continue
        a, b = str(names.get(rows2[i][2], rows2[i][2]))[:44], str(names.get(rows2[i + 1][2], rows2[i + 1][2]))[:44]
        by_next[b][0] += 1; by_next[b][1] += gap
        by_prev[a][0] += 1; by_prev[a][1] += gap
        big.append((gap, a, b))
    print('--- gap time by the kernel that FOLLOWS the gap (top 14)')
    for k, v in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f'{v[1]/1e3:9.1f} us in {v[0]:5d} gaps  avg {v[1]/v[0]/1e3:6.2f}  {k}')
    print('--- gap time by the kernel that PRECEDES the gap (top 8)')
    for k, v in sorted(by_prev.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f'{v[1]/1e3:9.1f} us in {v[0]:5d} gaps  avg {v[1]/v[0]/1e3:6.2f}  {k}')
    print('--- largest gaps')
    for gap, a, b in sorted(big, reverse=True)[:12]:
        print(f'{gap/1e3:8.1f} us  {a}  ->  {b}')
except Exception as e:      # the summary above is what run_profile.sh needs; this part is diagnostics
    print('gap breakdown unavailable:', e)
