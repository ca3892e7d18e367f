"""Timeline of one training step from a rocprofv3 --kernel-trace database (rocpd sqlite): how much of the step is idle time BETWEEN
kernels, how much do kernels overlap, which kernels are followed by the longest gaps -- the data behind "would fewer launches or a
captured graph help".

    python tools/timeline_gaps.py <results.db> <out.md> [skip_fraction]
The last (1 - skip_fraction) of the dispatches is analysed (default: the last 40 %, i.e. steady-state steps)."""
import sqlite3
import sys
from collections import defaultdict

db, out = sys.argv[1], sys.argv[2]
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
con = sqlite3.connect(db)
cur = con.cursor()
objs = list(cur.execute("select type, name from sqlite_master where type in ('table', 'view')"))
cand = None
schema = []
for typ, name in objs:
    try:
        cols = [r[1] for r in cur.execute(f'pragma table_info("{name}")')]
    except Exception:
        continue
    schema.append((typ, name, cols))
    low = [c.lower() for c in cols]
    if 'start' in low and 'end' in low and ('name' in low or 'kernel_name' in low) and 'kernel' in name.lower():
        if cand is None or typ == 'view':
            cand = (name, cols)
lines = []
if cand is None:
    lines.append('no kernel table found; schema:')
    lines += [f'{t} {n}: {c}' for t, n, c in schema]
    open(out, 'w').write('\n'.join(lines))
    sys.exit(0)
name, cols = cand
ncol = 'name' if 'name' in cols else 'kernel_name'
rows = list(cur.execute(f'select {ncol}, start, end from "{name}" order by start'))
rows = rows[int(len(rows) * skip):]
t0, t1 = rows[0][1], max(r[2] for r in rows)
wall = t1 - t0
busy = 0
cur_end = rows[0][1]
gaps = []
after = defaultdict(lambda: [0, 0])
overlap = 0
for i, (nm, s, e) in enumerate(rows):
    if s > cur_end:
        gaps.append((s - cur_end, rows[i - 1][0], nm))
        after[rows[i - 1][0]][0] += s - cur_end
        after[rows[i - 1][0]][1] += 1
        busy += e - s
        cur_end = e
    else:
        overlap += min(e, cur_end) - s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
ksum = sum(e - s for _, s, e in rows)
gsum = sum(g[0] for g in gaps)
lines.append(f'# kernel timeline ({name}, last {100 * (1 - skip):.0f} % of the dispatches)\n')
lines.append(f'dispatches {len(rows)}; wall {wall / 1e6:.3f} ms; union of kernel intervals {busy / 1e6:.3f} ms; sum of kernel durations {ksum / 1e6:.3f} ms '
             f'(overlap {overlap / 1e6:.3f} ms); idle between kernels {gsum / 1e6:.3f} ms in {len(gaps)} gaps (mean {gsum / max(1, len(gaps)) / 1e3:.2f} us)\n')
hist = defaultdict(int)
for g, _, _ in gaps:
    b = 1
    while b * 1000 < g:
        b *= 2
    hist[b] += 1
lines.append('gap histogram (upper bound us: count, total ms): ' + ', '.join(f'<{b}: {n}' for b, n in sorted(hist.items())) + '\n')
lines.append('| kernel followed by idle time | gaps | total us | mean us |\n|---|---:|---:|---:|')
for k, (tot, n) in sorted(after.items(), key=lambda kv: -kv[1][0])[:25]:
    lines.append(f'| `{k[:90]}` | {n} | {tot / 1e3:.1f} | {tot / n / 1e3:.2f} |')
lines.append('\nlargest single gaps (us, after -> before):')
for g, a, b in sorted(gaps, key=lambda x: -x[0])[:12]:
    lines.append(f'- {g / 1e3:.1f}: `{a[:60]}` -> `{b[:60]}`')
pairs = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    if 1e3 <= g < 3e5:                      # inside a step (a replayed graph): 1 us .. 300 us
        pairs[(a, b)][0] += g
        pairs[(a, b)][1] += 1
lines.append('\n| gap of 1-300 us: after | before | gaps | total us | mean us |\n|---|---|---:|---:|---:|')
for (a, b), (tot, n) in sorted(pairs.items(), key=lambda kv: -kv[1][0])[:40]:
    lines.append(f'| `{a[:48]}` | `{b[:48]}` | {n} | {tot / 1e3:.1f} | {tot / n / 1e3:.2f} |')
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:6]))
