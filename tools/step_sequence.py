"""Ordered kernel sequence of the LAST training step in a rocprofv3 --kernel-trace database (rocpd sqlite): one line per dispatch with its
duration, so that the cost of one module instance (e.g. the level-0 temporal attention: layernorm -> qkv projection -> attention -> to_out)
can be read off instead of inferred from per-kernel averages over all levels.

    python tools/step_sequence.py <results.db> <out.txt> [marker-substring (default: q_sample_cond)]
The step is taken from the last dispatch whose name contains the marker (the first kernel of p_losses) to the end of the trace."""
import re
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else 'q_sample_cond'
cur = sqlite3.connect(db).cursor()
cand = None
for typ, name in list(cur.execute("select type, name from sqlite_master where type in ('table', 'view')")):
    try:
        cols = [r[1] for r in cur.execute(f'pragma table_info("{name}")')]
    except Exception:
        continue
    low = [c.lower() for c in cols]
    if 'start' in low and 'end' in low and ('name' in low or 'kernel_name' in low) and 'kernel' in name.lower():
        if cand is None or typ == 'view':
            cand = (name, cols)
if cand is None:
    open(out, 'w').write('no kernel table found\n')
    sys.exit(0)
name, cols = cand
ncol = 'name' if 'name' in cols else 'kernel_name'
extra = [c for c in ('grid_size', 'grid_x', 'workgroup_size', 'workgroup_x') if c in cols]
rows = list(cur.execute(f'select {ncol}, start, end {"".join(", " + c for c in extra)} from "{name}" order by start'))
marks = [i for i, r in enumerate(rows) if marker in r[0]]
steps = rows[marks[-2]:marks[-1]] if len(marks) >= 2 else rows[(marks[-1] if marks else 0):]


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    m = re.match(r'_Z\d+([a-zA-Z_0-9]+?)(I.*)?$', n)
    if m:
        args = re.findall(r'Li(\d+)E|Lb([01])E', m.group(2) or '')
        n = m.group(1) + ('<' + ','.join(a or b for a, b in args) + '>' if args else '')
    return n[:90]


t0 = steps[0][1]
with open(out, 'w') as f:
    f.write(f'# {len(steps)} dispatches, {(steps[-1][2] - t0) / 1e6:.3f} ms from first start to last end; columns: index, start offset us, duration us, gap before us, kernel{", " + ", ".join(extra) if extra else ""}\n')
    prev_end = t0
    for i, r in enumerate(steps):
        f.write(f'{i:5d} {(r[1] - t0) / 1e3:10.1f} {(r[2] - r[1]) / 1e3:9.2f} {(r[1] - prev_end) / 1e3:7.2f}  {short(r[0])}' + ''.join(f' {v}' for v in r[3:]) + '\n')
        prev_end = max(prev_end, r[2])
print('wrote', out, len(steps), 'dispatches')
