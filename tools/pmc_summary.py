"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd results database -> markdown."""
import collections
import sqlite3
import sys

out = sys.argv[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for db in sys.argv[1:-1]:
    cur = sqlite3.connect(db).cursor()
    for name, cn, v in cur.execute('select kernel_name, counter_name, value from counters_collection'):
        agg[name][cn].append(v)
with open(out, 'w') as f:
    f.write('# rocprofv3 --pmc per-kernel averages (per dispatch)\n\n')
    counters = sorted({c for v in agg.values() for c in v})
    f.write('| kernel | dispatches | ' + ' | '.join(counters) + ' |\n|---|---:|' + '---:|' * len(counters) + '\n')
    rows = sorted(agg.items(), key=lambda kv: -sum(sum(x) for x in kv[1].values()))
    for name, v in rows[:40]:
        n = max(len(x) for x in v.values())
        short = name if len(name) < 90 else name[:87] + '...'
        f.write(f'| `{short}` | {n} | ' + ' | '.join(f'{sum(v[c]) / len(v[c]):.4g}' if c in v else '-' for c in counters) + ' |\n')
print('wrote', out)
