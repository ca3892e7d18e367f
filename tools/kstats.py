"""Per-kernel call counts and average durations from a rocprofv3 (rocpd sqlite) kernel trace: python tools/kstats.py <db> [substring ...]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3 from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f'total {tot / 1e3:.2f} ms, {sum(r[1] for r in rows)} dispatches')
for name, n, t in rows:
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
        continue
    print(f'{n:6d} {t:10.1f} us  avg {t / n:8.2f}  {name[:110]}')
