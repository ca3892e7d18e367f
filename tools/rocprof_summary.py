"""Turn a rocprofv3 --kernel-trace --stats results database (rocpd sqlite) into a small text summary for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db profiles/r01_bench_kernel_stats.md "command line"
"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ''
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
tot = sum(r[2] for r in rows)
with open(out, 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n')
    f.write(f'total kernel time: {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches (durations in microseconds)\n\n')
    f.write('| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n')
    for name, calls, total, avg, pct in rows:
        short = name if len(name) < 110 else name[:107] + '...'
        f.write(f'| `{short}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f} |\n')
print('wrote', out, len(rows), 'kernels')
