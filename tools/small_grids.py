"""Kernels of a rocprofv3 kernel trace (rocpd sqlite) that run on few workgroups yet take long: python tools/small_grids.py <db> [max_blocks] [min_us]"""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
maxb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
minus = float(sys.argv[3]) if len(sys.argv) > 3 else 12.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q = (f"select s.kernel_name, (d.grid_size_x/d.workgroup_size_x)*(d.grid_size_y/d.workgroup_size_y)*(d.grid_size_z/d.workgroup_size_z) as nb, "
     f"count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name, nb "
     f"having nb <= {maxb} and avg(d.end-d.start)/1e3 >= {minus} order by 5 desc")
for name, nb, n, avg, tot in c.execute(q):
    print(f'{int(nb):6d} blocks  {n:5d} x {avg:8.1f} us = {tot:9.1f} us  {name[:100]}')
