cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6k
mkdir -p $O
for n in 4 12; do
  rm -rf /tmp/prof_s$n
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s$n -o x --output-format rocpd -- python $R/bench.py --steps $n --warmup 1 --no-cpu-baseline --no-extras > /tmp/prof_s$n.log 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/prof_s$n -name "*.db" | head -1) $O/smoke_s${n}_kernel_stats.md "steps $n" > /dev/null
done
python - <<PY
import re
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r'\| \`(.*)\` \| (\d+) \| ([\d.]+) \|', l)
        if m: d[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return d
a, b = load('$O/smoke_s4_kernel_stats.md'), load('$O/smoke_s12_kernel_stats.md')
rows = []
for k in b:
    c0, t0 = a.get(k, (0, 0.0))
    c1, t1 = b[k]
    if c1 - c0: rows.append(((c1 - c0) / 8, (t1 - t0) / 8, k))
rows.sort(key=lambda r: -r[1])
tot_n = sum(r[0] for r in rows); tot_t = sum(r[1] for r in rows)
print(f'per replayed step: {tot_n:.1f} launches, {tot_t / 1e3:.3f} ms of kernel time')
for n, t, k in rows: print(f'{n:7.2f} {t:9.1f} us  {k[:100]}')
PY
