# timing and FETCH_SIZE of the weight-gradient kernels, round-robin vs XCD-grouped items
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 python $R/tools/bench_wgrad_xcd.py 2>&1 | grep -v amdgpu.ids
rm -rf /tmp/pmcw
ITERS=1 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcw -o p --output-format csv -- python $R/tools/bench_wgrad_xcd.py > /tmp/pmcw.log 2>&1
f=$(find /tmp/pmcw -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'wgrad_h3' in r['Kernel_Name'] and 'reduce' not in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
for r in rows:
    print(r['Kernel_Name'][:48], 'grid', r.get('Grid_Size'), 'fetch MB (2x corrected)', round(2 * float(r['Counter_Value']) * 1024 / 1e6, 1))
PY
