cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in 1 0; do
rm -rf /tmp/pa$a
WDNO_ACT16=$a timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa$a -o x --output-format rocpd -- python $R/bench.py --workload burgers-bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pa$a.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pa$a -name "*.db" | head -1) $R/gpurun_out/act16_$a.md "ACT16=$a burgers-bf16"
done
