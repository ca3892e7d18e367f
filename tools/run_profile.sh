# rocprofv3 kernel trace of the bench command -> gpurun_out/p/kernel_stats.md (small; the .db files stay on the box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats -d /tmp/prof -o x --output-format rocpd -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/prof.log 2>&1
tail -1 /tmp/prof.log | cut -c1-200
mkdir -p $R/gpurun_out/p
python $R/tools/rocprof_summary.py $(find /tmp/prof -name "*.db" | head -1) $R/gpurun_out/p/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
python $R/tools/gpu_busy.py $(find /tmp/prof -name "*.db" | head -1) 2100
