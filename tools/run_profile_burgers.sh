# rocprofv3 kernel trace of the Burgers workload (batch 64) -> gpurun_out/p/kernel_stats_burgers.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/profb
rocprofv3 --kernel-trace --stats -d /tmp/profb -o x --output-format rocpd -- python $R/tools/bench_burgers.py --batch ${1:-64} --steps 4 > /tmp/profb.log 2>&1
tail -2 /tmp/profb.log
mkdir -p $R/gpurun_out/p
python $R/tools/rocprof_summary.py $(find /tmp/profb -name "*.db" | head -1) $R/gpurun_out/p/kernel_stats_burgers.md "rocprofv3 --kernel-trace --stats -- python tools/bench_burgers.py --batch ${1:-64} --steps 4"
