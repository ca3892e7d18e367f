# kernel timeline of the smoke training step -> gpurun_out/timeline.md (+ kernel stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_t
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o x --output-format rocpd -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras "$@" > /tmp/prof_t.log 2>&1
DB=$(find /tmp/prof_t -name "*.db" | head -1)
python $R/tools/timeline_gaps.py $DB $R/gpurun_out/timeline.md 0.6
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/timeline_kernel_stats.md "rocprofv3 --kernel-trace --stats -- bench.py --steps 4 --warmup 2 $*"
