# rocprofv3 kernel stats of a few smoke training steps -> gpurun_out/quick_kernel_stats.md (see run_profile_r02.sh for the full set)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_q
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o x --output-format rocpd -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras "$@" > /tmp/prof_q.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/prof_q -name "*.db" | head -1) $R/gpurun_out/quick_kernel_stats.md "rocprofv3 --kernel-trace --stats -- bench.py --steps 4 --warmup 1 $*"
