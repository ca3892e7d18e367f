# kernel times of the fused temporal-attention block (tools/bench_tattn.py) -> gpurun_out/tattn_kernel_stats.md, then a short bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_t
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o x --output-format rocpd -- python $R/tools/bench_tattn.py "$@" > /tmp/prof_t.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/prof_t -name "*.db" | head -1) $R/gpurun_out/tattn_kernel_stats.md "rocprofv3 --kernel-trace --stats -- tools/bench_tattn.py $*"
tail -3 /tmp/prof_t.log
grep -E "tattn|kernel \|" $R/gpurun_out/tattn_kernel_stats.md | cut -c1-160
