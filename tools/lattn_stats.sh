# kernel times of the fused linear-attention block (tools/bench_lattn_fused.py --block-only) -> gpurun_out/lattn_kernel_stats.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_l
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o x --output-format rocpd -- python $R/tools/bench_lattn_fused.py ${1:-8} ${2:-40} --block-only > /tmp/prof_l.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/prof_l -name "*.db" | head -1) $R/gpurun_out/lattn_kernel_stats.md "rocprofv3 --kernel-trace --stats -- tools/bench_lattn_fused.py ${1:-8} ${2:-40} --block-only"
grep -E "with grad|fused " /tmp/prof_l.log | cut -c1-260
grep -E "lattn|kernel \|" $R/gpurun_out/lattn_kernel_stats.md | cut -c1-150
