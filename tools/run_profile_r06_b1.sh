# Round 6, late: kernel stats of the packing pipeline, batch-1 sampling, and the step timeline with (after -> before) gap pairs -> gpurun_out/r6i/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6i
mkdir -p $O
run_stats () {
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x --output-format rocpd -- "$@" > /tmp/prof_$name.log 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/prof_$name -name "*.db" | head -1) $O/${name}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $*" || tail -20 /tmp/prof_$name.log
}
run_stats pack python $R/tools/profile_pack.py
head -14 $O/pack_kernel_stats.md | cut -c1-200
run_stats smoke python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras
python $R/tools/timeline_gaps.py $(find /tmp/prof_smoke -name "*.db" | head -1) $O/smoke_timeline.md 0.3 > /dev/null
sed -n 36,90p $O/smoke_timeline.md | cut -c1-200
