cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6i
mkdir -p $O
rm -rf /tmp/prof_smoke
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_smoke -o x --output-format rocpd -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras > /tmp/prof_smoke.log 2>&1
tail -2 /tmp/prof_smoke.log | cut -c1-300
python $R/tools/timeline_gaps.py $(find /tmp/prof_smoke -name "*.db" | head -1) $O/smoke_timeline_replays.md 0.7 > /dev/null
cat $O/smoke_timeline_replays.md | cut -c1-200
