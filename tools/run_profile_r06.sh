# Round-6 profiles -> gpurun_out/p6/ (copy the .md / .json summaries to profiles/r06_*). Counters only with --kernel-trace, one rocprofv3
# run per counter group (MI355X_MICROARCH.md, HBM / rocprofv3 section).   bash tools/run_profile_r06.sh [quick]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/p6
mkdir -p $O
SMOKE="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
ONE="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
run_stats () {  # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x --output-format rocpd -- "$@" > /tmp/prof_$name.log 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/prof_$name -name "*.db" | head -1) $O/${name}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $*"
}
run_pmc () {    # name, command... : FETCH_SIZE, WRITE_SIZE, two SQ groups in separate passes
  name=$1; shift
  rm -rf /tmp/pmc_${name}_*
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_${name}_a -o a --output-format rocpd -- "$@" > /tmp/pmc_a.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_${name}_b -o b --output-format rocpd -- "$@" > /tmp/pmc_b.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d /tmp/pmc_${name}_c -o c --output-format rocpd -- "$@" > /tmp/pmc_c.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_${name}_d -o d --output-format rocpd -- "$@" > /tmp/pmc_d.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pmc_${name}_a /tmp/pmc_${name}_b /tmp/pmc_${name}_c /tmp/pmc_${name}_d -name "*.db") $O/${name}_pmc.md
}
t0=$(date +%s)
run_stats smoke $SMOKE; echo "smoke stats $(( $(date +%s) - t0 ))s"
python $R/tools/timeline_gaps.py $(find /tmp/prof_smoke -name "*.db" | head -1) $O/smoke_timeline.md 0.3
run_stats sampling python $R/tools/profile_sampling.py 6; echo "sampling stats $(( $(date +%s) - t0 ))s"
run_stats sr_sampling python $R/tools/profile_sr.py; echo "sr stats $(( $(date +%s) - t0 ))s"
if [ "$1" != "quick" ]; then
run_pmc smoke $ONE; echo "smoke pmc $(( $(date +%s) - t0 ))s"
python $R/tools/pmc_traffic_json.py $O/smoke_pmc.md $O/pmc_traffic.json
run_pmc tattn python $R/tools/bench_tattn.py; echo "tattn pmc $(( $(date +%s) - t0 ))s"
run_pmc wgrad python $R/tools/ab_wgrad_r6.py l0; echo "wgrad pmc $(( $(date +%s) - t0 ))s"
run_stats dwt python $R/tools/bench_dwt.py; echo "dwt stats $(( $(date +%s) - t0 ))s"
run_stats burgers_bf16 python $R/bench.py --workload burgers-bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-extras; echo "bf16 stats $(( $(date +%s) - t0 ))s"
run_stats burgers_b16 python $R/bench.py --workload burgers --steps 6 --warmup 1 --no-cpu-baseline --no-extras; echo "b16 stats $(( $(date +%s) - t0 ))s"
run_stats smoke_bf16 python $R/bench.py --workload smoke-bf16 --steps 4 --warmup 1 --no-cpu-baseline --no-extras; echo "smoke bf16 stats $(( $(date +%s) - t0 ))s"
python $R/tools/parity_report.py --out $O/parity_report.json > $O/parity_report.log 2>&1; echo "parity $(( $(date +%s) - t0 ))s"
python $R/bench.py --steps 100 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench $(( $(date +%s) - t0 ))s"
fi
ls -la $O
