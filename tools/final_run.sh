cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > gpurun_out/final/gpu_tests.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > gpurun_out/final/smoke.log 2>&1
( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/final/bench.json 2>gpurun_out/final/bench.err
( timeout 300 python bench.py --workload burgers --no-cpu-baseline --no-extras 2>&1 | tail -1 ) > gpurun_out/final/bench_burgers.json
( timeout 300 python bench.py --workload burgers-bf16 --no-cpu-baseline --no-extras 2>&1 | tail -1 ) > gpurun_out/final/bench_burgers_bf16.json
( timeout 900 python tools/parity_report.py 2>&1 | tail -15 ) > gpurun_out/final/parity.log 2>&1; cp gpurun_out/parity_report.json gpurun_out/final/ 2>/dev/null
bash tools/run_profile_r02.sh > gpurun_out/final/profile.log 2>&1
tail -3 gpurun_out/final/gpu_tests.log; cat gpurun_out/final/smoke.log; cut -c1-400 gpurun_out/final/bench.json
