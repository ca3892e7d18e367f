# PMC passes over one training step of the bench -> gpurun_out/p/pmc.md (per-kernel averages). Counters only with
# --kernel-trace, one rocprofv3 run per counter group (see MI355X_MICROARCH.md, HBM / rocprofv3 section).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
rm -rf /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c
t0=$(date +%s)
rm -rf /tmp/pmc_a2
# FETCH_SIZE and WRITE_SIZE are derived from per-channel TCC counters: together they exceed one pass ("error 38") -> one run each
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_a -o a --output-format rocpd -- $CMD > /tmp/pmc_a.log 2>&1; echo "pass a rc=$? $(( $(date +%s) - t0 ))s"
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_a2 -o a2 --output-format rocpd -- $CMD > /tmp/pmc_a2.log 2>&1; echo "pass a2 rc=$? $(( $(date +%s) - t0 ))s"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d /tmp/pmc_b -o b --output-format rocpd -- $CMD > /tmp/pmc_b.log 2>&1; echo "pass b rc=$? $(( $(date +%s) - t0 ))s"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_c -o c --output-format rocpd -- $CMD > /tmp/pmc_c.log 2>&1; echo "pass c rc=$? $(( $(date +%s) - t0 ))s"
tail -2 /tmp/pmc_a.log | cut -c1-200
mkdir -p $R/gpurun_out/p
python $R/tools/pmc_summary.py $(find /tmp/pmc_a /tmp/pmc_a2 /tmp/pmc_b /tmp/pmc_c -name "*.db") $R/gpurun_out/p/pmc.md
