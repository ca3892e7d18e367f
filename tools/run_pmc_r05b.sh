# PMC passes (separate rocprofv3 runs per counter group, --kernel-trace only) for the kernels added in the second half of round 5:
# the wide attention blocks (tools/bench_tattn_wide.py) and the tiled spatial attention (tools/bench_spatial_attn.py) -> gpurun_out/p5b/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/p5b
mkdir -p $O
run_pmc () {
  name=$1; shift
  rm -rf /tmp/pmc_${name}_*
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_${name}_a -o a --output-format rocpd -- "$@" > /tmp/pmc_a.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_${name}_b -o b --output-format rocpd -- "$@" > /tmp/pmc_b.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d /tmp/pmc_${name}_c -o c --output-format rocpd -- "$@" > /tmp/pmc_c.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_${name}_d -o d --output-format rocpd -- "$@" > /tmp/pmc_d.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pmc_${name}_a /tmp/pmc_${name}_b /tmp/pmc_${name}_c /tmp/pmc_${name}_d -name "*.db") $O/${name}_pmc.md
}
run_stats () {
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x --output-format rocpd -- "$@" > /tmp/prof_$name.log 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/prof_$name -name "*.db" | head -1) $O/${name}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $*"
}
run_stats wide_attn python $R/tools/bench_tattn_wide.py
run_pmc wide_attn python $R/tools/bench_tattn_wide.py
run_stats spatial_attn python $R/tools/bench_spatial_attn.py
run_pmc spatial_attn python $R/tools/bench_spatial_attn.py
ls -la $O
