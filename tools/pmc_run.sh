# PMC counters of one command, one rocprofv3 pass per counter group (MI355X_MICROARCH.md: counters only with --kernel-trace):
#   bash tools/pmc_run.sh <name> <command...>   -> gpurun_out/r4/<name>_pmc.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
name=$1; shift
rm -rf /tmp/pmc_${name}_*
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_${name}_a -o a --output-format rocpd -- "$@" > /tmp/pmc_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_${name}_b -o b --output-format rocpd -- "$@" > /tmp/pmc_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d /tmp/pmc_${name}_c -o c --output-format rocpd -- "$@" > /tmp/pmc_c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_${name}_d -o d --output-format rocpd -- "$@" > /tmp/pmc_d.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d /tmp/pmc_${name}_e -o e --output-format rocpd -- "$@" > /tmp/pmc_e.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pmc_${name}_a /tmp/pmc_${name}_b /tmp/pmc_${name}_c /tmp/pmc_${name}_d /tmp/pmc_${name}_e -name "*.db") $O/${name}_pmc.md
tail -3 /tmp/pmc_e.log
