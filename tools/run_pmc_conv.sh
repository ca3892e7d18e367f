# PMC pass over the conv micro-benchmark (one geometry): LDS conflicts / activity of the forward kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc1 /tmp/pmc2
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc1 -o p1 --output-format rocpd -- python $R/tools/bench_conv.py "l0 3x3x3 64->64" > /tmp/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY -d /tmp/pmc2 -o p2 --output-format rocpd -- python $R/tools/bench_conv.py "l0 3x3x3 64->64" > /tmp/pmc2.log 2>&1
tail -3 /tmp/pmc1.log
mkdir -p $R/gpurun_out/pmcconv
python $R/tools/pmc_summary.py $(find /tmp/pmc1 /tmp/pmc2 -name "*.db") $R/gpurun_out/pmcconv/pmc.md
grep -E "kernel \||conv_fwd_h3" $R/gpurun_out/pmcconv/pmc.md | cut -c1-400
