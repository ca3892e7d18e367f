# Instruction-mix counters of the temporal-attention kernels (tools/bench_attn.py) -> gpurun_out/attn_pmc.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pa_*
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d /tmp/pa_a -o a --output-format rocpd -- python $R/tools/bench_attn.py > /tmp/pa_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE -d /tmp/pa_b -o b --output-format rocpd -- python $R/tools/bench_attn.py > /tmp/pa_b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM -d /tmp/pa_c -o c --output-format rocpd -- python $R/tools/bench_attn.py > /tmp/pa_c.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pa_a /tmp/pa_b /tmp/pa_c -name "*.db") $R/gpurun_out/attn_pmc.md
tail -3 /tmp/pa_a.log /tmp/pa_b.log /tmp/pa_c.log
