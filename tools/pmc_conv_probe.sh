# LDS bank-conflict counters of the convolution micro-benchmark for a few debug switches -> gpurun_out/pmc_conv_probe.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dbg in 0 15 16 8; do
  rm -rf /tmp/pc_$dbg
  WDNO_DEBUG=$dbg timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d /tmp/pc_$dbg -o x --output-format rocpd -- python $R/tools/bench_conv.py "l0 3x3x3 64->64" > /tmp/pc_$dbg.log 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pc_$dbg -name "*.db") /tmp/pc_$dbg.md
  echo "== WDNO_DEBUG=$dbg"; grep "conv_fwd_h3\|conv_wgrad" /tmp/pc_$dbg.md | cut -c1-260
done > $R/gpurun_out/pmc_conv_probe.txt 2>&1
