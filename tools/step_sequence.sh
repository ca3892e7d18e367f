cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_e
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_e -o x --output-format rocpd -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --eager > /tmp/prof_e.log 2>&1
python $R/tools/step_sequence.py $(find /tmp/prof_e -name "*.db" | head -1) $R/gpurun_out/step_sequence_eager.txt
