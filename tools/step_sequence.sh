# Ordered kernel list of one eager training step:  bash tools/step_sequence.sh [workload (default smoke; burgers, burgers-bf16)] [out file under gpurun_out/]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-smoke}
OUT=${2:-step_sequence_eager.txt}
rm -rf /tmp/prof_e
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_e -o x --output-format rocpd -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extras --eager > /tmp/prof_e.log 2>&1
mkdir -p $R/gpurun_out
python $R/tools/step_sequence.py $(find /tmp/prof_e -name "*.db" | head -1) $R/gpurun_out/$OUT
