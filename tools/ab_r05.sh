# same-box A/Bs of round 5 (ms per smoke training step, graph replay, 40 timed steps; two alternating repetitions):
#   default            stem skips the all-zero / out-of-grid reduction stages (zero-box hint), two-level sums per tap row
#   WDNO_DEBUG=57      the stem runs every stage (same bits)
#   WDNO_DEBUG=58      the GroupNorm forward statistics pass is launched twice (same results): the difference to the default is what the pass
#                      costs inside the step = the upper bound of what statistics from the producing convolution's epilogue could save
#                      (VERDICT r4 item 7a). (NOT launching it is no measurement: garbage statistics turn the activations into NaN / zeros
#                      and the power-limited convolutions clock up -- 31.6 -> 26.4 ms was measured that way and means nothing.)
#   WDNO_DEBUG=59      the GroupNorm backward reduction pass twice; partial sums from the data-gradient epilogue (item 7b) would still
#                      read x there, so their bound is about half of this difference
cd $GRAFT_REPO_ROOT
run () { WDNO_DEBUG=$1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('WDNO_DEBUG=$1', d['ms_per_step'], 'ms', d['config']['step_launch'][:16])"; }
for rep in 1 2; do
run 0
run 57
run 58
run 59
done
