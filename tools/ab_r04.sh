# same-box A/B of the round-4 changes (ms per smoke training step): default (fused temporal attention forward + backward, graph replay),
# the 64-channel linear attention blocks layer by layer in training steps (ops.FUSED_LATTN_BWD = False), the temporal ones (ops.FUSED_TATTN_BWD = False), both
cd $GRAFT_REPO_ROOT
run () { python - "$@" <<'P'
import json, runpy, sys, io, contextlib
knob, args = sys.argv[1], sys.argv[2:]
import wdno_amd.ops as o
if knob == 'layers': o.FUSED_TATTN_BWD = False
if knob == 'lattn_layers': o.FUSED_LATTN_BWD = False
if knob == 'all_layers': o.FUSED_TATTN_BWD = o.FUSED_LATTN_BWD = False
sys.argv = ['bench.py', '--steps', '40', '--warmup', '5', '--no-cpu-baseline', '--no-extras'] + args
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path('bench.py', run_name='__main__')
    except SystemExit:
        pass
line = [l for l in buf.getvalue().splitlines() if l.startswith('{')][-1]
d = json.loads(line)
print(knob, args, d['ms_per_step'], 'ms', d['config'].get('step_launch', '')[:20])
P
}
for rep in 1 2; do
run default
run lattn_layers
run layers
run all_layers
done
