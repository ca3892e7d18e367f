# same-box A/B of the tiled MFMA softmax attention of the mid spatial block (csrc/attention.hip: attn_fwd_mfma_tiled_kernel, attn_bwd_mfma_tiled_kernel):
#   WDNO_DEBUG=0    default        WDNO_DEBUG=69   forward and backward on the thread-per-row kernels (round 4 / start of round 5)
# ms per smoke training step (graph replay, 40 timed steps) and graph-replayed sampling steps/s at batch 8; two alternating repetitions
cd $GRAFT_REPO_ROOT
run () { WDNO_DEBUG=$1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('WDNO_DEBUG=$1', d['ms_per_step'], 'ms per training step', d['config']['step_launch'][:16])"; }
samp () { WDNO_DEBUG=$1 python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bench
from wdno_amd import diffusion_core as K
dev = 'cuda'
dif = bench.build_model(dev)
r = bench.sampling_leg(dif, dev, 8, 60, roofline=False)
print('WDNO_DEBUG=' + os.environ.get('WDNO_DEBUG', '0'), r['batch8']['graph_steps_per_sec'], 'sample steps/s (batch 8, graph)')
PY
}
for rep in 1 2; do
run 0
run 69
samp 0
samp 69
done
