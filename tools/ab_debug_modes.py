"""Same-process A/B of the graph-replayed smoke training step (and, optionally, the batch-8 / batch-1 sampling step) under library debug modes:
alternating repetitions of `--steps` steps each, one captured graph per arm (the kernel-side switch is read at launch / capture time).
    python tools/ab_debug_modes.py 0 71 [--steps 40] [--reps 3] [--sampling]
Round 6: 71 = GroupNorm finalize kernels as one block per sample (round 5); 70 = the round-5 item plan of the window weight gradient."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import ops, diffusion_core as K
from wdno_amd.trainer import TrainStep, multistep_lr

args = sys.argv[1:]
steps = int(args[args.index('--steps') + 1]) if '--steps' in args else 40
reps = int(args[args.index('--reps') + 1]) if '--reps' in args else 3
modes = [int(a) for a in args if a.lstrip('-').isdigit() and (args.index(a) == 0 or args[args.index(a) - 1] not in ('--steps', '--reps'))]
if '--nodefer' in args:
    modes.append(-1)          # arm -1: debug 0 with ops.DEFER_WGRAD_REDUCE = False (one split reduction behind every weight-gradient kernel)
if '--norowdefer' in args:
    modes.append(-2)          # arm -2: debug 0 with ops.DEFER_ROW_SUMS = False (GroupNorm tails and bias sums launched where they arise)
dev = torch.device('cuda', 0)
lib = ops._lib_()
dif = bench.build_model(dev, 8)
ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
x = (torch.randn(8, 24, 42, 40, 40) * 0.5).to(dev)
for _ in range(3):
    ts.step(x)
caps = {}
for m in modes:
    lib.wdno_set_debug(max(m, 0))
    ops.DEFER_WGRAD_REDUCE = m != -1
    ops.DEFER_ROW_SUMS = m != -2
    ts._cap = None
    ts.capture(x, warmup=1)
    caps[m] = ts._cap
ops.DEFER_WGRAD_REDUCE = True
ops.DEFER_ROW_SUMS = True
res = {m: [] for m in modes}
for r in range(reps):
    for m in modes:
        lib.wdno_set_debug(max(m, 0))
        ts._cap = caps[m]
        for _ in range(3):
            ts.step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step(x)
        torch.cuda.synchronize()
        res[m].append(round((time.perf_counter() - t0) / steps * 1e3, 3))
lib.wdno_set_debug(0)
for m in modes:
    print(f'train step, debug {m}: ms per step {res[m]}  min {min(res[m])}')
if '--sampling' in args:
    ts._cap = None
    for b in (8, 1):
        shape = (b, 24, 42, 40, 40)
        desc = dif._desc(shape, dif.padded_shape)
        src = dif._condition_source(shape, dev, torch.randn(b, 24, 40, 40, device=dev), torch.randn(b, 24, 16, 40, 40, device=dev), None)
        out = {m: [] for m in modes}
        with torch.no_grad():
            for r in range(reps):
                for m in modes:
                    lib.wdno_set_debug(m)
                    K._graph_cache.pop(dif, None)
                    sg = K._step_graph(dif, shape, desc, False, False, dev)
                    sg.src.copy_(src)
                    sg.x.copy_(K.apply_cond(torch.randn(shape, device=dev), src, desc))
                    for _ in range(3):
                        sg.graph.replay()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(steps):
                        sg.t.fill_(500 - i)
                        sg.noise.normal_()
                        sg.graph.replay()
                    torch.cuda.synchronize()
                    out[m].append(round(steps / (time.perf_counter() - t0), 2))
        lib.wdno_set_debug(0)
        for m in modes:
            print(f'sampling batch {b}, debug {m}: steps/s {out[m]}  max {max(out[m])}')
