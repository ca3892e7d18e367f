"""Burgers base-resolution DDPM (BASELINE.json configs[0]/[1] shapes) on one MI355X: training steps/s and p_sample steps/s.
Unet2D(dim=128, (1,2,4,8), channels=9, groups=1) + GaussianDiffusion(seq_length=(64,64), cosine, T=1000), fp32.
Not part of bench.py's JSON line (that is the smoke workload the metric is quoted on); numbers go to DESIGN.md."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wdno_amd import _lib, ops, tree_path
for t in ('third_party', 'smoke', 'burgers'):
    sys.path.insert(0, tree_path(t))
from ddpm_burgers.unet import Unet2D
from ddpm_burgers.diffusion_1d import GaussianDiffusion
from wdno_amd.trainer import TrainStep, cosine_annealing_lr

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--steps', type=int, default=8)
ap.add_argument('--profile', action='store_true')
args = ap.parse_args()
_lib.load()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = Unet2D(dim=128, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
dif = GaussianDiffusion(net, seq_length=(64, 64), padded_shape=[41, 60], ori_shape=[81, 120], loss_layer_weight=torch.ones(1, 9, 1, 1),
                        is_condition_pad=True, is_condition_u0=True, is_condition_f=True, beta_schedule='cosine', timesteps=1000).to(dev)
ts = TrainStep(dif, lr=1e-4, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=lambda b, s: cosine_annealing_lr(b, s, 10000))
x = (torch.randn(args.batch, 9, 64, 64) * 0.5).to(dev)
for _ in range(3):
    ts.step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss, _ = ts.step(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(f'burgers train: batch {args.batch}  {dt*1e3:.2f} ms/step  {1/dt:.2f} steps/s  {args.batch/dt:.1f} samples/s  loss {float(loss):.4f}')
if args.profile:
    ops.PROFILE = {}
    ts.step(x); torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    for k, evs in sorted(prof.items(), key=lambda kv: -sum(e0.elapsed_time(e1) for e0, e1, _ in kv[1])):
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs); fl = sum(f for _, _, f in evs)
        print(f'   {k:34s} {len(evs):4d} launches {ms:7.2f} ms  {fl/ms/1e9 if ms else 0:7.1f} TF/s')
with torch.no_grad():
    xs = torch.randn(args.batch, 9, 64, 64, device=dev)
    for t in (500, 499):
        xs = dif.p_sample(xs, t)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        xs = dif.p_sample(xs, 400 - i)[0]
    torch.cuda.synchronize()
    ds = (time.perf_counter() - t0) / 10
print(f'burgers p_sample: batch {args.batch}  {ds*1e3:.2f} ms/step  {1/ds:.1f} steps/s')
