"""bench.py -- WDNO hot-path benchmark on MI355X.

Workload (BASELINE.json configs[2]/[3], the configuration the metric is quoted on): 2-D smoke base-resolution DDPM,
Unet3D_with_Conv3D(dim=64, dim_mults=(1,2,4), channels=42) + GaussianDiffusion(image_size=40, frames=24, T=1000) exactly as
smoke/train_2d.py:94-121 builds it, fp32, synthetic wavelet-coefficient tensors [8, 24, 42, 40, 40] per GPU resident in
HBM. One "step" = one full training step of the hot path on one batch: q_sample + conditioning, U-Net forward,
loss, U-Net backward, gradient all-reduce (N > 1), global-norm clip, Adam, EMA.  value = rank-steps / s over all GPUs
(weak scaling: 8 samples per GPU per step). The DDPM sampling step rate (U-Net forward + posterior update + condition
re-imposition) is measured after the timed region and reported alongside.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0     # v_mfma_f32_32x32x16_f16, dense (the 3 x fp16-split kernels spend 3 MFMA flop per algorithmic flop)
PEAK_HBM_TBS = 8.0


def build_model(device, batch):
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        sys.path.insert(0, tree_path(t))
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion
    torch.manual_seed(0)                      # identical replicas on every rank
    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    rescaler = torch.linspace(1.0, 22.0, 42).reshape(1, 1, 42, 1, 1)     # stand-in for data_2d.py:143-147 (only its mean matters)
    dif = GaussianDiffusion(net, rescaler, True, True, True, False, 'bior1.3', 'zero', (18, 34, 34), (32, 64, 64),
                            image_size=40, frames=24, timesteps=1000, sampling_timesteps=250, loss_type='l2')
    return dif.to(device)


def cpu_baseline(seconds_budget=25.0):
    """The oracle (CPU restatement, torch fp32 on the host cores) doing the same training step on a bounded sample."""
    from oracle import diffusion_ref as D, unet_ref as U
    torch.manual_seed(0)
    cores = min(os.cpu_count() or 1, 32)       # torch CPU convolutions stop scaling (and regress) beyond a few dozen threads
    torch.set_num_threads(cores)
    from wdno_amd import tree_path
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and not k.endswith('freqs') else v) for k, v in net.state_dict().items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.99))
    buf = D.make_buffers('sigmoid', 1000)
    lw = torch.linspace(1.0, 22.0, 42).reshape(1, 1, 42, 1, 1)
    model = lambda x, t: U.unet3d_forward(sd, x, t, dim=64, dim_mults=(1, 2, 4), groups=8)
    g = torch.Generator().manual_seed(1)
    times = []
    b = 1
    for it in range(3):
        x0 = torch.randn(b, 24, 42, 40, 40, generator=g) * 0.5
        noise = torch.randn(b, 24, 42, 40, 40, generator=g)
        t = torch.randint(0, 1000, (b,), generator=g)
        t0 = time.perf_counter()
        loss = D.smoke_p_losses(model, buf, x0, t, noise, padded_shape=(18, 34, 34), loss_layer_weight=lw)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        times.append(time.perf_counter() - t0)
        if sum(times) > seconds_budget:
            break
    per_sample = min(times[1:]) if len(times) > 1 else times[0]
    return {'value': 1.0 / (per_sample * 8), 'unit': 'steps/s (8-sample steps)', 'cores': cores, 'kind': 'port',
            'sample': f'{len(times)} training steps at batch 1 of the same [24,42,40,40] workload on the host CPU (oracle/), best {per_sample:.2f} s per sample; '
                      'value = 1 / (8 x that)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8, help='samples per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sample-steps', type=int, default=5)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)      # "nccl" is RCCL on ROCm
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback on the product path)'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    from wdno_amd import _lib, ops
    from wdno_amd.trainer import TrainStep, multistep_lr
    _lib.load()
    dif = build_model(device, args.batch)
    ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    batch = (torch.randn(args.batch, 24, 42, 40, 40, generator=g) * 0.5).to(device)      # resident in HBM before timing

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for _ in range(args.warmup):
        loss, _ = ts.step(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, gn = ts.step(batch)
        losses.append(loss)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    final_loss = float(losses[-1]) if losses else float('nan')

    # ---- DDPM sampling step rate (not part of `value`)
    x = torch.randn(args.batch, 24, 42, 40, 40, device=device)
    init = torch.randn(args.batch, 24, 40, 40, device=device)
    control = torch.randn(args.batch, 24, 16, 40, 40, device=device)
    from wdno_amd import diffusion_core as K
    desc = dif._desc(tuple(x.shape), dif.padded_shape)
    src = dif._condition_source(tuple(x.shape), device, init, control, None)
    with torch.no_grad():
        for _ in range(2):
            x, _ = dif.p_sample(tuple(x.shape), x, 500)
            x = K.apply_cond(x, src, desc)
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        for i in range(args.sample_steps):
            x, _ = dif.p_sample(tuple(x.shape), x, 500 - i)
            x = K.apply_cond(x, src, desc)
        torch.cuda.synchronize()
        sample_elapsed = time.perf_counter() - s0

    # ---- per-launch HIP-event timing of the convolution kernels over one extra step -> roofline of the dominant kernel
    roofline = None
    if rank == 0:
        ops.PROFILE = {}
        ts.step(batch)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        agg = {}
        for key, evs in prof.items():
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
            fl = sum(f for _, _, f in evs)
            agg[key] = (ms, fl, len(evs))
        if agg:
            dom = max(agg, key=lambda k: agg[k][0])
            ms, fl, n = agg[dom]
            achieved = fl / (ms * 1e-3) / 1e12
            peak = PEAK_F16_MFMA_TFLOPS if 'h3' in dom else PEAK_F32_MFMA_TFLOPS
            traffic = None          # HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_traffic.json)
            try:
                # profiling key 'conv_fwd_h3d_kernel<256,64>' -> mangled symbol prefix 'conv_fwd_h3d_kernelILi256ELi64E'
                fam, dims = dom.split('<')
                dims = dims.rstrip('>').split(',')
                sym = fam + 'I' + ''.join(f'Li{d}E' for d in dims) if all(d.isdigit() for d in dims) else None
                with open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')) as f:
                    for kname, rec in json.load(f)['kernels'].items():
                        if sym and sym in kname:          # entries are ordered by total time: the first match is the main instantiation
                            traffic = round(rec['hbm_bytes_per_launch'])
                            break
            except Exception:
                traffic = None
            roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                        'frac': round(achieved / peak, 4), 'traffic': traffic, 'launches_per_step': n,
                        'frac_of_fp32_equivalent_ceiling': round(achieved / (peak / 3), 4) if 'h3' in dom else None,
                        'note': ('fp32-equivalent 3 x fp16-split MFMA: 3 matrix flop per algorithmic flop, so frac <= 0.333; '
                                 'the exact-fp32 MFMA peak is 157.3 TFLOP/s') if 'h3' in dom else 'exact-fp32 MFMA',
                        'avg_launch_ms': round(ms / n, 4), 'gflop_per_launch': round(fl / n / 1e9, 3),
                        'conv_ms_per_step': {k: round(v[0], 3) for k, v in agg.items()}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
        except Exception as e:      # the baseline is informative only
            cpu = {'error': repr(e)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            'metric': 'diffusion train steps/sec, 2D smoke U-Net (8 samples per GPU per step)',
            'value': round(world * args.steps / elapsed, 4), 'unit': 'steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'smoke base-resolution DDPM train step: Unet3D_with_Conv3D(dim=64,(1,2,4),ch=42) on wavelet tensor '
                                   f'[{args.batch},24,42,40,40] per GPU, fp32 in/out, convolutions on the fp32-equivalent 3 x fp16-split MFMA path (small ones exact-fp32 MFMA), Adam+clip+EMA',
                       'global_batch': args.batch * world, 'parallelism': f'dp{world}', 'grad_allreduce_MB': 95.3 if world > 1 else 0},
            'samples_per_sec': round(world * args.steps * args.batch / elapsed, 3),
            'ddpm_sample_steps_per_sec': round(args.sample_steps / sample_elapsed, 3),
            'final_loss': final_loss,
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
