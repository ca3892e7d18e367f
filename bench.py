"""bench.py -- WDNO hot-path benchmark on MI355X.

Default workload (BASELINE.json configs[2]/[3], the configuration the metric is quoted on): 2-D smoke base-resolution DDPM,
Unet3D_with_Conv3D(dim=64, dim_mults=(1,2,4), channels=42) + GaussianDiffusion(image_size=40, frames=24, T=1000) exactly as
smoke/train_2d.py:94-121 builds it, fp32, synthetic wavelet-coefficient tensors [8, 24, 42, 40, 40] per GPU resident in
HBM. One "step" = one full training step of the hot path on one batch: q_sample + conditioning, U-Net forward,
loss, U-Net backward, gradient all-reduce (N > 1), global-norm clip, Adam, EMA.  value = rank-steps / s over all GPUs
(weak scaling: 8 samples per GPU per step).

Reported next to it in the same JSON line (never part of `value`):
  sampling : DDPM sampling steps/s of the same model (U-Net forward + posterior update + condition re-imposition), issued launch by
             launch and as replays of one captured HIP graph, at the bench batch and at batch 1;
  dwt      : the four wavelet transforms of the path at BASELINE.json's synthetic shapes, HIP-event timed, GB/s of algorithmic
             bytes (in + out) and the fraction of the 8 TB/s HBM peak;
  burgers  : (--workload burgers / burgers-bf16 make it the main line instead) the Burgers base model's train / sample step;
  roofline : the dominant convolution kernel of the main workload; cpu_baseline: the oracle on the host cores.

    python bench.py --gpus 1 --steps 100 --warmup 5
    python bench.py --gpus N ...          (no launcher: re-executes itself as N ranks under torch.distributed.run on a free port)
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
PEAK_F16_MFMA_TFLOPS = 2500.0     # v_mfma_f32_32x32x16_f16 / _bf16, dense (the 3 x fp16-split kernels spend 3 MFMA flop per algorithmic flop)
PEAK_HBM_TBS = 8.0
PROFILE_JSON = os.path.join(ROOT, 'profiles', 'r06_pmc_traffic.json')


def _trees():
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)


def build_model(device, batch=8, channels=42):
    """The smoke base model exactly as smoke/train_2d.py:94-121 builds it. channels = 42: the reference's five fields (5 x 8 sub-bands + initial
    density + smoke-out); 34: BASELINE's four synthetic fields [B, 4, 32, 64, 64] (smoke/ddpm/utils.py:62-63's rule scaled: 4 x 8 + 2)."""
    _trees()
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion
    torch.manual_seed(0)                      # identical replicas on every rank
    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=channels)
    rescaler = torch.linspace(1.0, 22.0, channels).reshape(1, 1, channels, 1, 1)     # stand-in for data_2d.py:143-147 (only its mean matters)
    dif = GaussianDiffusion(net, rescaler, True, True, True, False, 'bior1.3', 'zero', (18, 34, 34), (32, 64, 64),
                            image_size=40, frames=24, timesteps=1000, sampling_timesteps=250, loss_type='l2')
    return dif.to(device)


def build_burgers(device, grid=(64, 64)):
    """The Burgers base model exactly as burgers/train_ddpm_burgers.py:128-182 builds it (scripts/burgers/train_base_sim.sh). grid (64, 64)
    is the reference-native tensor (41 x 60 coefficients zero-padded); (80, 64) the north-star's synthetic one: the DWT of a
    [2, 160, 128] field fills it completely."""
    _trees()
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion
    torch.manual_seed(0)
    net = Unet2D(dim=128, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
    coef, ori = ([41, 60], [81, 120]) if tuple(grid) == (64, 64) else (list(grid), [2 * grid[0], 2 * grid[1]])
    dif = GaussianDiffusion(net, seq_length=tuple(grid), padded_shape=coef, ori_shape=ori, loss_layer_weight=torch.ones(1, 9, 1, 1),
                            is_condition_pad=True, is_condition_u0=True, is_condition_f=True, beta_schedule='cosine', timesteps=1000)
    return dif.to(device)


def burgers_fields_to_state(fields, rescaler=None):
    """North-star Burgers input on the GPU: fields [B, 2, 160, 128] (u, f) -> ONE fused HIP DWT launch (bior2.4 / periodization, sub-bands
    written in coef_to_tensor order) -> [B, 8, 80, 64] + the condition channel (1-D DWT of u(t=0) and u(t=T) as four horizontal
    stripes, data_burgers_1d.py:51-82) -> [B, 9, 80, 64] / rescaler."""
    from wdno_amd import wavelets as Wv
    b, _, nt, nx = fields.shape
    packed = Wv.dwt_packed(fields, 'bior2.4', 'periodization', 2).reshape(b, 8, nt // 2, nx // 2)
    lo, hi = Wv.DWT1DForward(J=1, mode='periodization', wave='bior2.4')(fields[:, 0, [0, nt - 1], :].contiguous())      # [B, 2, nx / 2] each
    q = nt // 8
    cond = torch.stack((lo[:, 0], hi[0][:, 0], lo[:, 1], hi[0][:, 1]), dim=1)            # [B, 4, nx / 2]
    cond = cond.unsqueeze(2).expand(b, 4, q, nx // 2).reshape(b, 1, nt // 2, nx // 2)
    x = torch.cat((packed, cond), dim=1)
    return x if rescaler is None else x / rescaler


# ---------------------------------------------------------------------------------------------------------------- CPU baseline
def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _median(v):
    v = sorted(v)
    return v[len(v) // 2]


def cpu_baseline(budget_s=30.0):
    """The oracle (CPU restatement of the reference's path, torch fp32 / numpy on the host cores) timed on the GPU box's host: the training
    step and the p_sample step of both models AT THE BENCH BATCH (smoke 8, Burgers 16) where one such step fits what is left of the budget --
    a warm-up at batch 1 (smoke) / 4 (Burgers) pages the code in and predicts the cost -- and the four wavelet transforms (BASELINE.md
    section 3). A leg that cannot afford its real batch is timed on the small batch, scaled, and its key says `..._extrapolated`."""
    from oracle import diffusion_ref as D, unet_ref as U, dwt_ref as R
    import numpy as np
    _trees()
    torch.manual_seed(0)
    cores = min(os.cpu_count() or 1, 32)       # torch CPU convolutions stop scaling (and regress) beyond a few dozen threads
    torch.set_num_threads(cores)
    t_begin = time.perf_counter()
    out = {'cores': cores, 'cpu_model': _cpu_model(), 'kind': 'port'}
    left = lambda: budget_s - (time.perf_counter() - t_begin)

    def once(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    def leg(name, make, small, full, share):
        """make(batch) -> a callable running one step at that batch. One step at `small` (warm-up, untimed), one more (timed: the prediction),
        then one at `full` if `share` of the remaining budget covers the predicted cost."""
        f_small = make(small)
        once(f_small)
        s_small = once(f_small)
        predicted = s_small * full / small
        if predicted <= share * max(left(), 0.0):
            f_full = make(full)
            s_full = once(f_full)
            out[name] = {'batch': full, 'seconds_per_step': round(s_full, 3), 'steps_per_sec': round(1.0 / s_full, 4), 'iters': 1,
                         'warmup': f'2 steps at batch {small} ({s_small:.2f} s the second)'}
            return out[name]['steps_per_sec'], False
        out[name + '_extrapolated'] = {'batch': small, 'seconds_per_step_at_that_batch': round(s_small, 3), f'steps_per_sec_at_batch{full}_extrapolated': round(1.0 / predicted, 4),
                                       'iters': 1, 'note': f'one step at batch {full} was predicted at {predicted:.1f} s, more than this leg\'s share of the {budget_s:.0f} s budget'}
        return round(1.0 / predicted, 4), True

    # ---- smoke
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and not k.endswith('freqs') else v) for k, v in net.state_dict().items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.99))
    buf = D.make_buffers('sigmoid', 1000)
    lw = torch.linspace(1.0, 22.0, 42).reshape(1, 1, 42, 1, 1)
    model = lambda x, t: U.unet3d_forward(sd, x, t, dim=64, dim_mults=(1, 2, 4), groups=8)
    g = torch.Generator().manual_seed(1)

    def smoke_train(b):
        x0 = torch.randn(b, 24, 42, 40, 40, generator=g) * 0.5
        noise = torch.randn(b, 24, 42, 40, 40, generator=g)
        t = torch.randint(0, 1000, (b,), generator=g)

        def step():
            loss = D.smoke_p_losses(model, buf, x0, t, noise, padded_shape=(18, 34, 34), loss_layer_weight=lw)
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
        return step

    def smoke_sample(b):
        x0 = torch.randn(b, 24, 42, 40, 40, generator=g) * 0.5
        noise = torch.randn(b, 24, 42, 40, 40, generator=g)

        def step():
            with torch.no_grad():
                D.smoke_p_sample(model, buf, x0, 500, noise)
        return step
    value, extrapolated = leg('smoke_train_step', smoke_train, 1, 8, 0.55)
    leg('smoke_p_sample_step', smoke_sample, 1, 8, 0.45)
    del sd, params, opt, net

    # ---- Burgers
    from ddpm_burgers.unet import Unet2D
    netb = Unet2D(dim=128, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
    sdb = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in netb.state_dict().items()}
    pb = [v for v in sdb.values() if v.requires_grad]
    optb = torch.optim.Adam(pb, lr=1e-4, betas=(0.9, 0.99))
    bufb = D.make_buffers('cosine', 1000)
    modelb = lambda x, t: U.unet2d_forward(sdb, x, t, dim=128, dim_mults=(1, 2, 4, 8), groups=1)
    flags = dict(pad=True, u0=True, uT=False, f=True)

    def burgers_train(b):
        xb = torch.randn(b, 9, 64, 64, generator=g) * 0.5
        nb = torch.randn(b, 9, 64, 64, generator=g)
        tb = torch.randint(0, 1000, (b,), generator=g)

        def step():
            loss = D.burgers_p_losses(modelb, bufb, xb, tb, nb, padded_shape=[41, 60], loss_layer_weight=torch.ones(1, 9, 1, 1), flags=flags)
            optb.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(pb, 1.0)
            optb.step()
        return step

    def burgers_sample(b):
        xb = torch.randn(b, 9, 64, 64, generator=g) * 0.5
        nb = torch.randn(b, 9, 64, 64, generator=g)

        def step():
            with torch.no_grad():
                tt = torch.full((b,), 500, dtype=torch.long)
                _, xs = D.burgers_model_predictions(modelb, bufb, xb, tt)
                D.posterior_step(bufb, xb, 500, xs.clamp(-1., 1.), nb)
        return step
    leg('burgers_train_step', burgers_train, 4, 16, 0.6)
    leg('burgers_p_sample_step', burgers_sample, 4, 16, 0.6)
    del sdb, pb, optb, netb

    # ---- wavelet transforms (numpy restatement, 1 thread), 1/8 of the synthetic batch: scaled, and named so
    x2 = np.random.default_rng(0).standard_normal((8, 2, 160, 128)).astype(np.float32)
    x3 = np.random.default_rng(1).standard_normal((4, 32, 64, 64)).astype(np.float32)
    yl, yh = R.dwt2(x2, 'bior2.4', 'periodization')
    lll, det = R.dwt3(x3, 'bior1.3')
    for key, fn, scale, nbytes in (('dwt2_fwd', lambda: R.dwt2(x2, 'bior2.4', 'periodization'), 8, 20.97e6),
                                   ('dwt2_inv', lambda: R.idwt2(yl, yh, 'bior2.4', 'periodization'), 8, 20.97e6),
                                   ('dwt3_fwd', lambda: R.dwt3(x3, 'bior1.3'), 8, 38.08e6),
                                   ('dwt3_inv', lambda: R.idwt3(lll, det, 'bior1.3'), 8, 38.08e6)):
        once(fn)
        s_ = _median([once(fn) for _ in range(3)])
        out[key + '_extrapolated'] = {'ms_full_shape_extrapolated': round(s_ * scale * 1e3, 2), 'GB/s': round(nbytes / (s_ * scale) / 1e9, 3), 'iters': 3, 'threads': 1,
                                      'sample': f'1/{scale} of the synthetic batch'}
    out['seconds_total'] = round(time.perf_counter() - t_begin, 1)
    # the JSON contract's required keys: the baseline of the main metric
    how = (f'ONE training step at the bench batch [8,24,42,40,40] (q_sample + conditioning, U-Net forward, loss, backward, clip, Adam) after two warm-up steps at batch 1'
           if not extrapolated else 'one training step at batch 1, scaled to 8 samples (EXTRAPOLATED: the batch-8 step did not fit the budget)')
    out.update(value=value, unit='steps/s (8-sample steps)', extrapolated=extrapolated,
               sample=f'oracle/ (CPU restatement, torch fp32) on {cores} of {os.cpu_count()} hardware threads of {out["cpu_model"]} (torch CPU convolutions regress beyond ~32): '
                      + how + '. Other entries: the p_sample step, the Burgers model at batch 16, the numpy DWT oracle on 1/8 of the synthetic batch (1 thread); keys that end in _extrapolated were scaled.')
    return out


# ---------------------------------------------------------------------------------------------------------------- side legs
def _ev_time(fn, iters, warm=3):
    """Average duration in ms of fn() measured with HIP events on the launch stream."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def dwt_leg(device):
    """W1/W2/W4/W5 at BASELINE.json's synthetic shapes: one fused launch each (csrc/dwt.hip), algorithmic bytes = input + output."""
    from wdno_amd import wavelets
    out = {}
    x2 = torch.randn(64, 2, 160, 128, device=device)
    c2 = wavelets.dwt_packed(x2, 'bior2.4', 'periodization', 2)
    x3 = torch.randn(32, 32, 64, 64, device=device)
    c3 = wavelets.dwt_packed(x3, 'bior1.3', 'zero', 3)
    cases = (('dwt2_per_bior2.4_fwd [64,2,160,128]', lambda: wavelets.dwt_packed(x2, 'bior2.4', 'periodization', 2), (x2.numel() + c2.numel()) * 4),
             ('dwt2_per_bior2.4_inv', lambda: wavelets.idwt_packed(c2, 'bior2.4', 'periodization', 2), (x2.numel() + c2.numel()) * 4),
             ('dwt3_zero_bior1.3_fwd [32,32,64,64]', lambda: wavelets.dwt_packed(x3, 'bior1.3', 'zero', 3), (x3.numel() + c3.numel()) * 4),
             ('dwt3_zero_bior1.3_inv', lambda: wavelets.idwt_packed(c3, 'bior1.3', 'zero', 3), (x3.numel() + c3.numel()) * 4))
    g = torch.cuda.CUDAGraph()
    for name, fn, nbytes in cases:
        ms_call = _ev_time(fn, 50)              # includes the host side of one Python call per transform
        # kernel alone: 20 launches captured in one graph (no host gaps between them)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        ms_k = _ev_time(g.replay, 10) / 20
        out[name] = {'us_per_launch': round(ms_k * 1e3, 2), 'us_per_python_call': round(ms_call * 1e3, 2), 'MB_algorithmic': round(nbytes / 1e6, 2),
                     'GB/s': round(nbytes / (ms_k * 1e-3) / 1e9, 1), 'frac_of_hbm_peak': round(nbytes / (ms_k * 1e-3) / 1e12 / PEAK_HBM_TBS, 4)}
    return out


def sampling_leg(dif, device, batch, steps, barrier=None, roofline=True):
    """DDPM sampling steps/s: the step issued launch by launch vs replays of one captured HIP graph (same arithmetic, bit-equal).
    barrier (N > 1): every rank samples its own batch (no collective on this path, smoke/inference_2d.py:123-152); the timed loops start
    together so that the per-rank rates are concurrent rates."""
    from wdno_amd import diffusion_core as K
    out = {}
    for b in sorted({batch, 1} if barrier is None else {batch}):
        shape = (b, 24, 42, 40, 40)
        x = torch.randn(shape, device=device)
        init = torch.randn(b, 24, 40, 40, device=device)
        control = torch.randn(b, 24, 16, 40, 40, device=device)
        desc = dif._desc(shape, dif.padded_shape)
        src = dif._condition_source(shape, device, init, control, None)
        with torch.no_grad():
            for _ in range(2):
                x, _ = dif.p_sample(shape, x, 500)
                x = K.apply_cond(x, src, desc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                x, _ = dif.p_sample(shape, x, 500 - i)
                x = K.apply_cond(x, src, desc)
            torch.cuda.synchronize()
            eager = steps / (time.perf_counter() - t0)
            sg = K._step_graph(dif, shape, desc, False, False, device)
            sg.src.copy_(src)
            sg.x.copy_(x)
            for _ in range(2):
                sg.graph.replay()
            torch.cuda.synchronize()
            if barrier is not None:
                barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                sg.t.fill_(500 - i)
                sg.noise.normal_()
                sg.graph.replay()
            torch.cuda.synchronize()
            graph = steps / (time.perf_counter() - t0)
        out[f'batch{b}'] = {'eager_steps_per_sec': round(eager, 2), 'graph_steps_per_sec': round(graph, 2)}
        if b == batch and roofline:
            out['roofline'] = sampling_roofline(dif, shape, x, src, desc, graph)
        K._graph_cache.pop(dif, None)
    return out


def sampling_wide_ab(dif, device, batch, steps):
    """Same-process A/B of the graph-replayed sampling step: the temporal / linear attention blocks of the 128- / 256-channel levels on
    csrc/attn_fused_wide.hip / linattn_fused_wide.hip (default) against the layer-by-layer launches of round 4 (test knobs off), two alternating
    repetitions of `steps` replays each."""
    from wdno_amd import ops, diffusion_core as K
    shape = (batch, 24, 42, 40, 40)
    x0 = torch.randn(shape, device=device)
    init = torch.randn(batch, 24, 40, 40, device=device)
    control = torch.randn(batch, 24, 16, 40, 40, device=device)
    desc = dif._desc(shape, dif.padded_shape)
    src = dif._condition_source(shape, device, init, control, None)
    res = {'fused': [], 'layer_by_layer': []}
    try:
        with torch.no_grad():
            for _ in range(2):
                for name, wide in (('fused', True), ('layer_by_layer', False)):
                    ops.FUSED_TATTN_WIDE = ops.FUSED_LATTN_WIDE = wide
                    K._graph_cache.pop(dif, None)
                    sg = K._step_graph(dif, shape, desc, False, False, device)
                    sg.src.copy_(src)
                    sg.x.copy_(x0)
                    for _ in range(3):
                        sg.graph.replay()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(steps):
                        sg.t.fill_(500 - i)
                        sg.noise.normal_()
                        sg.graph.replay()
                    torch.cuda.synchronize()
                    res[name].append(round(steps / (time.perf_counter() - t0), 2))
    finally:
        ops.FUSED_TATTN_WIDE = ops.FUSED_LATTN_WIDE = True
        K._graph_cache.pop(dif, None)
    return {'graph_steps_per_sec': res, 'note': 'same process, alternating; ops.FUSED_TATTN_WIDE / FUSED_LATTN_WIDE on / off'}


def sampling_roofline(dif, shape, x, src, desc, steps_per_sec):
    """The sampling half of the metric against the rooflines (SURVEY 8d counting rule: one p_sample step = one U-Net forward, 326.3 GFLOP and
    2.02 GB of un-fused leaf-op traffic per sample, + 5 streams of the state): the kernel with the largest total time in one eager step
    (HIP events around every profiled launch, on the launch stream) and the whole graph-replayed step."""
    from wdno_amd import ops, diffusion_core as K
    with torch.no_grad():
        ops.PROFILE = {}
        y, _ = dif.p_sample(shape, x, 321)
        K.apply_cond(y, src, desc)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
    agg = {k: (sum(e0.elapsed_time(e1) for e0, e1, _ in v), sum(f for _, _, f in v), len(v)) for k, v in prof.items()}
    dom = max(agg, key=lambda k: agg[k][0])
    ms, fl, n = agg[dom]
    b = shape[0]
    gf, gb = 326.3, 2.02 + 5 * 4 * 24 * 42 * 40 * 40 / 1e9
    tfl = b * gf / 1e3 * steps_per_sec
    gbs = b * gb * steps_per_sec
    split = 'h3' in dom or 'tattn' in dom
    peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    return {'bound': 'mfma', 'kernel': dom, 'achieved': round(fl / (ms * 1e-3) / 1e12, 2), 'peak': peak, 'unit': 'TFLOP/s',
            'frac': round(fl / (ms * 1e-3) / 1e12 / peak, 4), 'frac_of_fp32_equivalent_ceiling': round(fl / (ms * 1e-3) / 1e12 / (peak / 3), 4) if split else None,
            'launches_per_step': n, 'avg_launch_ms': round(ms / n, 4), 'gflop_per_launch': round(fl / n / 1e9, 3),
            'step': {'tflops_algorithmic': round(tfl, 1), 'frac_mfma': round(tfl / PEAK_F16_MFMA_TFLOPS, 4),
                     'frac_mfma_of_fp32_equivalent_ceiling': round(tfl / (PEAK_F16_MFMA_TFLOPS / 3), 4),
                     'hbm_GBps_unfused': round(gbs, 1), 'frac_hbm': round(gbs / (PEAK_HBM_TBS * 1e3), 4), 'gflop_per_sample': gf,
                     'GB_per_sample_unfused': round(gb, 3)},
            'kernel_ms_per_step': {k: round(v[0], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]}}


def sr_leg(device, batch=2, steps=10, barrier=None):
    """BASELINE.json configs[4] on one GPU (its batch shards over 8 GPUs with no communication): super-resolution DDIM sampling with
    the step replayed from one HIP graph, then the IDWT reconstruction -- the cascade step of smoke/inference_2d.py:155-232 on a
    tensor doubled in time AND space ([B, 48, 82, 80, 80]: the reference's 5-field channel layout, 82 = 2 * 40 + 2, rather than the
    66 channels SURVEY 8d derives for 4 synthetic fields, because the conditioning predicates follow the reference's fixed channel
    positions) -> coefficients [B*5, 8, 34, 66, 66] -> fields [B, 5, 64, 128, 128]."""
    _trees()
    import ptwt
    import pywt
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion
    from wave_trans_2d import tensor_to_coef
    torch.manual_seed(0)
    shapes = [[18, 34, 34], [34, 66, 66]]
    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=82)
    dif = GaussianDiffusion(net, torch.ones(1, 1, 82, 1, 1), True, True, True, True, 'bior1.3', 'zero', shapes, [[32, 64, 64], [64, 128, 128]],
                            image_size=80, frames=48, timesteps=1000, sampling_timesteps=steps, loss_type='l2', ddim_sampling_eta=1.0).to(device)
    low = torch.randn(batch, 48, 40, 80, 80, device=device) * 0.3
    init = torch.randn(batch, 48, 80, 80, device=device)
    control = torch.randn(batch, 48, 16, 80, 80, device=device)
    out = {}
    with torch.no_grad():
        for tag, ug in (('eager', False), ('graph', True)):
            dif.use_graph = ug
            res = dif.sample(batch_size=batch, N_upsample=1, init=init, control=control, low=low)       # warm-up (and capture)
            torch.cuda.synchronize()
            if barrier is not None:         # N > 1: every rank samples its shard of the global batch; no collective (inference_2d.py:155-232)
                barrier()
            t0 = time.perf_counter()
            res = dif.sample(batch_size=batch, N_upsample=1, init=init, control=control, low=low)
            torch.cuda.synchronize()
            out[f'{tag}_ddim_steps_per_sec'] = round(steps / (time.perf_counter() - t0), 2)

        def reconstruct():
            coef = tensor_to_coef(res[:, :, :40].permute(0, 2, 1, 3, 4), shapes[1], upsample_type='space')
            rec = ptwt.waverec3([coef[0].contiguous(), {k: v.contiguous() for k, v in coef[1].items()}], pywt.Wavelet('bior1.3'))
            return rec[:, :64, :128, :128].reshape(-1, 5, 64, 128, 128)
        fields = reconstruct()
        out['idwt_reconstruction_ms'] = round(_ev_time(reconstruct, 10), 3)
        out['fields'] = list(fields.shape)
    out.update(batch=batch, tensor=[batch, 48, 82, 80, 80], ddim_steps=steps)
    return out


def burgers_leg(device, batch, steps, lowp=None, grid=(64, 64)):
    """Burgers base model: train step and p_sample step (BASELINE.json configs[0] shape at batch 16, configs[1] at batch 256). grid
    (80, 64): the north-star's synthetic shape; the training step then starts from FIELDS [B, 2, 160, 128] resident in HBM (HIP DWT +
    packing inside the timed step), and the step on a pre-packed tensor is reported next to it."""
    from wdno_amd import ops
    from wdno_amd.trainer import TrainStep, cosine_annealing_lr
    prev = ops.CONV_MATH
    if lowp:
        ops.CONV_MATH = lowp
    try:
        dif = build_burgers(device, grid)
        ts = TrainStep(dif, lr=1e-4, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=lambda b, s: cosine_annealing_lr(b, s, 10000))
        gh, gw = grid
        x = (torch.randn(batch, 9, gh, gw) * 0.5).to(device)
        for _ in range(3):
            ts.step(x)

        def rate(fn, n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n, r
        # launch by launch (~930 launches per step at batch 16: the host enqueue time is about the kernel time, so this number follows the
        # box's CPU), then the same step replayed from ONE captured HIP graph like the main line (trainer.CapturedStep; bit-identical,
        # tests/test_gpu_graph.py) -- the Trainers' default and the number that measures the GPU
        dt_eager, (loss, _) = rate(lambda: ts.step(x), max(3, steps // 2))
        launch = 'hip_graph_replay'
        try:
            ts.capture(x, warmup=1)
            ts.step(x)
        except Exception as e:
            launch, ts._cap = 'launch by launch (capture failed: ' + repr(e)[:120] + ')', None
            torch.cuda.synchronize()
        dt, (loss, _) = rate(lambda: ts.step(x), steps)
        pipe = None
        if tuple(grid) != (64, 64):
            fields = torch.randn(batch, 2, 2 * gh, 2 * gw, device=device)
            resc = torch.full((1, 9, 1, 1), 4.0, device=device)
            for _ in range(2):
                ts.step(burgers_fields_to_state(fields, resc))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss, _ = ts.step(burgers_fields_to_state(fields, resc))
            torch.cuda.synchronize()
            pipe = (time.perf_counter() - t0) / steps
        ts._cap = None
        from wdno_amd import diffusion_core as K
        with torch.no_grad():
            xs = torch.randn(batch, 9, gh, gw, device=device)
            for t in (500, 499):
                xs = dif.p_sample(xs, t)[0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                xs = dif.p_sample(xs, 400 - i)[0]
            torch.cuda.synchronize()
            ds_eager = (time.perf_counter() - t0) / steps
            # the sampling step as replays of one captured graph (conditions -> U-Net -> posterior update), as diffusion_1d's loops run it
            shape = (batch, 9, gh, gw)
            cw = dif.padded_shape[-1]
            src, desc = dif._sampling_setup(shape, dict(u_init=torch.randn(batch, gh // 2, cw, device=device), f=torch.randn(batch, 4, gh, gw, device=device)))
            sg = K._step_graph(dif, shape, desc, False, True, device)
            sg.src.copy_(src)
            sg.x.copy_(xs)
            for _ in range(2):
                sg.graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                sg.t.fill_(400 - i)
                sg.noise.normal_()
                sg.graph.replay()
            torch.cuda.synchronize()
            ds = (time.perf_counter() - t0) / steps
            K._graph_cache.pop(dif, None)
        out = {'batch': batch, 'tensor': [batch, 9, gh, gw], 'conv_math': ops.CONV_MATH, 'step_launch': launch,
               'train_ms_per_step': round(dt * 1e3, 2), 'train_steps_per_sec': round(1 / dt, 2), 'train_samples_per_sec': round(batch / dt, 1),
               'train_ms_per_step_eager': round(dt_eager * 1e3, 2),
               'p_sample_ms_per_step': round(ds * 1e3, 2), 'p_sample_ms_per_step_eager': round(ds_eager * 1e3, 2), 'final_loss': float(loss)}
        if pipe is not None:
            out['fields_to_step'] = {'fields': [batch, 2, 2 * gh, 2 * gw], 'ms_per_step': round(pipe * 1e3, 2), 'steps_per_sec': round(1 / pipe, 2),
                                     'note': 'HIP DWT + packing + condition channel + train step, fields resident in HBM'}
        return out
    finally:
        ops.CONV_MATH = prev


SMOKE_BF16_NOTE = ('convolutions (3x3x3, 7x7x7 stem, 1x1, strided, transposed; forward, data and weight gradients) on ONE bf16 plane per operand '
                   '(v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 master weights / outputs / norms / softmax); the seven fused 64-channel attention blocks '
                   'keep their fp32-equivalent 3 x fp16-split projections and exact-fp32 score products; the attention blocks of the 128- / 256-channel levels '
                   'project on bf16 planes and attend in fp32')


def smoke_bf16_leg(device, batch, steps):
    """the smoke training step on the single-product kernels (accelerate-style mixed precision, the semantics of
    burgers/ddpm_burgers/train_diffusion.py:61-62,71-74 applied to the smoke model) as a LABELLED side leg -- where the north-star's
    ">= 40 % of the HBM roofline" lands once the matrix work is one product instead of three. Never the headline: it cannot meet 1e-5."""
    from wdno_amd import ops
    from wdno_amd.trainer import TrainStep, multistep_lr
    prev = ops.CONV_MATH
    ops.CONV_MATH = 'bf16'
    try:
        dif = build_model(device, batch)
        ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
        x = (torch.randn(batch, 24, 42, 40, 40, generator=torch.Generator().manual_seed(77)) * 0.5).to(device)
        for _ in range(3):
            ts.step(x)
        launch = 'hip_graph_replay'
        try:
            ts.capture(x, warmup=1)
            ts.step(x)
        except Exception as e:
            launch, ts._cap = 'launch by launch (capture failed: ' + repr(e)[:120] + ')', None
            torch.cuda.synchronize()
            ts.step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, _ = ts.step(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ts._cap = None
        ops.PROFILE = {}
        ts.step(x)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        agg = {k: (sum(e0.elapsed_time(e1) for e0, e1, _ in v), sum(f for _, _, f in v), len(v)) for k, v in prof.items()}
        dom = max(agg, key=lambda k: agg[k][0]) if agg else None
        gf, gb, par_mb = 908.2, 6.06, 95.3
        tfl = batch * gf / 1e3 / dt
        gbs = (batch * gb + 8 * par_mb / 1e3) / dt
        out = {'batch': batch, 'tensor': [batch, 24, 42, 40, 40], 'conv_math': 'bf16', 'ms_per_step': round(dt * 1e3, 3), 'steps_per_sec': round(1 / dt, 2),
               'step_launch': launch, 'final_loss': float(loss), 'tflops_algorithmic': round(tfl, 1), 'frac_mfma': round(tfl / PEAK_F16_MFMA_TFLOPS, 4),
               'mfma_peak_tflops': PEAK_F16_MFMA_TFLOPS, 'hbm_GBps_unfused': round(gbs, 1), 'frac_hbm': round(gbs / (PEAK_HBM_TBS * 1e3), 4),
               'north_star_target_frac_hbm': 0.40, 'arithmetic': SMOKE_BF16_NOTE}
        if dom is not None:
            ms, fl, n = agg[dom]
            out['dominant_kernel'] = {'kernel': dom, 'launches_per_step': n, 'avg_launch_ms': round(ms / n, 4), 'achieved_tflops': round(fl / (ms * 1e-3) / 1e12, 1),
                                      'frac_of_bf16_peak': round(fl / (ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4)}
            out['kernel_ms_per_step'] = {k: round(v[0], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:10]}
        return out
    finally:
        ops.CONV_MATH = prev


def train_graph_leg(ts, x, steps, cap=None):
    """The same training step with loss -> backward -> gradient gather replayed from ONE captured HIP graph (TrainStep.capture; draws,
    clip + Adam and EMA stay outside): host time to enqueue a step and steps/s, eager vs graph. Same kernels either way -- at batch 8
    the GPU is the bound, so the rate barely moves; what changes is the host side (18 ms of Python per step -> a graph launch)."""
    def measure():
        enq = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ts.step(x)
            enq.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step(x)
        torch.cuda.synchronize()
        return round(_median(enq), 2), round((time.perf_counter() - t0) / steps * 1e3, 3)
    ts._cap = None
    e_enq, e_ms = measure()
    if cap is not None:
        ts._cap = cap
    else:
        ts.capture(x, warmup=1)
    for _ in range(2):
        ts.step(x)
    g_enq, g_ms = measure()
    ts._cap = None
    return {'eager': {'host_enqueue_ms_per_step': e_enq, 'ms_per_step': e_ms}, 'graph': {'host_enqueue_ms_per_step': g_enq, 'ms_per_step': g_ms}}


def smoke_fields_to_state(fields, rs, curve=None):
    """fields [B, nf, 32, 64, 64] and the smoke-out curve [B, 32], resident in HBM -> the training tensor [B, 24, 8 nf + 2, 40, 40]
    (wave_trans_2d.py:150-170 + data_2d.py:156-221) in TWO launches (ddpm.data_2d.pack_smoke_fields): the fused 3-D DWT of the fields (bior1.3 /
    zero) and a packing launch that pads, divides by the RESCALER and transforms the two condition channels (2-D DWT of rho(t = 0), 1-D DWT of
    the curve) on the fly. curve None: a synthetic curve per call (one more launch)."""
    from ddpm.data_2d import pack_smoke_fields
    if curve is None:
        curve = torch.rand(fields.shape[0], fields.shape[2], device=fields.device)
    return pack_smoke_fields(fields, curve, rs)


def smoke34_leg(device, batch, steps, sample_steps):
    """BASELINE's own synthetic tensors through the U-Net (VERDICT r5 missing #4): fields [B, 4, 32, 64, 64] -> HIP DWT + packing ->
    [B, 24, 34, 40, 40] -> Unet3D_with_Conv3D(dim=64, (1,2,4), channels=34): the training step (graph replay, fields resident in HBM, transform
    inside the timed step) and the DDPM sampling step (graph replay). 3.0 % less matrix work than the reference-native 42 channels (881.1 vs 908.2
    GFLOP per sample): the 34 -> 64 stem and the 64 -> 34 head are the only layers that differ."""
    from wdno_amd import diffusion_core as K
    from wdno_amd.trainer import TrainStep, multistep_lr
    from ddpm.data_2d import _RESCALERS
    dif = build_model(device, batch, channels=34)
    ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
    r42 = torch.tensor(_RESCALERS['bior1.3'], dtype=torch.float32, device=device).reshape(1, 42, 1, 1)
    r34 = torch.cat((r42[:, :32], r42[:, -2:]), dim=1)
    fields = torch.randn(batch, 4, 32, 64, 64, device=device)
    curve = torch.rand(batch, 32, device=device)
    x = smoke_fields_to_state(fields, r34, curve)
    assert tuple(x.shape) == (batch, 24, 34, 40, 40)
    for _ in range(3):
        ts.step(x)
    launch = 'hip_graph_replay'
    try:
        ts.capture(x, warmup=1)
        ts.step(x)
    except Exception as e:
        launch, ts._cap = 'launch by launch (capture failed: ' + repr(e)[:120] + ')', None
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = ts.step(smoke_fields_to_state(fields, r34, curve))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ts._cap = None
    shape = (batch, 24, 34, 40, 40)
    with torch.no_grad():
        desc = dif._desc(shape, dif.padded_shape)
        src = dif._condition_source(shape, device, torch.randn(batch, 24, 40, 40, device=device), torch.randn(batch, 24, 10, 40, 40, device=device), None)
        sg = K._step_graph(dif, shape, desc, False, False, device)
        sg.src.copy_(src)
        sg.x.copy_(K.apply_cond(torch.randn(shape, device=device), src, desc))
        for _ in range(2):
            sg.graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(sample_steps):
            sg.t.fill_(500 - i)
            sg.noise.normal_()
            sg.graph.replay()
        torch.cuda.synchronize()
        sps = sample_steps / (time.perf_counter() - t0)
    K._graph_cache.pop(dif, None)
    gf, gb, par_mb = 881.1, 6.05, 95.2
    return {'fields': [batch, 4, 32, 64, 64], 'tensor': list(shape), 'channels': 34, 'step_launch': launch, 'train_ms_per_step': round(dt * 1e3, 3),
            'train_steps_per_sec': round(1 / dt, 2), 'final_loss': float(loss), 'ddpm_sample_steps_per_sec': round(sps, 2),
            'tflops_algorithmic': round(batch * gf / 1e3 / dt, 1), 'frac_hbm': round((batch * gb + 8 * par_mb / 1e3) / dt / (PEAK_HBM_TBS * 1e3), 4),
            'note': 'train step = DWT + packing of the fields + q_sample ... Adam + EMA; conditioning predicates at the reference\'s channel positions (C - 2, C - 1, 24:40 clipped to C)'}


def loader_leg(ts, x, device, batch, steps, n_sims=64):
    """The real-data input path (VERDICT r5 missing #5): n_sims synthetic simulations written to local disk in the offline transform's own format
    (smoke/wave_trans_2d.py:172-185: a torch.save'd dict per simulation) -> ddpm.data_2d.Smoke_wave -> the training step, three ways:
      resident_tensor : the step on a pre-packed tensor resident in HBM (the main line's input);
      resident_loader : wdno_amd.loader.ResidentSmokeLoader -- epoch 1 reads every file once (worker processes, pinned staging) into HBM-resident
                        stores of the RAW arrays, every batch is packed by ONE launch of csrc/pack.hip; epoch 2 onward touches no file and no
                        host packing (timed separately);
      dataloader      : the reference's pipeline -- torch DataLoader, 16 workers, torch.load + host packing per sample, pinned copy per batch."""
    import shutil
    import tempfile
    _trees()
    from ddpm.data_2d import Smoke_wave
    from wdno_amd.loader import ResidentSmokeLoader
    from wdno_amd.trainer import cycle_loader
    root = tempfile.mkdtemp(prefix='wdno_bench_')
    out = {'simulations': n_sims, 'batch': batch}
    try:
        d = os.path.join(root, 'train', 'bior1.3_zero', 'time_downsample')
        os.makedirs(d)
        g = torch.Generator().manual_seed(0)
        for i in range(n_sims):
            torch.save({'coef': [torch.randn(5, 8, 18, 34, 34, generator=g) * 3], 'init_coef': [torch.randn(5, 4, 34, 34, generator=g)],
                        'smokeout': [torch.rand(2, 18, generator=g)], 'shape': [(18, 34, 34)], 'ori_shape': (32, 64, 64)}, os.path.join(d, f'{i:06d}'))
        ds = Smoke_wave(root, 'bior1.3', 'zero')
        ds.n_simu = n_sims

        def rate(next_batch, n):
            for _ in range(2):
                ts.step(next_batch())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                ts.step(next_batch())
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        dt0 = rate(lambda: x, steps)
        ld = ResidentSmokeLoader(ds, batch, device, num_workers=8)
        it = cycle_loader(ld)
        per_epoch = len(ld)
        t0 = time.perf_counter()
        for _ in range(per_epoch):                        # epoch 1: files -> HBM (and the steps that consume them)
            ts.step(next(it)[0])
        torch.cuda.synchronize()
        dt_first = (time.perf_counter() - t0) / per_epoch
        dt1 = rate(lambda: next(it)[0], steps)            # resident: gather + pack on the GPU
        pack_ms = _ev_time(lambda: next(it)[0], 20)
        from torch.utils.data import DataLoader
        dl = cycle_loader(DataLoader(ds, batch_size=batch, shuffle=True, pin_memory=True, num_workers=16))
        dt2 = rate(lambda: next(dl)[0].to(device, non_blocking=True), steps)
        del dl
        out.update(resident_tensor_ms_per_step=round(dt0 * 1e3, 3), resident_loader_ms_per_step=round(dt1 * 1e3, 3), resident_loader_steps_per_sec=round(1 / dt1, 2),
                   resident_loader_vs_resident_tensor=round(dt0 / dt1, 4), first_epoch_ms_per_step=round(dt_first * 1e3, 3), gather_pack_ms_per_batch=round(pack_ms, 4),
                   resident_MB=round(ld.resident_bytes() / 1e6, 1), dataloader_16_workers_ms_per_step=round(dt2 * 1e3, 3),
                   dataloader_vs_resident_tensor=round(dt0 / dt2, 4))
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return out


def smoke_pipeline_leg(ts, device, batch, steps):
    """The smoke pipeline end to end on the GPU: fields [B, 5, 32, 64, 64] (rho, vx, vy, cx, cy) resident in HBM -> ONE fused 3-D HIP DWT
    launch (bior1.3 / zero) -> [B*5, 8, 18, 34, 34] -> pack_smoke_batch (+ init-density and smoke-out condition channels, / RESCALER)
    -> [B, 24, 42, 40, 40] -> the training step of the main line (data_2d.py:156-221 + diffusion_2d.py:1257-1307). The 5-field layout
    is the reference's (its conditioning predicates sit at fixed channel positions); the north-star's 4-field synthetic tensor
    [B, 4, 32, 64, 64] -> [B, 24, 34, 40, 40] differs in the channel count of the first and last convolution only and is timed through
    the transform + packing as well."""
    from ddpm.data_2d import _RESCALERS
    out = {}
    resc = torch.tensor(_RESCALERS['bior1.3'], dtype=torch.float32, device=device).reshape(1, 42, 1, 1)
    to_state = smoke_fields_to_state
    fields = torch.randn(batch, 5, 32, 64, 64, device=device)
    curve = torch.rand(batch, 32, device=device)
    st = to_state(fields, resc, curve)
    assert tuple(st.shape) == (batch, 24, 42, 40, 40)

    def both(fn):
        """per Python call (host enqueue included) and per replay of 20 calls captured in one graph (the two launches alone)"""
        ms_call = _ev_time(fn, 20)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        return round(ms_call, 3), round(_ev_time(g.replay, 10) / 20, 4)
    out['transform_and_pack_ms'], out['transform_and_pack_kernels_ms'] = both(lambda: to_state(fields, resc, curve))
    out['launches'] = 'dwt_analysis_fused<packed store> (3-D, all fields, sub-bands -> channels of the state, / RESCALER) + pack_smoke_state_kernel<fill> (padding + condition channels)'
    for _ in range(2):
        ts.step(to_state(fields, resc, curve))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step(to_state(fields, resc, curve))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out.update(fields=[batch, 5, 32, 64, 64], state=[batch, 24, 42, 40, 40], fields_to_step_ms=round(dt * 1e3, 2), steps_per_sec=round(1 / dt, 2))
    f4 = torch.randn(batch, 4, 32, 64, 64, device=device)
    r4 = torch.cat((resc[:, :32], resc[:, -2:]), dim=1)
    s4 = to_state(f4, r4, curve)
    t4 = both(lambda: to_state(f4, r4, curve))
    out['four_field_synthetic'] = {'fields': [batch, 4, 32, 64, 64], 'state': list(s4.shape), 'transform_and_pack_ms': t4[0], 'transform_and_pack_kernels_ms': t4[1]}
    return out


def conv_roofline(ts_step, ops):
    """Per-launch HIP-event timing of the convolution kernels over one extra step -> roofline of the dominant kernel."""
    ops.PROFILE = {}
    ts_step()
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for key, evs in prof.items():
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
        fl = sum(f for _, _, f in evs)
        agg[key] = (ms, fl, len(evs))
    if not agg:
        return None
    dom = max(agg, key=lambda k: agg[k][0])
    ms, fl, n = agg[dom]
    achieved = fl / (ms * 1e-3) / 1e12
    split = 'h3' in dom and ops.CONV_MATH == 'f16x3'
    peak = PEAK_F32_MFMA_TFLOPS if 'h3' not in dom else PEAK_F16_MFMA_TFLOPS
    traffic = path = stale = None  # HBM bytes per launch from the committed PMC passes (profiles/*_pmc_traffic.json); stale: the kernel sources changed since
    try:
        fam, dims = dom.split('<')
        dims = dims.rstrip('>').split(',')
        sym = fam + 'I' + ''.join(f'Li{d}E' for d in dims) if all(d.isdigit() for d in dims) else None
        path = next((q for q in (PROFILE_JSON, os.path.join(ROOT, 'profiles', 'r04_pmc_traffic.json'), os.path.join(ROOT, 'profiles', 'r03_pmc_traffic.json'))
                     if os.path.exists(q)), PROFILE_JSON)
        with open(path) as f:
            prof = json.load(f)
            recs = prof['kernels']
            import glob, hashlib
            hh = hashlib.sha256()
            for q in sorted(glob.glob(os.path.join(ROOT, 'wdno_amd', 'csrc', '*.h*'))):
                hh.update(open(q, 'rb').read())
            stale = None if 'csrc_sha16' not in prof else prof['csrc_sha16'] != hh.hexdigest()[:16]
            hit = [r for k, r in recs.items() if sym and sym in k] or [r for k, r in recs.items() if (fam + 'I') in k]
            if hit:                               # entries are ordered by total time: the first match is the main instantiation
                traffic = round(hit[0]['hbm_bytes_per_launch'])
    except Exception:
        traffic = None
    return {'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
            'frac': round(achieved / peak, 4), 'traffic': traffic, 'traffic_stale': stale if traffic is not None else None, 'traffic_source': (os.path.relpath(path, ROOT) + ' (committed rocprofv3 --pmc passes of this kernel; a lookup, not a counter read in this run)') if traffic is not None else None,
            'launches_per_step': n,
            'frac_of_fp32_equivalent_ceiling': round(achieved / (peak / 3), 4) if split else None,
            'note': ('fp32-equivalent 3 x fp16-split MFMA: 3 matrix flop per algorithmic flop, so frac <= 0.333; the exact-fp32 MFMA peak is 157.3 TFLOP/s'
                     if split else ('single-product 16-bit MFMA' if 'h3' in dom else 'exact-fp32 MFMA')),
            'avg_launch_ms': round(ms / n, 4), 'gflop_per_launch': round(fl / n / 1e9, 3),
            'conv_ms_per_step': {k: round(v[0], 3) for k, v in agg.items()}}


def _rccl_version():
    try:
        return list(torch.cuda.nccl.version())
    except Exception as e:
        return repr(e)


def self_launch(n):
    """Re-execute this script as n ranks under torch.distributed.run on a free local port; rank 0's JSON line passes through."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=150)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='samples per GPU per step (default 8 smoke, 16 burgers, 256 burgers-bf16)')
    ap.add_argument('--workload', default='smoke', choices=['smoke', 'smoke-bf16', 'burgers', 'burgers-bf16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the sampling / DWT / Burgers side legs')
    ap.add_argument('--eager', action='store_true', help='timed steps launch by launch instead of replaying the captured HIP graph')
    ap.add_argument('--sample-steps', type=int, default=20)
    ap.add_argument('--burgers-grid', default='64x64', help="tensor size of the Burgers workloads: 64x64 (reference-native) or 80x64 (north-star synthetic: fields [B,2,160,128] through the HIP DWT)")
    ap.add_argument('--max-seconds', type=float, default=float(os.environ.get('WDNO_BENCH_MAX_SECONDS', '1500')),
                    help='wall-clock guard for the whole run: a SIDE leg is skipped (and named in `skipped_legs`) when its budgeted cost no longer fits; the main line is never dropped')
    args = ap.parse_args()
    t_run0 = time.perf_counter()

    if args.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the reference's
        # `accelerate launch --num_processes N`, scripts/smoke/train_base_sim.sh:3-9) and hand their exit code back.
        sys.exit(self_launch(args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback on the product path)'
    from wdno_amd.trainer import init_distributed
    rank, world, local = init_distributed()                # torchrun: set_device(LOCAL_RANK) + RCCL ("nccl") process group
    import torch.distributed as dist
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the process group has {world} rank(s) '
                         f'(WORLD_SIZE={os.environ.get("WORLD_SIZE")}, RANK={os.environ.get("RANK")}): refusing to report a {world}-rank number as {args.gpus}')
    if world > 1 and os.environ.get('WDNO_DIST_SHARE_GPU') != '1' and torch.cuda.device_count() < world:
        raise SystemExit(f'bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) visible')
    device = torch.device('cuda', local)

    from wdno_amd import _lib, ops
    from wdno_amd.trainer import TrainStep, cosine_annealing_lr, multistep_lr
    _lib.load()
    smoke = args.workload in ('smoke', 'smoke-bf16')
    if args.workload in ('burgers-bf16', 'smoke-bf16'):
        ops.CONV_MATH = 'bf16'
    batch = args.batch or (8 if smoke else (256 if args.workload == 'burgers-bf16' else 16))
    if smoke:
        dif = build_model(device, batch)
        ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
        shape = (batch, 24, 42, 40, 40)
        grad_mb = 95.3
    else:
        bgrid = tuple(int(v) for v in args.burgers_grid.split('x'))
        dif = build_burgers(device, bgrid)
        ts = TrainStep(dif, lr=1e-4, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=lambda b, s: cosine_annealing_lr(b, s, 10000), use_ema=True)
        shape = (batch, 9, *bgrid)
        grad_mb = 563.0
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    x = (torch.randn(shape, generator=g) * 0.5).to(device)      # resident in HBM before timing

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ts.time_comm = world > 1
    losses = []
    for _ in range(args.warmup):
        loss, _ = ts.step(x)
    # The timed steps replay ONE captured HIP graph (loss -> backward -> gradient gather; the draws, the RCCL exchange, clip + Adam and EMA are
    # launched around it, wdno_amd.trainer.CapturedStep): the drop-in Trainers' default. Same kernels in the same order as the eager step
    # (bit-identical, tests/test_gpu_graph.py). --eager: launch by launch.
    graphed = not args.eager                 # (with the overlapped bucket exchange the all-reduces are captured inside the graph, trainer.CapturedStep)
    capture_error = None
    if graphed:
        try:
            ts.capture(x, warmup=1)
            ts.step(x)
        except Exception as e:          # a failed capture must not cost the measurement: the same step launch by launch, and the line says so
            capture_error, graphed, ts._cap = repr(e)[:200], False, None
            torch.cuda.synchronize()
            ts.step(x)
    barrier()
    ts.comm_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, gn = ts.step(x)
        losses.append(loss)
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = {'ms_per_step': [round(local_elapsed / args.steps * 1e3, 3)], 'exposed_allreduce_ms': [0.0]}      # same schema at N = 1 (no exchange)
    if world > 1:
        comm_ms = sum(a.elapsed_time(b) for a, b in ts.comm_events) / max(1, len(ts.comm_events))
        tt = torch.tensor([elapsed, local_elapsed / args.steps * 1e3, comm_ms], device=device, dtype=torch.float64)
        allv = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allv, tt)
        elapsed = max(float(v[0]) for v in allv)
        per_rank = {'ms_per_step': [round(float(v[1]), 3) for v in allv], 'exposed_allreduce_ms': [round(float(v[2]), 3) for v in allv]}
    final_loss = float(losses[-1]) if losses else float('nan')

    extras = {}
    roofline = cpu = None
    cap, ts._cap = getattr(ts, '_cap', None), None          # the profiled extra step runs launch by launch (HIP events around every convolution)
    if rank != 0 and world > 1:
        ts.step(x)              # rank 0's profiled extra step below contains the gradient all-reduce: every rank has to take part in it
    if rank == 0:
        roofline = conv_roofline(lambda: ts.step(x), ops)
    run_multi_side = world > 1 and smoke and not args.no_extras
    if run_multi_side:
        # every rank has to take the same decision (the legs end in an all_gather): rank 0's clock decides, broadcast to all
        flag = torch.tensor([1.0 if (time.perf_counter() - t_run0) + 150 <= args.max_seconds else 0.0], device=device)
        dist.broadcast(flag, 0)
        run_multi_side = bool(flag.item() == 1.0)
        if not run_multi_side and rank == 0:
            extras['skipped_legs'] = [{'leg': 'sampling + sr_sampling at N > 1', 'budgeted_s': 150, 'elapsed_s': round(time.perf_counter() - t_run0, 1)}]
    if run_multi_side:
        # The sampling half of the metric at N > 1 (BASELINE configs[3] / [4]): every rank samples its own batch -- the path shards with no
        # collective (smoke/inference_2d.py:123-152, 155-232) -- from loops that start together (barrier); the global rate is what the slowest
        # rank allows: world x min over ranks = samples of all ranks / max-over-ranks time.
        try:
            sl = sampling_leg(dif, device, batch, args.sample_steps, barrier=barrier, roofline=False)
            del ts, dif
            torch.cuda.empty_cache()
            sr = sr_leg(device, batch=2, steps=10, barrier=barrier)
            mine = [sl[f'batch{batch}']['eager_steps_per_sec'], sl[f'batch{batch}']['graph_steps_per_sec'],
                    sr['eager_ddim_steps_per_sec'], sr['graph_ddim_steps_per_sec'], sr['idwt_reconstruction_ms'], 0.0]
        except Exception as e:          # every rank still takes part in the gather below
            import traceback
            mine, sr = [0.0] * 5 + [1.0], {'error': repr(e) + ' | ' + traceback.format_exc()[-400:]}
        tt = torch.tensor(mine, device=device, dtype=torch.float64)
        allv = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allv, tt)
        if rank == 0:
            col = lambda i: [round(float(v[i]), 2) for v in allv]
            if any(float(v[5]) for v in allv):
                extras['sampling_error'] = sr.get('error', 'a rank other than 0 failed in its sampling legs')
            else:
                extras['sampling'] = {'batch_per_rank': batch, 'tensor_per_rank': [batch, 24, 42, 40, 40], 'ranks': world,
                                      'graph_steps_per_sec_global': round(world * min(col(1)), 2), 'eager_steps_per_sec_global': round(world * min(col(0)), 2),
                                      'per_rank': {'graph_steps_per_sec': col(1), 'eager_steps_per_sec': col(0)},
                                      'note': 'rank-steps/s: every rank advances its own batch of 8 by one DDPM step; no collective on this path'}
                extras['sr_sampling'] = {'batch_per_rank': 2, 'global_batch': 2 * world, 'tensor_per_rank': [2, 48, 82, 80, 80], 'ddim_steps': 10, 'ranks': world,
                                         'graph_ddim_steps_per_sec_global': round(world * min(col(3)), 2), 'eager_ddim_steps_per_sec_global': round(world * min(col(2)), 2),
                                         'samples_per_sec_global': round(2 * world * min(col(3)) / 10, 3),
                                         'per_rank': {'graph_ddim_steps_per_sec': col(3), 'eager_ddim_steps_per_sec': col(2), 'idwt_reconstruction_ms': col(4)},
                                         'fields': sr.get('fields'),
                                         'note': 'BASELINE configs[4]: the global batch shards over the ranks, each replays its captured DDIM step and reconstructs its fields (IDWT); rank-steps/s'}
    skipped = []

    def fits(name, cost_s):
        """--max-seconds guard: may a side leg with this budgeted cost (seconds, generous) still start? The main line above is never subject to it."""
        ok = (time.perf_counter() - t_run0) + cost_s <= args.max_seconds
        if not ok:
            skipped.append({'leg': name, 'budgeted_s': cost_s, 'elapsed_s': round(time.perf_counter() - t_run0, 1)})
        return ok
    if rank == 0:
        if world == 1 and not args.no_extras:
            try:
                if smoke:
                    if fits('sampling', 40):
                        extras['sampling'] = sampling_leg(dif, device, batch, args.sample_steps)
                        if fits('sampling.wide_attention_ab', 30):
                            extras['sampling']['wide_attention_ab'] = sampling_wide_ab(dif, device, batch, args.sample_steps)
                    if fits('fields_pipeline', 20):
                        extras['fields_pipeline'] = smoke_pipeline_leg(ts, device, batch, 10)
                    if fits('train_step_graph', 20):
                        extras['train_step_graph'] = train_graph_leg(ts, x, 20, cap)
                    if args.workload == 'smoke' and fits('loader', 60):
                        ts._cap = cap
                        extras['loader'] = loader_leg(ts, x, device, batch, 20)
                        ts._cap = None
                if fits('dwt', 15):
                    extras['dwt'] = dwt_leg(device)
                if smoke:
                    del ts, dif
                    torch.cuda.empty_cache()
                    if args.workload == 'smoke' and fits('smoke_synthetic_34ch', 40):
                        extras['smoke_synthetic_34ch'] = smoke34_leg(device, batch, 20, args.sample_steps)
                        torch.cuda.empty_cache()
                    if fits('sr_sampling', 60):
                        extras['sr_sampling'] = sr_leg(device)
                        torch.cuda.empty_cache()
                    if args.workload == 'smoke' and ops.LOWP_AVAILABLE and fits('smoke_bf16', 40):
                        extras['smoke_bf16'] = smoke_bf16_leg(device, batch, 20)
                        torch.cuda.empty_cache()
                    bl = {}
                    if fits('burgers.fp32_equivalent_batch16', 40):
                        bl['fp32_equivalent_batch16'] = burgers_leg(device, 16, 20)
                    if ops.LOWP_AVAILABLE and fits('burgers.bf16_batch256', 60):
                        bl['bf16_batch256'] = burgers_leg(device, 256, 5, lowp='bf16')
                    if fits('burgers.north_star_80x64.fp32_equivalent_batch16', 40):
                        bl.setdefault('north_star_80x64', {})['fp32_equivalent_batch16'] = burgers_leg(device, 16, 20, grid=(80, 64))
                    if ops.LOWP_AVAILABLE and fits('burgers.north_star_80x64.bf16_batch256', 70):
                        bl.setdefault('north_star_80x64', {})['bf16_batch256'] = burgers_leg(device, 256, 5, lowp='bf16', grid=(80, 64))
                    if bl:
                        extras['burgers'] = bl
                elif bgrid != (64, 64) and fits('fields_pipeline', 60):
                    extras['fields_pipeline'] = burgers_leg(device, batch, 10, lowp='bf16' if args.workload == 'burgers-bf16' else None, grid=bgrid).get('fields_to_step')
            except Exception as e:      # side legs never invalidate the main line
                import traceback
                extras['error'] = repr(e) + ' | ' + traceback.format_exc()[-600:]
        if world == 1 and not args.no_cpu_baseline:
            if fits('cpu_baseline', 45):
                try:
                    cpu = cpu_baseline()
                except Exception as e:      # the baseline is informative only
                    cpu = {'error': repr(e)}
            else:
                cpu = {'skipped': '--max-seconds'}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        if smoke and args.workload == 'smoke-bf16':
            metric = 'diffusion train steps/sec, 2D smoke U-Net (8 samples per GPU per step), bf16 single-product convolutions'
            wl = (f'smoke base-resolution DDPM train step in MIXED PRECISION (not the headline configuration): Unet3D_with_Conv3D(dim=64,(1,2,4),ch=42) on wavelet tensor '
                  f'[{batch},24,42,40,40] per GPU, fp32 in/out; ' + SMOKE_BF16_NOTE + '; Adam+clip+EMA')
            dtype = 'bf16'
        elif smoke:
            metric = 'diffusion train steps/sec, 2D smoke U-Net (8 samples per GPU per step)'
            wl = (f'smoke base-resolution DDPM train step: Unet3D_with_Conv3D(dim=64,(1,2,4),ch=42) on wavelet tensor [{batch},24,42,40,40] per GPU, '
                  'fp32 in/out, convolutions on the fp32-equivalent 3 x fp16-split MFMA path (small ones exact-fp32 MFMA), Adam+clip+EMA')
            dtype = 'f32'
        else:
            metric = f'diffusion train steps/sec, 1D Burgers U-Net ({batch} samples per GPU per step)'
            math = 'bf16 single-product MFMA with fp32 master weights and accumulators' if args.workload == 'burgers-bf16' else 'fp32-equivalent 3 x fp16-split MFMA'
            wl = f'Burgers base-resolution DDPM train step: Unet2D(dim=128,(1,2,4,8),ch=9,groups=1) on [{batch},9,{bgrid[0]},{bgrid[1]}] per GPU, convolutions: {math}, Adam+clip+EMA'
            dtype = 'bf16' if args.workload == 'burgers-bf16' else 'f32'
        # whole-step roofline (SURVEY 8d counting rule: FlopCounterMode flop of the reference model, fwd + bwd = 3 x fwd; bytes = un-fused
        # leaf-module input + output bytes, + parameters once, + the optimiser's 7 parameter-sized streams)
        if smoke:
            gf, gb, par_mb = 908.2, 6.06, 95.3
        elif bgrid == (64, 64):
            gf, gb, par_mb = 167.9, 1.05, 563.0
        else:
            gf, gb, par_mb = 209.8 * (bgrid[0] * bgrid[1]) / 5120.0, 1.32 * (bgrid[0] * bgrid[1]) / 5120.0, 563.0
        step_s = ms_per_step * 1e-3
        tfl = batch * gf / 1e3 / step_s
        gbs = (batch * gb + 8 * par_mb / 1e3) / step_s
        mfma_peak = PEAK_F16_MFMA_TFLOPS
        step_roofline = {'tflops_algorithmic': round(tfl, 1), 'frac_mfma': round(tfl / mfma_peak, 4), 'mfma_peak_tflops': mfma_peak,
                         'frac_mfma_of_fp32_equivalent_ceiling': round(tfl / (mfma_peak / 3), 4) if dtype == 'f32' else None,
                         'hbm_GBps_unfused': round(gbs, 1), 'frac_hbm': round(gbs / (PEAK_HBM_TBS * 1e3), 4),
                         'bound': 'mfma', 'gflop_per_sample': round(gf, 1), 'GB_per_sample_unfused': round(gb, 3),
                         'note': 'the step is bound by matrix work (about 160 flop per un-fused byte against a ridge of 2500 / 8 = 312 flop/B for ONE '
                                 '16-bit product and 104 flop/B for the 3-product fp32-equivalent split); both fractions are of the whole driver-timed step'}
        # value = RANK-steps/s: under weak scaling every rank advances its own 8 samples per step, so the whole-job aggregate of the metric's unit is
        # world x (global optimiser steps/s); global_steps_per_sec is the rate of optimiser steps on the global batch (8 x world samples each)
        out = {
            'metric': metric, 'value': round(world * args.steps / elapsed, 4), 'unit': 'steps/s',
            'value_definition': f'{batch}-sample rank-steps/s summed over the {world} rank(s) = samples_per_sec / {batch} (weak scaling: the per-GPU batch is fixed)',
            'global_steps_per_sec': round(args.steps / elapsed, 4), 'global_batch': batch * world,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': wl, 'global_batch': batch * world, 'parallelism': f'dp{world}', 'grad_allreduce_MB': grad_mb if world > 1 else 0,
                       'step_launch': ('hip_graph_replay (loss + backward + gradient gather captured' + ('; the overlapped bucket all-reduces are nodes of the graph; draws, clip + Adam, EMA around it)' if getattr(cap, 'exchanged', False) else '; draws, exchange, clip + Adam, EMA around it)')) if graphed else ('launch by launch' + (f' (capture failed: {capture_error})' if capture_error else '')),
                       'process_group': ({'backend': dist.get_backend(), 'ranks': dist.get_world_size(),
                                          'rccl_version': _rccl_version() if dist.get_backend() == 'nccl' else None,
                                          'devices_visible': torch.cuda.device_count()} if world > 1 else None)},
            'samples_per_sec': round(world * args.steps * batch / elapsed, 3),
            'final_loss': final_loss,
            'roofline': roofline, 'step_roofline': step_roofline, 'cpu_baseline': cpu,
        }
        if 'sampling' in extras:
            out['ddpm_sample_steps_per_sec'] = (extras['sampling']['graph_steps_per_sec_global'] if world > 1 else
                                                extras['sampling'][f'batch{batch}']['graph_steps_per_sec'])
            out['sampling_roofline'] = extras['sampling'].pop('roofline', None)
        out['per_rank'] = per_rank
        if skipped:
            extras.setdefault('skipped_legs', []).extend(skipped)
        out['wall_seconds'] = {'total': round(time.perf_counter() - t_run0, 1), 'max_seconds': args.max_seconds}
        out.update(extras)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
