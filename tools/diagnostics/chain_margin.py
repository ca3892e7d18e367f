"""the margin of the split-fp16 path on the full-size 10-step DDIM chain, over seeds and with the hand-over forms
switched one at a time, WITHOUT the +1e-6 slack of the test gate:

    hip_vs_exact / cpu32_vs_exact      per seed, for: default f16x3 | exact-fp32 convolutions | f16x3 without plane hand-over between layers
                                       | f16x3 with the attention blocks layer by layer | f16x3 with per-launch amax sweeps

Writes gpurun_out/chain_margin.json.   python tools/diagnostics/chain_margin.py [seed ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import parity_report as R  # noqa: E402
from wdno_amd import ops  # noqa: E402
from tests.helpers import rel_l2  # noqa: E402
from oracle import diffusion_ref as D, unet_ref as U  # noqa: E402  (checker)

DEV = 'cuda'


def chain(seed, steps=10, batch=1):
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion
    torch.manual_seed(0)
    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    lw = torch.ones(1, 1, 42, 1, 1)
    g = torch.Generator().manual_seed(seed)
    shape = (batch, 24, 42, 40, 40)
    ns = [torch.randn(shape, generator=g) for _ in range(steps + 1)]
    init = torch.randn(batch, 24, 40, 40, generator=g) * 0.3
    control = torch.randn(batch, 24, 16, 40, 40, generator=g) * 0.3

    def cpu(dt):
        sd = R.cast_sd(sd0, dt)
        model = lambda x, tt: U.unet3d_forward(sd, x, tt, dim=64, dim_mults=(1, 2, 4), groups=8)
        with torch.no_grad():
            return D.smoke_ddim_sample(model, R.cast_buf(D.make_buffers('sigmoid', 1000), dt), [n.to(dt) for n in ns], 1000, steps, 1.0,
                                       padded_shape=(18, 34, 34), init=init.to(dt), control=control.to(dt))
    o32, o64 = cpu(torch.float32), cpu(torch.float64)
    ref = rel_l2(o32, o64)
    row = {'seed': seed, 'cpu32_vs_exact': ref}

    def hip(tag, math='f16x3', **knobs):
        prev = {k: getattr(ops, k) for k in knobs}
        for k, v in knobs.items():
            setattr(ops, k, v)
        ops.CONV_MATH = math
        ops.bump_weight_epoch()
        try:
            net_h = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
            net_h.load_state_dict(sd0)
            dif = GaussianDiffusion(net_h, lw, True, True, True, False, 'bior1.3', 'zero', (18, 34, 34), (32, 64, 64), image_size=40, frames=24,
                                    sampling_timesteps=steps, ddim_sampling_eta=1.0).to(DEV)
            dif.use_graph = False
            seq = iter([n.to(DEV) for n in ns])
            dif.sample_noise = lambda shape, device: next(seq)
            o = dif.sample(batch_size=batch, init=init.to(DEV), control=control.to(DEV)).cpu()
        finally:
            for k, v in prev.items():
                setattr(ops, k, v)
            ops.CONV_MATH = 'f16x3'
        h = rel_l2(o, o64)
        row[tag] = {'hip_vs_exact': h, 'ratio_to_cpu32': h / ref, 'hip_vs_cpu32': rel_l2(o, o32)}
    hip('f16x3')
    hip('fp32', math='fp32')
    hip('f16x3_no_plane_handover', PLANES_FWD=False)
    hip('f16x3_attention_layer_by_layer', FUSED_TATTN=False, FUSED_LATTN=False)
    hip('f16x3_amax_sweeps', AMAX_HINTS=False)
    return row


def main():
    seeds = [int(v) for v in sys.argv[1:]] or [7, 8, 9]
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    out = {'rows': []}
    for s in seeds:
        t0 = time.perf_counter()
        row = chain(s)
        row['seconds'] = round(time.perf_counter() - t0, 1)
        out['rows'].append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'chain_margin.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
