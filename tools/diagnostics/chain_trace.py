"""Debugging aid (GPU box): follow one multi-seed smoke DDIM-4 chain (tests/golden/ref_round3.npz) step by step in three evaluations --
hip (product path), cpu32 (oracle fp32) and exact (oracle fp64), each on its OWN state -- and print where hip leaves cpu32.
    python tools/diagnostics/chain_trace.py [seed]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wdno_amd import tree_path  # noqa: E402
for _t in ('third_party', 'smoke', 'burgers'):
    sys.path.insert(0, tree_path(_t))
from oracle import diffusion_ref as D, unet_ref as U  # noqa: E402
from tests.helpers import GOLDEN, load_npz, randomise, rel_l2  # noqa: E402
from wdno_amd import diffusion_core as K  # noqa: E402
from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D  # noqa: E402
from ddpm.diffusion_2d import GaussianDiffusion as GD2  # noqa: E402

s = int(sys.argv[1]) if len(sys.argv) > 1 else 0
g = load_npz('ref_round3.npz')
meta = json.load(open(os.path.join(GOLDEN, 'ref_round3_manifest.json')))['mseed']
u, d = meta['smoke']['unet'], dict(meta['smoke']['diffusion'])
d['padded_shape'] = tuple(d['padded_shape']); d['ori_shape'] = tuple(d['ori_shape'])
base = {k[len('mseed_base::sm::'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('mseed_base::sm::')}
sub = {k[len(f'mseed_smoke{s}::'):]: g[k] for k in g.files if k.startswith(f'mseed_smoke{s}::')}
net = Unet3D_with_Conv3D(dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_groups=u['resnet_groups'], init_kernel_size=u['init_kernel_size'])
net.load_state_dict(base)
randomise(net, torch.Generator().manual_seed(1000 + s))
sd32 = {k: v.detach().clone() for k, v in net.state_dict().items()}
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}
cfg = dict(dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_groups'])
ns = [torch.from_numpy(sub[f'ddim_noise_{i}']) for i in range(int(sub['ddim_n_noise']))]
init, control = torch.from_numpy(sub['init']), torch.from_numpy(sub['control'])
b32 = D.make_buffers('sigmoid', 1000)
b64 = {k: v.double() for k, v in b32.items()}
dif = GD2(net, loss_layer_weight=torch.ones(1, 1, 42, 1, 1), **d).to('cuda')
shape = tuple(ns[0].shape)
desc = dif._desc(shape, dif.padded_shape)
src = dif._condition_source(shape, 'cuda', init.cuda(), control.cuda(), None)
m64 = lambda a, b: U.unet3d_forward(sd64, a, b, **cfg)
m32 = lambda a, b: U.unet3d_forward(sd32, a, b, **cfg)
x64 = D.smoke_apply_conditions(ns[0].double().clone(), d['padded_shape'], init=init.double(), control=control.double())
x32 = D.smoke_apply_conditions(ns[0].clone(), d['padded_shape'], init=init, control=control)
xh = K.apply_cond(ns[0].cuda().contiguous(), src, desc)
xt = xh.clone()          # hip U-Net + reference-style torch element-wise update on the device
print('start', rel_l2(xh, x64), rel_l2(x32, x64))
with torch.no_grad():
    for i, (time, time_next) in enumerate(D.ddim_times(1000, d['sampling_timesteps'])):
        t = torch.full((shape[0],), time, dtype=torch.long)
        e64, s64 = D.smoke_model_predictions(m64, b64, x64, t, clip_x_start=True, rederive=True)
        e32, s32 = D.smoke_model_predictions(m32, b32, x32, t, clip_x_start=True, rederive=True)
        rawh = dif.model(xh, t.cuda(), None)
        rawt = dif.model(xt, t.cuda(), None)
        pn, xs_t = dif.model_predictions(shape, xt, t.cuda(), None, clip_x_start=True, rederive_pred_noise=True)
        loc64 = m64(xh.cpu().double(), t)            # exact U-Net output at hip's own state
        if time_next < 0:
            x64, x32 = s64, s32
            xh, sh = K.ddim_update(dif, xh, rawh, None, t.cuda(), 0., 0., 0.)
            xt = xs_t
        else:
            x64 = D.smoke_apply_conditions(D.ddim_update(b64, s64, e64, time, time_next, 1.0, ns[i + 1].double()), d['padded_shape'], init=init.double(), control=control.double())
            x32 = D.smoke_apply_conditions(D.ddim_update(b32, s32, e32, time, time_next, 1.0, ns[i + 1]), d['padded_shape'], init=init, control=control)
            sigma, cc, sqrt_an = K.ddim_coefficients(dif._ac_host, time, time_next, 1.0)
            xh, sh = K.ddim_update(dif, xh, rawh, ns[i + 1].cuda(), t.cuda(), sqrt_an, cc, sigma)
            xh = K.apply_cond(xh, src, desc)
            xt = K.apply_cond((xs_t * sqrt_an + cc * pn + sigma * ns[i + 1].cuda()).contiguous(), src, desc)
        print(f't={time}->{time_next}: state vs exact: hip {rel_l2(xh, x64):.2e} hip-torch-update {rel_l2(xt, x64):.2e} cpu32 {rel_l2(x32, x64):.2e} | hip vs cpu32 {rel_l2(xh, x32):.2e}'
              f' | eps(hip state) hip-vs-exact {rel_l2(rawh, loc64):.2e}  x_start hip {rel_l2(sh, s64):.2e} cpu32 {rel_l2(s32, s64):.2e}'
              f' frac unclipped {(s64.abs() < 1).double().mean().item():.3f}')
print('final vs reference: hip', rel_l2(xh, sub['ddim_out']), 'cpu32', rel_l2(x32, sub['ddim_out']))
