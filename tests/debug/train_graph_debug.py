"""Debugging aid (GPU box): which part of the Burgers training step breaks hipStreamEndCapture? mode: fwd | fwdbwd | full; model: tiny | wide"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wdno_amd import tree_path, ops  # noqa: E402
for _t in ('third_party', 'smoke', 'burgers'):
    sys.path.insert(0, tree_path(_t))
from ddpm_burgers.unet import Unet2D  # noqa: E402
from ddpm_burgers.diffusion_1d import GaussianDiffusion as GD1  # noqa: E402
from wdno_amd.trainer import TrainStep  # noqa: E402

mode, model = sys.argv[1], sys.argv[2]
torch.manual_seed(4)
if model == 'tiny':
    net = Unet2D(dim=32, dim_mults=(1, 2, 4), channels=9, resnet_block_groups=1)
    dif = GD1(net, seq_length=(32, 32), padded_shape=[21, 28], ori_shape=[41, 56], loss_layer_weight=torch.ones(1, 9, 1, 1), is_condition_pad=True, is_condition_u0=True, is_condition_f=True)
    x = torch.randn(4, 9, 32, 32, device='cuda') * 0.5
else:
    net = Unet2D(dim=128, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
    dif = GD1(net, seq_length=(64, 64), padded_shape=[41, 60], ori_shape=[81, 120], loss_layer_weight=torch.ones(1, 9, 1, 1), is_condition_pad=True, is_condition_u0=True, is_condition_f=True)
    x = torch.randn(4, 9, 64, 64, device='cuda') * 0.5
ts = TrainStep(dif.to('cuda'), lr=1e-3, use_ema=False)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        ts.step(x)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
t = torch.zeros(4, device='cuda', dtype=torch.long)
nz = torch.zeros_like(x)
ts.opt.buf.prepare_capture()
ts.opt.zero_grad()
g = torch.cuda.CUDAGraph()
if mode == 'fwd':
    with torch.no_grad(), ops.graph_capture(g, stream=side):
        loss = dif.p_losses(x, t, noise=nz)
else:
    with ops.graph_capture(g, stream=side):
        loss = dif.p_losses(x, t, noise=nz)
        loss.backward()
        if mode == 'full':
            ts.opt.buf.gather_grads(capture=True)
print(mode, model, 'captured')
g.replay()
torch.cuda.synchronize()
print(mode, model, 'replayed', float(loss))
