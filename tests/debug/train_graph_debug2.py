"""Debugging aid (GPU box): TrainStep.capture under the conditions of tests/test_gpu_graph.py. flags: prior twin ema"""
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
from wdno_amd.trainer import TrainStep, multistep_lr  # noqa: E402

flags = set(sys.argv[1:])


def make():
    torch.manual_seed(4)
    net = Unet2D(dim=32, dim_mults=(1, 2, 4), channels=9, resnet_block_groups=1)
    dif = GD1(net, seq_length=(32, 32), padded_shape=[21, 28], ori_shape=[41, 56], loss_layer_weight=torch.ones(1, 9, 1, 1), is_condition_pad=True, is_condition_u0=True, is_condition_f=True)
    return TrainStep(dif.to('cuda'), lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema='ema' in flags, ema_update_every=2)


x = torch.randn(4, 9, 32, 32, device='cuda') * 0.5
if 'twin' in flags:
    te = make()
    for _ in range(3):
        te.step(x)
ts = make()
if 'prior' in flags:
    ts.step(x)
ts.capture(x, warmup=1)
print(sorted(flags), 'captured')
for _ in range(3):
    out = ts.step(x)
torch.cuda.synchronize()
print(sorted(flags), 'replayed', float(out[0]))
