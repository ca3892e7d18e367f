"""Where does the HIP forward of the (tiny golden) smoke U-Net pick up its round-off? Records the input of every channel
LayerNorm and the output of every ResnetBlock in the HIP modules and in the CPU oracle (fp32 and fp64), and prints the relative
error of each against the fp64 evaluation, in execution order: a jump between two rows names the operator in between."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wdno_amd import tree_path  # noqa: E402
for _t in ('third_party', 'smoke', 'burgers'):
    sys.path.insert(0, tree_path(_t))
from oracle import unet_ref as U  # noqa: E402
from tests.helpers import load_npz, manifest, rel_l2, weights  # noqa: E402

M = manifest()
g = load_npz('ref_smoke_diffusion.npz')
u = M['smoke_diffusion']['unet']
tval = int(os.environ.get('T', '999'))
full = os.environ.get('FULL', '0') == '1'
if full:
    torch.manual_seed(0)
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D as _U
    sd0 = {k: v.clone() for k, v in _U(dim=64, dim_mults=(1, 2, 4), channels=42).state_dict().items()}
    cfg = dict(dim=64, dim_mults=(1, 2, 4), groups=8)
    x = torch.randn(1, 24, 42, 40, 40, generator=torch.Generator().manual_seed(3))
else:
    sd0 = weights(g, 'w::model.')
    cfg = dict(dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_groups'])
    x = torch.from_numpy(g['psample_xt'])
t = torch.full((x.shape[0],), tval, dtype=torch.long)


def oracle_trace(dt):
    rec = []
    orig_ln, orig_rb = U.channel_layernorm, U.resnet_block

    def ln(xx, gg, eps=1e-5):
        rec.append(('ln_in', xx.detach().double()))
        return orig_ln(xx, gg, eps)

    def rb(*a, **k):
        o = orig_rb(*a, **k)
        rec.append(('res_out', o.detach().double()))
        return o
    U.channel_layernorm, U.resnet_block = ln, rb
    try:
        sd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd0.items()}
        with torch.no_grad():
            out = U.unet3d_forward(sd, x.to(dt), t, **cfg)
    finally:
        U.channel_layernorm, U.resnet_block = orig_ln, orig_rb
    rec.append(('out', out.double().permute(0, 2, 1, 3, 4)))
    return rec


def hip_trace():
    from video_diffusion_pytorch import video_diffusion_pytorch_conv3d as V
    net = V.Unet3D_with_Conv3D(dim=cfg['dim'], dim_mults=cfg['dim_mults'], channels=42, resnet_groups=cfg['groups'])
    net.load_state_dict(sd0)
    net = net.cuda()
    rec = []
    cl = lambda y, c: y.detach().double().cpu().permute(0, 4, 1, 2, 3)[:, :c]
    for m in net.modules():
        if isinstance(m, V.PreNorm):         # (the fused attention blocks never call their LayerNorm module: hook the PreNorm around it)
            m.register_forward_pre_hook(lambda mod, inp: rec.append(('ln_in', cl(inp[0], mod.norm.gamma.shape[1]))))
        if isinstance(m, V.ResnetBlock):
            m.register_forward_hook(lambda mod, inp, out: rec.append(('res_out', cl(out, mod.block2.proj.weight.shape[0]))))
    with torch.no_grad():
        out = net(x.cuda(), t.cuda())
    rec.append(('out', out.double().cpu().permute(0, 2, 1, 3, 4)))
    return rec


def main():
    """MODES=f16x3,fp32 (default both): the HIP trace under each convolution arithmetic, side by side."""
    from wdno_amd import ops
    modes = os.environ.get('MODES', 'f16x3,fp32').split(',')
    r64, r32 = oracle_trace(torch.float64), oracle_trace(torch.float32)
    traces = {}
    for mode in modes:
        ops.CONV_MATH = mode
        ops.bump_weight_epoch()
        traces[mode] = hip_trace()
        assert len(r64) == len(traces[mode]) == len(r32), (len(r64), len(r32), len(traces[mode]))
    ops.CONV_MATH = 'f16x3'
    print(f'{"#":>3} {"point":8} {"shape":24} {"cpu32 vs exact":>15} ' + ' '.join(f'{"hip " + m + " vs exact":>20} {"ratio":>6}' for m in modes))
    rows = []
    for i, ((n64, a), (n32, b)) in enumerate(zip(r64, r32)):
        e32 = rel_l2(b, a)
        cells, row = [], {'i': i, 'point': n64, 'shape': list(a.shape), 'cpu32': e32}
        for m in modes:
            nh, c = traces[m][i]
            assert n64 == nh, (i, n64, nh)
            eh = rel_l2(c, a)
            row[m] = eh
            cells.append(f'{eh:20.3e} {eh / max(e32, 1e-30):6.2f}')
        rows.append(row)
        print(f'{i:3d} {n64:8} {str(tuple(a.shape)):24} {e32:15.3e} ' + ' '.join(cells))
    return rows


if __name__ == '__main__':
    main()
