"""The single-product bf16 convolution path (BASELINE.json configs[1]; WDNO_CONV_MATH=bf16 / ops.CONV_MATH = 'bf16').

Two levels:
  * kernels: every operand is rounded to bf16 (round-to-nearest-even) and multiplied exactly, accumulated in fp32 -> the result
    must equal an fp64 convolution of the bf16-ROUNDED operands to fp32-accumulation accuracy (2e-6), for forward, data gradient
    and weight gradient, on the register-staged and the persistent LDS-DMA kernels. This pins the arithmetic, not a tolerance.
  * model: the documented bf16 tolerance of a whole Burgers training step against the fp32 oracle. bf16 keeps 8 significant
    bits per operand (2^-9 relative rounding), so a full-width U-Net agrees with fp32 to ~1e-2: loss within 2 %, parameter
    gradients within 10 % rel-L2 (median ~2 %). The fp32-equivalent default path meets 1e-5 on the same step
    (tests/test_gpu_fullsize.py); this one cannot and is not the default.
GPU box only."""
import math
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture()
def ops():
    from wdno_amd import ops as o
    o._lib_()
    prev = o.CONV_MATH
    o.CONV_MATH = 'bf16'
    yield o
    o.CONV_MATH = prev
    o.ACT16 = False


def bf(t):
    return t.float().to(torch.bfloat16).double()


def g(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64) * scale


def to_cl(x):
    return x.permute(0, *range(2, x.dim()), 1).contiguous()


def from_cl(x):
    return x.permute(0, x.dim() - 1, *range(1, x.dim() - 1)).contiguous()


CASES = [
    ('3x3x3_64', (1, 64, 8, 20, 20), (64, 64, 3, 3, 3), 1, 1),
    ('3x3x3_wide', (2, 128, 4, 16, 16), (256, 128, 3, 3, 3), 1, 1),
    ('3x3x3_oddC', (1, 36, 8, 16, 16), (20, 36, 3, 3, 3), 1, 1),
    ('7x7x7_init', (1, 42, 8, 16, 16), (64, 42, 7, 7, 7), 1, 3),
    ('1x1x1', (4, 256, 4, 16, 16), (128, 256, 1, 1, 1), 1, 0),
    ('2d_3x3', (8, 128, 16, 16), (128, 128, 3, 3), 1, 1),
    ('2d_3x3_1024', (16, 1024, 8, 8), (1024, 1024, 3, 3), 1, 1),
    ('down_144', (2, 64, 4, 32, 32), (64, 64, 1, 4, 4), (1, 2, 2), (0, 1, 1)),
    ('2d_3x3_many_tiles', (32, 64, 64, 64), (128, 64, 3, 3), 1, 1),      # 512 tiles of 256 x 128: the wide tap-resident tiles of the bf16 mode
    ('persist_k96', (1, 16, 24, 40, 40), (96, 16, 3, 3, 3), 1, 1),
    ('persist_k32b', (4, 16, 24, 40, 40), (32, 16, 3, 3, 3), 1, 1),
]


@pytest.mark.parametrize('mode', [0, 5], ids=['default', 'register_staged'])
@pytest.mark.parametrize('name,xs,ws,stride,padding', CASES, ids=[c[0] for c in CASES])
def test_bf16_conv_equals_exact_product_of_rounded_operands(ops, name, xs, ws, stride, padding, mode):
    if mode == 5 and name == '2d_3x3_1024':
        pytest.skip('one kernel family is enough for the largest case')
    nd = len(ws) - 2
    seed = sum(name.encode()) % 1000
    x, w, b = g(xs, seed), g(ws, seed + 1, 1.0 / math.sqrt(np.prod(ws[1:]))), g((ws[0],), seed + 2)
    conv = F.conv3d if nd == 3 else F.conv2d
    grad = torch.nn.grad
    yr = conv(bf(x), bf(w), b, stride=stride, padding=padding)
    go = g(tuple(yr.shape), seed + 4)
    if nd == 3:
        gxr = grad.conv3d_input(x.shape, bf(w), bf(go), stride=stride, padding=padding)
        gwr = grad.conv3d_weight(bf(x), w.shape, bf(go), stride=stride, padding=padding)
    else:
        gxr = grad.conv2d_input(x.shape, bf(w), bf(go), stride=stride, padding=padding)
        gwr = grad.conv2d_weight(bf(x), w.shape, bf(go), stride=stride, padding=padding)
    cp, kp = ops.pad4(xs[1]), ops.pad4(ws[0])
    xd = torch.zeros(xs[0], *xs[2:], cp, dtype=torch.float64)
    xd[..., :xs[1]] = to_cl(x)
    xd = xd.float().to(DEV).requires_grad_(True)
    wd, bd = w.float().to(DEV).requires_grad_(True), b.float().to(DEV).requires_grad_(True)
    lib = ops._lib_()
    lib.wdno_set_debug(mode)
    ops.PROFILE = {}
    try:
        y = ops.conv_cl(xd, wd, bd, stride=stride, padding=padding)
        god = torch.zeros(tuple(y.shape), dtype=torch.float64)
        god[..., :ws[0]] = to_cl(go)
        y.backward(god.float().to(DEV))
        torch.cuda.synchronize()
    finally:
        used, ops.PROFILE = set(ops.PROFILE), None
        lib.wdno_set_debug(0)
    assert any('h3' in k for k in used), used
    assert rel_l2(from_cl(y.detach().cpu())[:, :ws[0]], yr) < 2e-6, 'forward'
    # (the data gradient of the strided down-sampling convolution runs as four parity-class convolutions on the same bf16 operands)
    assert rel_l2(from_cl(xd.grad.cpu())[:, :xs[1]], gxr) < 2e-6, 'dgrad'
    assert rel_l2(wd.grad.cpu(), gwr) < 2e-6, 'wgrad'
    assert rel_l2(bd.grad.cpu(), go.float().double().sum(dim=[0] + list(range(2, go.dim())))) < 2e-6, 'bias grad'
    # and the distance to the un-rounded fp64 convolution is bf16-sized, not fp32-sized
    y64 = conv(x, w, b, stride=stride, padding=padding)
    assert 1e-4 < rel_l2(from_cl(y.detach().cpu())[:, :ws[0]], y64) < 1e-2


def test_bf16_transposed_conv(ops):
    x, w, b = g((2, 64, 4, 16, 16), 1), g((64, 64, 1, 4, 4), 2, 0.03), g((64,), 3)
    yr = F.conv_transpose3d(bf(x), bf(w), b, stride=(1, 2, 2), padding=(0, 1, 1))
    xd = to_cl(x).float().to(DEV).requires_grad_(True)
    wd = w.float().to(DEV).requires_grad_(True)
    y = ops.conv_transpose_cl(xd, wd, b.float().to(DEV))
    assert rel_l2(from_cl(y.detach().cpu()), yr) < 2e-6
    go = g(tuple(yr.shape), 4)
    y.backward(to_cl(go).float().to(DEV))
    gxr = F.conv3d(bf(go), bf(w), None, stride=(1, 2, 2), padding=(0, 1, 1))
    assert rel_l2(from_cl(xd.grad.cpu()), gxr) < 2e-6
    assert torch.isfinite(wd.grad).all()


@pytest.mark.parametrize('batch', [4, 32, 256])
def test_bf16_burgers_train_step_vs_autocast_arbiter(ops, batch):
    """Full-width Unet2D(dim=128) training step [B, 9, 64, 64] on the single-product bf16 path (BASELINE configs[1]); B = 32 is the smallest batch
    at which the level-0 layers take the 256 x 128 single-plane tiles that carry configs[1]'s batch of 256 (asserted). The reference's
    semantics for that configuration are accelerate's mixed precision (train_diffusion.py:62, 71-74): fp32 master weights, the forward under
    torch.autocast(bfloat16). Three evaluations of the same step on the same weights and inputs: HIP bf16, the oracle under CPU
    autocast(bfloat16) (= what the reference computes), the oracle in fp64 (the arbiter). Gate: the HIP path is no further from exact than
    1.5 x the autocast reference -- loss, median and worst parameter gradient."""
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion
    from oracle import diffusion_ref as D, unet_ref as U
    import os
    if batch == 256:
        # BASELINE configs[1] at its OWN batch: the fp64 arbiter and the autocast oracle of 256 samples take ~5 minutes and
        # ~150 GB of host memory -- run on request (WDNO_SLOW_TESTS=1; its output of round 6 is committed as profiles/r06_bf16_batch256_arbiter.txt)
        avail = 0
        with open('/proc/meminfo') as fh:
            for line in fh:
                if line.startswith('MemAvailable'):
                    avail = int(line.split()[1]) / 1e6
        if os.environ.get('WDNO_SLOW_TESTS') != '1' or avail < 400:
            pytest.skip(f'batch 256 arbiter: WDNO_SLOW_TESTS=1 and >= 400 GB of host memory (have {avail:.0f} GB)')
    torch.manual_seed(1)
    net = Unet2D(dim=128, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32 if batch < 256 else 64))
    gen = torch.Generator().manual_seed(6)
    x0 = torch.randn(batch, 9, 64, 64, generator=gen) * 0.5
    noise = torch.randn(batch, 9, 64, 64, generator=gen)
    t = torch.tensor(([77, 805, 310, 999] * 64)[:batch])
    lw = torch.ones(1, 9, 1, 1)

    def oracle(dt, autocast):
        sd = {k: (v.clone().to(dt).requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        model = lambda x, tt: U.unet2d_forward(sd, x, tt, dim=128, dim_mults=(1, 2, 4, 8), groups=1)
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
            loss = D.burgers_p_losses(model, D.make_buffers('cosine', 1000), x0.to(dt), t, noise.to(dt), padded_shape=[41, 60],
                                      loss_layer_weight=lw.to(dt), flags=dict(pad=True, u0=True, uT=False, f=True))
        loss.float().backward() if autocast else loss.backward()
        return float(loss.detach()), {k: v.grad for k, v in sd.items() if v.is_floating_point() and v.grad is not None}

    l_e, g_e = oracle(torch.float64, False)
    l_a, g_a = oracle(torch.float32, True)
    dif = GaussianDiffusion(net, seq_length=(64, 64), padded_shape=[41, 60], ori_shape=[81, 120], loss_layer_weight=lw,
                            is_condition_pad=True, is_condition_u0=True, is_condition_f=True).to(DEV)
    ops.PROFILE = {}
    loss = dif.p_losses(x0.to(DEV), t.to(DEV), noise=noise.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    used, ops.PROFILE = set(ops.PROFILE), None
    assert any('h3d' in k or 'h3t' in k for k in used)
    assert ('conv_fwd_h3t_kernel<256,128>' in used) == (batch >= 32), used
    names = [k for k, _ in net.named_parameters()]
    e_h = sorted(rel_l2(p.grad, g_e[k]) for k, p in net.named_parameters())
    e_a = sorted(rel_l2(g_a[k], g_e[k]) for k in names)
    rl_h, rl_a = abs(loss.item() - l_e) / abs(l_e), abs(l_a - l_e) / abs(l_e)
    mid = len(names) // 2
    print(f'bf16 Burgers step vs exact: loss rel HIP {rl_h:.3e} / autocast oracle {rl_a:.3e}; gradient rel-L2 median HIP {e_h[mid]:.3e} / autocast {e_a[mid]:.3e}, '
          f'worst HIP {e_h[-1]:.3e} / autocast {e_a[-1]:.3e}')
    # (the loss is ONE number: its error is a draw from the bf16 noise, which the gradients measure 276 times -- the autocast oracle's own draw
    # can be 30 x below its typical size, so the loss is held to the typical size)
    assert rl_h <= 1.5 * max(rl_a, e_a[mid])
    assert e_h[mid] <= 1.5 * e_a[mid] and e_h[-1] <= 1.5 * e_a[-1]


def test_bf16_smoke_train_step_vs_autocast_arbiter(ops):
    """the smoke model at the bench's own tensor size, [2, 24, 42, 40, 40] through Unet3D_with_Conv3D(dim=64, (1,2,4), 42), on
    the single-product bf16 kernels (bench.py leg `smoke_bf16` / --workload smoke-bf16). Same three evaluations as the Burgers test above: HIP
    bf16, the oracle under CPU autocast(bfloat16) (accelerate mixed-precision semantics), the oracle in fp64 (arbiter). The seven 64-channel
    attention blocks stay on their fused fp32-equivalent kernels (asserted: they ran), so the HIP step may only be closer to exact."""
    import os
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion
    from oracle import diffusion_ref as D, unet_ref as U
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    torch.manual_seed(0)
    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    lw = torch.linspace(1.0, 22.0, 42).reshape(1, 1, 42, 1, 1)
    gen = torch.Generator().manual_seed(5)
    x0 = torch.randn(2, 24, 42, 40, 40, generator=gen) * 0.5
    noise = torch.randn(2, 24, 42, 40, 40, generator=gen)
    t = torch.tensor([37, 911])

    def oracle(dt, autocast):
        sd = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() and not k.endswith('freqs') else (v.to(dt) if v.is_floating_point() else v))
              for k, v in sd0.items()}
        model = lambda x, tt: U.unet3d_forward(sd, x, tt, dim=64, dim_mults=(1, 2, 4), groups=8)
        buf = {k: v.to(dt) for k, v in D.make_buffers('sigmoid', 1000).items()}
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
            loss = D.smoke_p_losses(model, buf, x0.to(dt), t, noise.to(dt), padded_shape=(18, 34, 34), loss_layer_weight=lw.to(dt))
        loss.float().backward() if autocast else loss.backward()
        return float(loss.detach()), {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}

    l_e, g_e = oracle(torch.float64, False)
    l_a, g_a = oracle(torch.float32, True)
    dif = GaussianDiffusion(net, lw, True, True, True, False, 'bior1.3', 'zero', (18, 34, 34), (32, 64, 64), image_size=40, frames=24).to(DEV)
    ops.PROFILE = {}
    loss = dif.p_losses(x0.to(DEV), t.to(DEV), noise=noise.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    used, ops.PROFILE = set(ops.PROFILE), None
    assert any('h3t' in k for k in used) and any('wgrad_h3' in k for k in used), used
    assert 'tattn_fused_bwd_kernel' in used and 'lattn_fused_bwd_kernels' in used, used        # the blocks the leg's label says stayed fp32-equivalent
    names = [k for k, p in net.named_parameters() if p.grad is not None and k in g_e]
    assert len(names) == 228
    grads = dict(net.named_parameters())
    e_h = sorted(rel_l2(grads[k].grad, g_e[k]) for k in names)
    e_a = sorted(rel_l2(g_a[k], g_e[k]) for k in names)
    rl_h, rl_a = abs(loss.item() - l_e) / abs(l_e), abs(l_a - l_e) / abs(l_e)
    mid = len(names) // 2
    print(f'bf16 smoke step [2,24,42,40,40] vs exact: loss rel HIP {rl_h:.3e} / autocast oracle {rl_a:.3e}; gradient rel-L2 median HIP {e_h[mid]:.3e} / '
          f'autocast {e_a[mid]:.3e}, worst HIP {e_h[-1]:.3e} / autocast {e_a[-1]:.3e}')
    assert rl_h <= 1.5 * max(rl_a, e_a[mid])
    assert e_h[mid] <= 1.5 * e_a[mid] and e_h[-1] <= 1.5 * e_a[-1]


def test_act16_conv_output_is_the_rounded_fp32_output(ops):
    """Single-product mode stores the output of a convolution whose only reader is a GroupNorm as bf16 (ops.ACT16, wdno_conv_fwd_bf16_ex): the
    stored values are the round-to-nearest-even of what the fp32-output launch writes, on the tap-resident, the chunked and the register-staged
    kernel; without the caller's to_norm statement nothing changes."""
    for name, xs, ws, pad in (('tap', (2, 64, 8, 20, 20), (64, 64, 3, 3, 3), 1), ('chunked_1x1', (4, 256, 4, 16, 16), (128, 256, 1, 1, 1), 0),
                              ('register_staged', (1, 32, 2, 8, 8), (64, 32, 3, 3, 3), 1), ('2d', (8, 128, 16, 16), (128, 128, 3, 3), 1)):
        x, w, b = g(xs, 11).float(), (g(ws, 12) / math.sqrt(np.prod(ws[1:]))).float(), g((ws[0],), 13).float()
        xd, wd, bd = to_cl(x).to(DEV), w.to(DEV), b.to(DEV)
        ops.ACT16 = True
        y16 = ops.conv_cl(xd, wd, bd, padding=pad, grad_planes=True, to_norm=True)
        y_plain = ops.conv_cl(xd, wd, bd, padding=pad, grad_planes=True)
        ops.ACT16 = False
        y32 = ops.conv_cl(xd, wd, bd, padding=pad, grad_planes=True, to_norm=True)
        ops.ACT16 = True
        assert y32.dtype == torch.float32 and y_plain.dtype == torch.float32 and torch.equal(y32, y_plain), name
        if y16.dtype == torch.bfloat16:
            assert torch.equal(y16, y32.to(torch.bfloat16)), name
        else:                             # (layers below the split-kernel thresholds keep fp32 storage)
            assert name == 'register_staged' and torch.equal(y16, y32), name
    ops.ACT16 = False


@pytest.mark.parametrize('shape,groups', [((2, 6, 20, 20, 64), 8), ((3, 16, 16, 128), 1), ((2, 4, 8, 8, 24), 4)])
def test_act16_groupnorm_reads_bf16_storage_exactly(ops, shape, groups):
    """GroupNorm on a bf16-stored activation == GroupNorm on the same values stored as fp32, bit for bit, in every form the U-Nets use: fp32
    output, planes output, + residual; and backward (dx planes, column sums, parameter gradients) with dy stored as fp32 and as bf16."""
    gen = torch.Generator().manual_seed(sum(shape))
    c = shape[-1]
    x16 = (torch.randn(shape, generator=gen) * 2 + 0.3).to(torch.bfloat16).to(DEV)
    gam, bet = (torch.rand(c, generator=gen) + 0.5).to(DEV), (torch.randn(c, generator=gen) * 0.1).to(DEV)
    ss = (torch.randn(shape[0], 2 * c, generator=gen) * 0.2).to(DEV)
    res = torch.randn(shape, generator=gen).to(DEV)
    gy32 = torch.randn(shape, generator=gen).to(torch.bfloat16).float().to(DEV)

    planes_ok = c % 8 == 0 and (c // 8) & (c // 8 - 1) == 0          # the channel counts whose backward hands dx over as planes (the only ones a
                                                                      # convolution stores as bf16 in front of a norm)

    def run(xin, gy):
        out = {}
        if not planes_ok:
            with torch.no_grad():
                out['y'] = ops.groupnorm_act(xin, gam, bet, groups, ss, act=True).clone()
            return out
        xin = xin.clone().requires_grad_(True)
        xin._wdno_grad_planes = True                      # (what conv_cl(..., grad_planes=True) states about its output)
        g_, b_, s_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True), ss.clone().requires_grad_(True)
        got = []
        xin.register_hook(lambda t: got.append(t))
        y = ops.groupnorm_act(xin, g_, b_, groups, s_, act=True)
        out['y'] = y.detach().clone()
        y.backward(gy)
        po = got[0]._wdno_planes_only
        out['dx_hi'], out['dx_colsum'] = po[0][0].clone(), po[1].clone()
        out['dg'], out['db'], out['dss'] = g_.grad.clone(), b_.grad.clone(), s_.grad.clone()
        with torch.no_grad():
            yp = ops.groupnorm_act(xin.detach(), gam, bet, groups, ss, act=True, out_planes=True)
            out['y_planes'] = yp._wdno_planes[0][0].clone()
            out['y_add'] = ops.groupnorm_act_add(xin.detach(), gam, bet, groups, res, ss).clone()
        return out
    a = run(x16, gy32)
    b = run(x16.float(), gy32)
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    if planes_ok:                                             # dy stored as bf16 as well (the data gradient of the convolution behind the norm)
        d = run(x16, gy32.to(torch.bfloat16))
        for k in ('dx_hi', 'dx_colsum', 'dg', 'db', 'dss'):
            assert torch.equal(d[k], b[k]), k


def test_act16_changes_a_training_step_only_by_bf16_rounding(ops):
    """A Burgers training step (dim 32, grids 64 .. 8) with bf16-stored convolution outputs against the same step with fp32 storage: the loss
    agrees to bf16 rounding and the bf16 tensors were really used (the convolution outputs in front of the norms are bfloat16)."""
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion
    res = {}
    seen = []
    orig = ops.groupnorm_act

    def spy(x, *a, **k):
        seen.append(x.dtype)
        return orig(x, *a, **k)
    for flag in (True, False):
        ops.ACT16 = flag
        torch.manual_seed(3)
        net = Unet2D(dim=32, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
        dif = GaussianDiffusion(net, seq_length=(64, 64), padded_shape=[41, 60], ori_shape=[81, 120], loss_layer_weight=torch.ones(1, 9, 1, 1),
                                is_condition_pad=True, is_condition_u0=True, is_condition_f=True).to(DEV)
        gen = torch.Generator().manual_seed(4)
        x0 = (torch.randn(8, 9, 64, 64, generator=gen) * 0.5).to(DEV)
        t = torch.randint(0, 1000, (8,), generator=gen).to(DEV)
        noise = torch.randn(8, 9, 64, 64, generator=gen).to(DEV)
        seen.clear()
        ops.groupnorm_act = spy
        try:
            loss = dif.p_losses(x0, t, noise=noise)
            loss.backward()
        finally:
            ops.groupnorm_act = orig
        torch.cuda.synchronize()
        res[flag] = (float(loss), {k: p.grad.clone() for k, p in net.named_parameters()}, list(seen))
    ops.ACT16 = False
    assert torch.bfloat16 in res[True][2] and torch.bfloat16 not in res[False][2]
    assert abs(res[True][0] - res[False][0]) < 2e-2 * abs(res[False][0])
    errs = sorted(rel_l2(res[True][1][k], res[False][1][k]) for k in res[True][1])
    print('ACT16 vs fp32 storage, gradient rel-L2: median', errs[len(errs) // 2], 'worst', errs[-1])
    assert errs[len(errs) // 2] < 5e-2


def test_bf16_training_reduces_the_loss(ops):
    """20 optimiser steps on one fixed batch with fp32 master weights: the loss must fall like it does on the fp32 path."""
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion
    from wdno_amd.trainer import TrainStep
    losses = {}
    for mode in ('bf16', 'f16x3'):
        ops.CONV_MATH = mode
        torch.manual_seed(3)
        net = Unet2D(dim=32, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)      # mid attention over 8 x 8 tokens
        dif = GaussianDiffusion(net, seq_length=(64, 64), padded_shape=[41, 60], ori_shape=[81, 120], loss_layer_weight=torch.ones(1, 9, 1, 1),
                                is_condition_pad=True, is_condition_u0=True, is_condition_f=True).to(DEV)
        ts = TrainStep(dif, lr=2e-4, use_ema=False)
        gen = torch.Generator().manual_seed(4)
        x0 = (torch.randn(8, 9, 64, 64, generator=gen) * 0.5).to(DEV)
        t = torch.randint(0, 1000, (8,), generator=gen).to(DEV)
        noise = torch.randn(8, 9, 64, 64, generator=gen).to(DEV)
        losses[mode] = [float(ts.step_with(x0, t, noise)[0]) for _ in range(20)]
    ops.CONV_MATH = 'bf16'
    print('loss bf16 ', [round(v, 4) for v in losses['bf16'][::4]], '\nloss f16x3', [round(v, 4) for v in losses['f16x3'][::4]])
    assert losses['bf16'][-1] < 0.8 * losses['bf16'][0]
    assert abs(losses['bf16'][-1] - losses['f16x3'][-1]) < 0.1 * losses['f16x3'][-1]


def test_bf16_mode_sampling_uses_the_wide_attention_blocks(ops):
    """The single-product mode keeps the fused attention blocks on their split-fp16 projections (ops._in_f16x3), the 128- / 256-channel ones
    (csrc/attn_fused_wide.hip, csrc/linattn_fused_wide.hip, fragment-ordered operands from the f16x3 family of weight plans) included: the
    denoiser under no_grad with every convolution on one bf16 plane runs them, and its output agrees with the same pass with those blocks layer
    by layer (bf16 projections there) to the bf16 tolerance while being closer to the fp32-equivalent pass."""
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    from video_diffusion_pytorch import video_diffusion_pytorch_conv3d as V
    torch.manual_seed(21)
    net = V.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42).to(DEV)
    x, t = (torch.randn(2, 24, 42, 16, 16) * 0.7).to(DEV), torch.tensor([433, 17], device=DEV)
    with torch.no_grad():
        ops.PROFILE = {}
        out = net(x, t)
        n_t, n_l = len(ops.PROFILE.get('tattn_fused_fwd_kernel', [])), len(ops.PROFILE.get('lattn_fused_fwd_kernels', []))
        ops.PROFILE = None
        assert torch.equal(out, net(x, t))
        ops.FUSED_TATTN_WIDE = ops.FUSED_LATTN_WIDE = False
        try:
            out_narrow = net(x, t)
        finally:
            ops.FUSED_TATTN_WIDE = ops.FUSED_LATTN_WIDE = True
        ops.CONV_MATH = 'f16x3'
        ref = net(x, t)
        ops.CONV_MATH = 'bf16'
    assert n_t == 5 and n_l == 4, (n_t, n_l)          # 2 x 16 x 16 sequences / 8 x 8 tokens: the second level's blocks are taken, the 4 x 4 ones are not
    e_w, e_n, e_wn = rel_l2(out, ref), rel_l2(out_narrow, ref), rel_l2(out, out_narrow)
    print(f'bf16-mode U-Net forward vs fp32-equivalent: wide blocks fused {e_w:.2e}, layer by layer {e_n:.2e}; between them {e_wn:.2e}')
    assert e_w < 3e-2 and e_w <= 1.1 * e_n and e_wn < 3e-2
