"""Rows T2, D6 (guidance hook), 8f-2 and 8f-3 against reference-generated fixtures (tests/golden/ref_round2.npz, written by
tests/golden/make_ref_round2_golden.py from the imported reference). GPU box only.

  * T2   : three optimiser steps of the smoke training loop (smoke/ddpm/diffusion_2d.py:1257-1293) through TrainStep
  * D6   : sample(..., design_fn=...) / sample(..., nablaJ=..., J_scheduler=...) through model_predictions' guidance hook
           (diffusion_2d.py:723-754, diffusion_1d.py:205-227, model_utils.py:35-50)
  * 8f-3 : the super-resolution cascade of smoke/inference_2d.py:155-215, stage by stage and end to end, finished by the IDWT
Unguided chains are compared at 1e-5; chains that start at t = 999 (ill conditioned: the reference's own fp32 result is 2e-5 .. 6e-5 from the
exact chain) are gated against the fp64 oracle as arbiter -- no further from exact than 1.5 x the reference is (tests/arbiter.py);
index / packing stages are bit-exact."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda'
G = np.load(os.path.join(GOLDEN, 'ref_round2.npz'))
with open(os.path.join(GOLDEN, 'ref_round2_manifest.json')) as f:
    META = json.load(f)


def sub(tag):
    pre = tag + '::'
    return {k[len(pre):]: G[k] for k in G.files if k.startswith(pre)}


def W(gz, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in gz.items() if k.startswith(prefix)}


def noises(gz, tag):
    return iter([torch.from_numpy(gz[f'{tag}_noise_{i}']).to(DEV) for i in range(int(gz[f'{tag}_n_noise']))])


@pytest.fixture(scope='module')
def trees():
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion as GD1
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion as GD2
    return dict(Unet2D=Unet2D, GD1=GD1, Unet3D=Unet3D_with_Conv3D, GD2=GD2)


def _smoke_dif(trees, m, lw, **over):
    u, d = m['unet'], dict(m['diffusion'])
    d.update(over)
    for k in ('padded_shape', 'ori_shape'):
        if isinstance(d.get(k), list) and d[k] and not isinstance(d[k][0], list):
            d[k] = tuple(d[k])
    net = trees['Unet3D'](dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_groups=u['resnet_groups'],
                          init_kernel_size=u.get('init_kernel_size', 7))
    return trees['GD2'](net, loss_layer_weight=lw, **d)


def test_smoke_three_optimizer_steps_vs_reference(trees):
    """T2: loss -> backward -> clip(1.0) -> Adam(1e-3, (0.9, 0.99)) -> MultiStepLR([5e4, 1.5e5, 3e5], 0.1), three steps."""
    from wdno_amd.trainer import TrainStep, multistep_lr
    gz = sub('t2')
    dif = _smoke_dif(trees, META['t2'], torch.from_numpy(gz['lw']))
    dif.load_state_dict(W(gz, 'w0::'), strict=True)
    dif = dif.to(DEV)
    ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=False)
    from tests import arbiter as A
    le, ge, exact = A.smoke_train3(torch.float64)          # the same three steps by the oracle in fp64: the arbiter
    rel = lambda a, b: abs(a - b) / abs(b)
    for step in range(3):
        x0, t, noise = (torch.from_numpy(gz[f's{step}_{k}']).to(DEV) for k in ('x0', 't', 'noise'))
        loss, gn = ts.step_with(x0, t, noise)
        # loss and gradient norm: the north-star bar (1e-5) against the exact value, or no further from it than the reference's own fp32 run
        assert A.gate_or_bar(rel(loss.item(), le[step]), rel(float(gz[f's{step}_loss']), le[step])), (step, loss.item(), le[step])
        assert A.gate_or_bar(rel(gn.item(), ge[step]), rel(float(gz[f's{step}_gnorm']), ge[step])), (step, gn.item(), ge[step])
    w0 = {k[len('model.'):]: v for k, v in W(gz, 'w0::').items() if k.startswith('model.')}
    ref = {k[len('model.'):]: v for k, v in W(gz, 'w3::').items() if k.startswith('model.')}
    ours = {k[len('model.'):]: v.cpu() for k, v in dif.state_dict().items() if k.startswith('model.')}
    A.check_updates('T2 smoke', w0, ours, ref, exact)


def test_smoke_guided_sampling_vs_reference(trees):
    gz = sub('gs')
    dif = _smoke_dif(trees, META['gs'], torch.from_numpy(gz['lw']))
    dif.load_state_dict(W(gz, 'w::'), strict=True)
    dif = dif.to(DEV)
    target, wgt = torch.from_numpy(gz['target']).to(DEV), torch.from_numpy(gz['wgt']).to(DEV)
    init, init_u = torch.from_numpy(gz['init']).to(DEV), torch.from_numpy(gz['init_u']).to(DEV)
    calls = []

    def design_fn(x, low=None, init=None, init_u=None):
        calls.append(tuple(x.shape))
        J = 0.5 * (wgt * (x - target) ** 2).sum() + 0.1 * ((x[:, 0, 0] - init_u) ** 2).sum()
        return torch.autograd.grad(J, x)[0]
    # one guided model_predictions call
    xt = torch.from_numpy(gz['mp_xt']).to(DEV)
    tt = torch.full((2,), 300, device=DEV, dtype=torch.long)
    with torch.no_grad():
        mp = dif.model_predictions(tuple(xt.shape), xt, tt, clip_x_start=True, rederive_pred_noise=True, design_fn=design_fn, design_guidance='standard',
                                   init=init, init_u=init_u)
    assert rel_l2(mp.pred_noise, gz['mp_pred_noise']) < 1e-5 and rel_l2(mp.pred_x_start, gz['mp_x_start']) < 1e-5
    seq = noises(gz, 'ddim')
    dif.sample_noise = lambda shape, device: next(seq)
    out = dif.sample(batch_size=2, design_fn=design_fn, design_guidance='standard', init=init, init_u=init_u)
    e1 = rel_l2(out, gz['ddim_out'])
    d6 = dict(META['gs']['diffusion']); d6.update(timesteps=6, sampling_timesteps=None)
    dif6 = _smoke_dif(trees, {'unet': META['gs']['unet'], 'diffusion': d6}, torch.from_numpy(gz['lw']))
    dif6.model = dif.model
    dif6 = dif6.to(DEV)
    seq6 = noises(gz, 'ddpm6')
    dif6.sample_noise = lambda shape, device: next(seq6)
    out6 = dif6.sample(batch_size=2, design_fn=design_fn, design_guidance='standard-alpha', init=init, init_u=init_u)
    e2 = rel_l2(out6, gz['ddpm6_out'])
    from tests import arbiter as A
    ex = A.smoke_guided(torch.float64)                      # the same two guided chains by the oracle in fp64
    h1, r1 = rel_l2(out, ex['ddim']), rel_l2(gz['ddim_out'], ex['ddim'])
    h2, r2 = rel_l2(out6, ex['ddpm6']), rel_l2(gz['ddpm6_out'], ex['ddpm6'])
    print('smoke guided chains: ddim4 hip vs reference', e1, 'hip vs exact', h1, 'reference vs exact', r1, '| ddpm6', e2, h2, r2)
    # DDIM-4 from t = 999: the guidance term enters pred_noise BEFORE x_start = c1 x - c2 eps (c2 = 1.8e3) is clipped, so the few entries that
    # stay inside (-1, 1) carry 1.8e3 x the last-bit differences of whoever evaluates the chain -- the reference's own fp32 result is 2.1e-5 from
    # the exact chain. Gate: HIP no further from exact than 1.5 x that (and from the reference no further than the two distances together).
    assert len(calls) == 1 + 4 + 6
    assert A.gate(h1, r1) and e1 < 4.5e-5, (e1, h1, r1)             # measured 2.2e-5 (round 3/4): ceiling 2 x that
    assert e2 < 1e-5 and A.gate_or_bar(h2, r2), (e2, h2, r2)


def test_burgers_guided_sampling_vs_reference(trees):
    from ddpm_burgers import model_utils as MU
    gz, m = sub('gb'), META['gb']
    u, d = m['unet'], dict(m['diffusion'])
    d['seq_length'] = tuple(d['seq_length'])
    net = trees['Unet2D'](dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_block_groups=u['resnet_block_groups'])
    dif = trees['GD1'](net, loss_layer_weight=torch.from_numpy(gz['lw']), **d)
    dif.load_state_dict(W(gz, 'w::'), strict=True)
    dif = dif.to(DEV)
    targ, wq = torch.from_numpy(gz['targ']).to(DEV), torch.from_numpy(gz['wq']).to(DEV)
    nablaJ = MU.get_nablaJ(lambda x: 0.5 * (wq * (x - targ) ** 2).sum(dim=(1, 2, 3)))
    u_init = torch.from_numpy(gz['u_init']).to(DEV)
    seq = noises(gz, 'ddim')
    dif.sample_noise = lambda shape, device: next(seq)
    out = dif.sample(batch_size=2, u_init=u_init, nablaJ=nablaJ, J_scheduler=MU.get_scheduler('cosine'))
    e1 = rel_l2(out, gz['ddim_out'])
    d6 = dict(d); d6.update(timesteps=6, sampling_timesteps=None)
    dif6 = trees['GD1'](net, loss_layer_weight=torch.from_numpy(gz['lw']), **d6).to(DEV)
    seq6 = noises(gz, 'ddpm6')
    dif6.sample_noise = lambda shape, device: next(seq6)
    out6 = dif6.sample(batch_size=2, u_init=u_init, nablaJ=nablaJ, J_scheduler=lambda t: 0.2)
    e2 = rel_l2(out6, gz['ddpm6_out'])
    print('burgers guided chains vs reference: ddim4', e1, 'ddpm6', e2)
    assert e1 < 1e-5 and e2 < 1e-5


def test_smoke_super_resolution_cascade_vs_reference(trees):
    """inference_2d.py:155-215 at a reduced size. Stage by stage from the reference's own intermediates (index work bit-exact), then
    the whole cascade chained on our side, then the IDWT of the result against the oracle's."""
    import pywt
    import ptwt
    from ddpm.wave_utils import upsample_coef
    from wave_trans_2d import coef_to_tensor, tensor_to_coef
    from oracle import dwt_ref as R
    gz, m = sub('sr'), META['sr']
    shape, pad_t, (pad_x0, pad_x1), N = m['shape'], m['pad_t'], m['pad_x'], m['N']
    common = dict(m['common'])
    mk = lambda ch: trees['Unet3D'](dim=8, dim_mults=(1, 2), channels=ch, resnet_groups=4, init_kernel_size=3)
    m0 = trees['GD2'](mk(42), loss_layer_weight=torch.ones(1, 1, 42, 1, 1), is_super_model=False, padded_shape=tuple(shape[0]), ori_shape=(4, 8, 8),
                      image_size=pad_x0, frames=pad_t, **common)
    m1 = trees['GD2'](mk(82), loss_layer_weight=torch.ones(1, 1, 82, 1, 1), is_super_model=True, padded_shape=shape, ori_shape=None, image_size=pad_x1,
                      frames=pad_t, **common)
    m0.load_state_dict(W(gz, 'w0::'), strict=True)
    m1.load_state_dict(W(gz, 'w1::'), strict=True)
    m0, m1 = m0.to(DEV), m1.to(DEV)
    t = lambda k: torch.from_numpy(gz[k]).to(DEV)

    def pack(wave, shp, up_type=None):
        coef = tensor_to_coef(wave[:, :, :40].permute(0, 2, 1, 3, 4), shp, upsample_type=up_type) if up_type else \
            tensor_to_coef(wave[:, :, :40].permute(0, 2, 1, 3, 4), shp)
        return coef, coef_to_tensor(coef).reshape(-1, 5, 8, *shp).reshape(-1, 40, *shp).permute(0, 2, 1, 3, 4)

    def lowres(ret0):
        up = upsample_coef(ret0.contiguous(), shape[1], type='space')
        return up, torch.nn.functional.pad(up, (0, pad_x1 - up.shape[-1], 0, pad_x1 - up.shape[-2], 0, 0, 0, pad_t - up.shape[-4]), 'constant', 0)
    # --- stage by stage, each fed with the reference's intermediate
    _, ret0 = pack(t('wave0'), shape[0])
    assert torch.equal(ret0.cpu(), torch.from_numpy(gz['ret0']))
    up, low = lowres(t('ret0'))
    assert torch.equal(up.cpu(), torch.from_numpy(gz['up'])) and torch.equal(low.cpu(), torch.from_numpy(gz['low']))
    coef1, ret1 = pack(t('wave1'), shape[1], 'space')
    assert torch.equal(coef1[0].cpu(), torch.from_numpy(gz['coef1_yl']))
    assert torch.equal(torch.stack(list(coef1[1].values()), dim=1).cpu(), torch.from_numpy(gz['coef1_yh']))
    assert torch.equal(ret1.cpu(), torch.from_numpy(gz['ret1']))
    seq = noises(gz, 'sr')
    m1.sample_noise = lambda shp, device: next(seq)
    wave1 = m1.sample(batch_size=N, design_fn=None, N_upsample=1, low=t('low'), init=t('init1'), init_u=None, control=t('control1'))
    e_sr = rel_l2(wave1, gz['wave1'])
    # --- the whole cascade on our side
    seq0, seq1 = noises(gz, 'base'), noises(gz, 'sr')
    m0.sample_noise = lambda shp, device: next(seq0)
    m1.sample_noise = lambda shp, device: next(seq1)
    w0 = m0.sample(batch_size=N, design_fn=None, low=None, init=t('init0'), init_u=None, control=t('control0'))
    e_base = rel_l2(w0, gz['wave0'])
    _, r0 = pack(w0, shape[0])
    _, lo = lowres(r0)
    w1 = m1.sample(batch_size=N, design_fn=None, N_upsample=1, low=lo, init=t('init1'), init_u=None, control=t('control1'))
    e_chain = rel_l2(w1, gz['wave1'])
    print('SR cascade vs reference: base sample', e_base, 'SR sample from reference low', e_sr, 'end to end', e_chain)
    # the base chain is DDIM-3 from t = 999 on ONE sample: its first step clips all but ~4 of 10 752 entries of x_start = c1 x - c2 eps
    # (c2 = 1.8e3), and those few carry 1.8e3 x the U-Net's fp32 round-off. Arbiter: the oracle -- the reference's arithmetic -- in fp64.
    # The HIP result may be no further from the exact chain than 1.5 x what the reference's own fp32 evaluation is (measured 2.3e-5 vs
    # 1.9e-5). Its distance to the reference's fp32 output (3.2e-5) is then bounded by the sum of the two; the fp32 oracle on this
    # host lands 6e-6 from the fixture only because two CPU evaluations share their summation orders -- their round-off is
    # correlated, an independent implementation's is not -- so that number is printed, not used as a bar.
    from oracle import diffusion_ref as Dr, unet_ref as Ur
    sd32 = {k[len('model.'):]: v for k, v in W(gz, 'w0::').items() if k.startswith('model.')}
    buf32 = {k: v for k, v in W(gz, 'w0::').items() if not k.startswith('model.')}
    ns0 = [torch.from_numpy(gz[f'base_noise_{i}']) for i in range(int(gz['base_n_noise']))]
    chain = lambda sd, buf, dt: Dr.smoke_ddim_sample(lambda x, tt: Ur.unet3d_forward(sd, x, tt, dim=8, dim_mults=(1, 2), groups=4), buf, [n.to(dt) for n in ns0],
                                                     1000, 3, 1.0, padded_shape=tuple(shape[0]), init=torch.from_numpy(gz['init0']).to(dt),
                                                     control=torch.from_numpy(gz['control0']).to(dt))
    with torch.no_grad():
        o32 = chain(sd32, buf32, torch.float32)
        o64 = chain({k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}, {k: v.double() for k, v in buf32.items()}, torch.float64)
    ref_exact, hip_exact, host_ref = rel_l2(gz['wave0'], o64), rel_l2(w0, o64), rel_l2(o32, gz['wave0'])
    print('   base chain: reference vs exact', ref_exact, 'hip vs exact', hip_exact, "this host's fp32 oracle vs reference", host_ref)
    assert hip_exact < 1.5 * ref_exact + 1e-6 and e_base < 6.5e-5             # measured 3.2e-5: ceiling 2 x that
    # SR stage from the reference's own low-resolution input: 1e-5. End to end (base deviation carried through the SR chain, whose own t = 999
    # step amplifies differences of its conditioning input ~3x): arbiter = the whole cascade by the oracle in fp64 (tests/arbiter.py); the
    # reference's own fp32 cascade is 6.4e-5 from it.
    from tests import arbiter as A
    ex = A.smoke_cascade(torch.float64)
    h_chain, r_chain = rel_l2(w1, ex['wave1']), rel_l2(gz['wave1'], ex['wave1'])
    print('   cascade end to end: hip vs exact', h_chain, 'reference vs exact', r_chain)
    assert e_sr < 1e-5
    assert A.gate(h_chain, r_chain) and e_chain < 2e-4, (e_chain, h_chain, r_chain)      # the reference's own fp32 cascade is 6.4e-5 from exact
    # --- reconstruction: our waverec3 of the unpacked coefficients; arbiter = the exact cascade's IDWT (fp64), yardstick = the oracle's IDWT of the
    # reference's coefficients
    c_ours, _ = pack(w1, shape[1], 'space')
    rec = ptwt.waverec3([c_ours[0].contiguous(), {k: v.contiguous() for k, v in c_ours[1].items()}], pywt.Wavelet('bior1.3'))
    yh = gz['coef1_yh']
    rec_ref = R.idwt3(gz['coef1_yl'].astype(np.float64), {k: yh[:, i].astype(np.float64) for i, k in enumerate(R_BANDS)}, 'bior1.3')
    assert rec.shape == rec_ref.shape == tuple(ex['rec'].shape)
    h_rec, r_rec = rel_l2(rec, ex['rec']), rel_l2(rec_ref, ex['rec'])
    print('   reconstruction: hip vs exact', h_rec, 'reference vs exact', r_rec)
    assert A.gate(h_rec, r_rec), (h_rec, r_rec)
    rec2 = ptwt.waverec3([coef1[0].contiguous(), {k: v.contiguous() for k, v in coef1[1].items()}], pywt.Wavelet('bior1.3'))
    assert rel_l2(rec2, rec_ref) < 1e-6           # same coefficients in: the transform itself


R_BANDS = ('aad', 'ada', 'add', 'daa', 'dad', 'dda', 'ddd')
