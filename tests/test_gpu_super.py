"""Super-resolution variants of the diffusion operators (rows D4-D7 with `low` conditioning, BASELINE configs[4]) against the
reference (tests/golden/make_ref_super_golden.py): smoke space-SR (control) and time-SR models, Burgers SR model."""
import json
import os
import sys

import pytest
import torch

from tests.helpers import GOLDEN, load_npz, rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL, CHAIN_TOL = 1e-5, 1e-5
DDIM3_TOL = 1e-5      # measured 8.0e-6 and 4.7e-6


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


G = load_npz('ref_super_diffusion.npz')
with open(os.path.join(GOLDEN, 'ref_super_manifest.json')) as f:
    META = json.load(f)


def sub(tag):
    pre = tag + '::'
    return {k[len(pre):]: G[k] for k in G.files if k.startswith(pre)}


def check_grad_norms(module, gz):
    params = dict(module.named_parameters())
    n = 0
    for k, v in gz.items():
        if k.startswith('gn::'):
            ref = float(v); got = params[k[4:]].grad.double().norm().item()
            assert abs(got - ref) <= 2e-4 * max(ref, 1e-7), (k, got, ref)
            n += 1
    assert n > 20


@pytest.mark.parametrize('tag', ['smoke_space', 'smoke_time'])
def test_smoke_super_model(trees, tag):
    gz, m = sub(tag), META[tag]
    shape = tuple(m['shape'])
    net = trees['Unet3D'](dim=8, dim_mults=(1, 2), channels=82, resnet_groups=4, init_kernel_size=3)
    kw = dict(loss_layer_weight=torch.from_numpy(gz['lw']), is_condition_control=m['is_condition_control'], is_condition_pad=True, is_wavelet=True,
              is_super_model=True, wave_type='bior1.3', pad_mode='zero', padded_shape=m['padded_shape'], ori_shape=None, image_size=shape[-1],
              frames=shape[1], timesteps=1000, sampling_timesteps=3, loss_type='l2', ddim_sampling_eta=1.)
    dif = trees['GD2'](net, **kw)
    dif.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in gz.items() if k.startswith('w::')}, strict=True)
    dif = dif.to(DEV)
    x0, t, noise = (torch.from_numpy(gz[k]).to(DEV) for k in ('x0', 't', 'noise'))
    loss = dif.p_losses(x0, t, noise=noise)
    assert abs(loss.item() - float(gz['loss'])) < TOL * abs(float(gz['loss']))
    loss.backward()
    check_grad_norms(dif, gz)
    low, init = torch.from_numpy(gz['ddim_low']).to(DEV), torch.from_numpy(gz['ddim_init']).to(DEV)
    control = torch.from_numpy(gz['ddim_control']).to(DEV) if 'ddim_control' in gz else None
    seq = iter([torch.from_numpy(gz[f'ddim_noise_{i}']).to(DEV) for i in range(int(gz['ddim_n_noise']))])
    dif.sample_noise = lambda shp, device: next(seq)
    out = dif.sample(batch_size=shape[0], N_upsample=m['n_up'], init=init, control=control, low=low)
    print(tag, 'ddim3 chain vs reference rel-L2', rel_l2(out, gz['ddim_out']))
    assert rel_l2(out, gz['ddim_out']) < DDIM3_TOL
    if 'ddpm3_out' in gz:
        dif3 = trees['GD2'](dif.model, **{**kw, 'timesteps': 3, 'sampling_timesteps': None}).to(DEV)
        seq3 = iter([torch.from_numpy(gz[f'ddpm3_noise_{i}']).to(DEV) for i in range(int(gz['ddpm3_n_noise']))])
        dif3.sample_noise = lambda shp, device: next(seq3)
        out = dif3.sample(batch_size=shape[0], N_upsample=m['n_up'], init=init, control=control, low=low)
        print(tag, 'ddpm3 chain vs reference rel-L2', rel_l2(out, gz['ddpm3_out']))
        assert rel_l2(out, gz['ddpm3_out']) < CHAIN_TOL
        assert torch.equal(out[:, :, 40:80].cpu(), torch.from_numpy(gz['ddim_low']))     # p_sample_loop re-imposes the low-resolution channels


def test_burgers_super_model_loss_and_grads(trees):
    gz, kw = sub('burgers'), dict(META['burgers'])
    kw['seq_length'] = tuple(kw['seq_length'])
    net = trees['Unet2D'](dim=8, dim_mults=(1, 2, 4), channels=17, resnet_block_groups=1)
    dif = trees['GD1'](net, loss_layer_weight=torch.from_numpy(gz['lw']), **kw)
    dif.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in gz.items() if k.startswith('w::')}, strict=True)
    dif = dif.to(DEV)
    x0, t, noise = (torch.from_numpy(gz[k]).to(DEV) for k in ('x0', 't', 'noise'))
    loss = dif.p_losses(x0, t, noise=noise)
    assert abs(loss.item() - float(gz['loss'])) < TOL * abs(float(gz['loss']))
    loss.backward()
    check_grad_norms(dif, gz)


# ------------------------------------------------------------------------------------------------ round 3: Burgers SR sampling + cascade (row f3)
G3 = load_npz('ref_round3.npz')
with open(os.path.join(GOLDEN, 'ref_round3_manifest.json')) as f:
    META3 = json.load(f)


def sub3(tag):
    pre = tag + '::'
    return {k[len(pre):]: G3[k] for k in G3.files if k.startswith(pre)}


def _noises(gz, tag):
    return iter([torch.from_numpy(gz[f'{tag}_noise_{i}']).to(DEV) for i in range(int(gz[f'{tag}_n_noise']))])


def _burgers_dif(trees, unet, kw, sd, n_ch):
    kw = dict(kw)
    kw['seq_length'] = tuple(kw['seq_length'])
    net = trees['Unet2D'](dim=unet['dim'], dim_mults=tuple(unet['dim_mults']), channels=unet['channels'], resnet_block_groups=unet['resnet_block_groups'])
    dif = trees['GD1'](net, loss_layer_weight=torch.ones(1, n_ch, 1, 1), **kw)
    dif.load_state_dict(sd, strict=any(not k.startswith('model.') for k in sd))       # model-only dicts keep the module's own schedule
    return dif.to(DEV)


def test_burgers_super_sampling_vs_reference(trees):
    """GaussianDiffusion.sample(low=..., N_upsample=...) of the 17-channel Burgers SR model (diffusion_1d.py:462-497 -> :310-460) with
    u0 + uT + f + pad + low conditions: DDIM-4 (eta = 1) and ancestral-3 chains against the reference's outputs on its own noise."""
    gz, m = sub3('bsr'), META3['bsr']
    sd = {k[3:]: torch.from_numpy(v) for k, v in gz.items() if k.startswith('w::')}
    t = lambda k: torch.from_numpy(gz[k]).to(DEV)
    dif = _burgers_dif(trees, m['unet'], m['diffusion'], sd, 17)
    seq = _noises(gz, 'ddim')
    dif.sample_noise = lambda shp, device: next(seq)
    out = dif.sample(batch_size=m['batch'], N_upsample=m['N_upsample'], low=t('low'), u_init=t('u_init'), u_final=t('u_final'), f=t('f'))
    e = rel_l2(out, gz['ddim_out'])
    print('burgers SR ddim4 vs reference', e)
    assert out.shape == gz['ddim_out'].shape and e < CHAIN_TOL
    dif3 = _burgers_dif(trees, m['unet'], {**m['diffusion'], 'timesteps': 3, 'sampling_timesteps': None}, {k: v for k, v in sd.items() if k.startswith('model.')}, 17)
    seq3 = _noises(gz, 'ddpm3')
    dif3.sample_noise = lambda shp, device: next(seq3)
    out3 = dif3.sample(batch_size=m['batch'], N_upsample=m['N_upsample'], low=t('low'), u_init=t('u_init'), u_final=t('u_final'), f=t('f'))
    e3 = rel_l2(out3, gz['ddpm3_out'])
    print('burgers SR ddpm3 vs reference', e3)
    assert e3 < CHAIN_TOL
    # the loops re-impose every condition on the result (diffusion_1d.py:349-371): index work, bit-exact
    ps = m['diffusion']['padded_shape'][m['N_upsample'] - 1]
    ch, cw = ps[0] + 1, ps[1]
    for o in (out, out3):
        assert torch.equal(o[:, 8:16, :ch, :cw].cpu(), torch.from_numpy(gz['low'])[:, :, :ch, :cw])
        assert torch.equal(o[:, 4:8, :ch, :cw].cpu(), torch.from_numpy(gz['f'])[:, :, :ch, :cw])
        assert torch.equal(o[:, -1, :16, :cw].cpu(), torch.from_numpy(gz['u_init'])[:, :, :cw])
        assert torch.equal(o[:, -1, -16:, :cw].cpu(), torch.from_numpy(gz['u_final'])[:, :, :cw])
        assert float(o[:, :-1, ch:].abs().max()) == 0.0 and float(o[:, :, :, cw:].abs().max()) == 0.0


def test_burgers_super_resolution_cascade_vs_reference(trees):
    """eval_ddpm_burgers.py:185-195,305-338 at a reduced size, stage by stage from the reference's own intermediates (index work
    bit-exact), then the whole cascade chained on our side, then the inverse DWT of the result."""
    from ddpm_burgers.wavelet_utils import upsample_coef
    from wave_trans import tensor_to_coef_super
    from pytorch_wavelets import DWTInverse
    gz, m = sub3('bcas'), META3['bcas']
    t = lambda k: torch.from_numpy(gz[k]).to(DEV)
    W0 = {k[4:]: torch.from_numpy(v) for k, v in gz.items() if k.startswith('w0::')}
    W1 = {k[4:]: torch.from_numpy(v) for k, v in gz.items() if k.startswith('w1::')}
    m0 = _burgers_dif(trees, m['unet0'], m['base'], W0, 9)
    m1 = _burgers_dif(trees, m['unet1'], m['sr'], W1, 17)
    k, pad_size, B = m['k'], m['pad_size'], m['batch']
    shape0, shape1, ori1 = m['base']['padded_shape'], m['sr']['padded_shape'][k - 1], m['sr']['ori_shape'][k - 1]
    resc0, resc1 = t('resc0'), t('resc1')

    def lowres(x0):
        coef0 = x0[:, :, :shape0[-2], :shape0[-1]]
        up = upsample_coef(coef0.contiguous(), m1.padded_shape[k - 1])
        padded = torch.nn.functional.pad(up, (0, pad_size - up.shape[-1], 0, pad_size - up.shape[-2]), 'constant', 0)
        return coef0, up, padded, (padded[:, :8] / resc1[:, 8:16]).contiguous()
    # --- stage by stage, each fed with the reference's intermediate
    coef0, up, padded, low = lowres(t('x0'))
    for name, v in (('coef0', coef0), ('up', up), ('padded', padded), ('low', low)):
        assert torch.equal(v.cpu(), torch.from_numpy(gz[name])), name
    Yl, Yh = tensor_to_coef_super(t('x1'), shape1)
    assert torch.equal(Yl.cpu(), torch.from_numpy(gz['Yl'])) and torch.equal(Yh[0].cpu(), torch.from_numpy(gz['Yh']))
    ifm = DWTInverse(mode='periodization', wave='bior2.4')
    u_f = ifm((Yl.contiguous(), [Yh[0].contiguous()]))[:, :, :ori1[-2], :ori1[-1]]
    assert u_f.shape == gz['u_f'].shape and rel_l2(u_f, gz['u_f']) < 1e-6           # same coefficients in: the transform itself
    seq = _noises(gz, 'sr')
    m1.sample_noise = lambda shp, device: next(seq)
    x1 = m1.sample(batch_size=B, N_upsample=k, low=t('low'), u_init=t('u0_s'), u_final=t('uT_s'), f=t('f_s')) * resc1
    e_sr = rel_l2(x1, gz['x1'])
    # --- the whole cascade on our side
    seq0, seq1 = _noises(gz, 'base'), _noises(gz, 'sr')
    m0.sample_noise = lambda shp, device: next(seq0)
    m1.sample_noise = lambda shp, device: next(seq1)
    x0 = m0.sample(batch_size=B, u_init=t('u0_b'), f=t('f_b')) * resc0
    e_base = rel_l2(x0, gz['x0'])
    _, _, _, lo = lowres(x0)
    x1c = m1.sample(batch_size=B, N_upsample=k, low=lo, u_init=t('u0_s'), u_final=t('uT_s'), f=t('f_s')) * resc1
    e_chain = rel_l2(x1c, gz['x1'])
    Ylc, Yhc = tensor_to_coef_super(x1c, shape1)
    u_fc = ifm((Ylc.contiguous(), [Yhc[0].contiguous()]))[:, :, :ori1[-2], :ori1[-1]]
    e_field = rel_l2(u_fc, gz['u_f'])
    print('Burgers SR cascade vs reference: base sample', e_base, 'SR sample from reference low', e_sr, 'end to end', e_chain, 'fields', e_field)
    assert e_base < CHAIN_TOL and e_sr < CHAIN_TOL and e_chain < 2 * CHAIN_TOL and e_field < 2 * CHAIN_TOL


# ------------------------------------------------------------------------------------------------ round 3: Burgers objectives pred_x0 / pred_v
@pytest.mark.parametrize('obj', ['pred_x0', 'pred_v'])
def test_burgers_objectives_vs_reference(trees, obj):
    """objective = 'pred_x0' / 'pred_v' of the Burgers operator: loss weights (diffusion_1d.py:147-156), targets (:596-602) and
    model_predictions (:229-238) -- loss, gradient norms, DDIM-4 (eta = 1) and ancestral-3 chains against the reference."""
    gz, m = sub3('bobj_' + obj), META3['bobj_' + obj]
    kw = dict(m['diffusion']); kw['seq_length'] = tuple(kw['seq_length'])
    u = m['unet']
    mk = lambda: trees['Unet2D'](dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_block_groups=u['resnet_block_groups'])
    lw = torch.from_numpy(gz['lw'])
    dif = trees['GD1'](mk(), loss_layer_weight=lw, **kw)
    assert dif.objective == obj
    sd = {k[3:]: torch.from_numpy(v) for k, v in gz.items() if k.startswith('w::')}
    ref_lw = sd['loss_weight'].double()
    assert rel_l2(dif.loss_weight.double(), ref_lw) < 1e-6            # snr / snr/(snr+1), float32 of float64 schedule arithmetic
    dif.load_state_dict(sd, strict=True)
    dif = dif.to(DEV)
    t = lambda k: torch.from_numpy(gz[k]).to(DEV)
    loss = dif.p_losses(t('x0'), t('t'), noise=t('noise'))
    print(obj, 'loss', loss.item(), float(gz['loss']))
    assert abs(loss.item() - float(gz['loss'])) < TOL * abs(float(gz['loss']))
    loss.backward()
    check_grad_norms(dif, gz)
    seq = _noises(gz, 'ddim')
    dif.sample_noise = lambda shp, device: next(seq)
    out = dif.sample(batch_size=2, u_init=t('u_init'), f=t('f'))
    e = rel_l2(out, gz['ddim_out'])
    print(obj, 'ddim4 vs reference', e)
    assert e < CHAIN_TOL
    dif3 = trees['GD1'](mk(), loss_layer_weight=lw, **{**kw, 'timesteps': 3, 'sampling_timesteps': None})
    dif3.load_state_dict({k: v for k, v in sd.items() if k.startswith('model.')}, strict=False)
    dif3 = dif3.to(DEV)
    seq3 = _noises(gz, 'ddpm3')
    dif3.sample_noise = lambda shp, device: next(seq3)
    out3 = dif3.sample(batch_size=2, u_init=t('u_init'), f=t('f'))
    e3 = rel_l2(out3, gz['ddpm3_out'])
    print(obj, 'ddpm3 vs reference', e3)
    assert e3 < CHAIN_TOL
