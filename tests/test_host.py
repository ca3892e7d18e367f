"""CPU-side checks: the C-ABI library loads and exports every symbol of include/wdno_hip.h, the drop-in module trees
have the reference's state_dict layout and schedule buffers, host-side helpers match the oracle, and the product path
refuses to run without a GPU (no silent CPU fallback)."""
import json
import math
import os
import re
import sys

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, load_npz, manifest

ROOT = os.path.dirname(GOLDEN.rstrip('/')).rsplit('/tests', 1)[0]
M = manifest()


@pytest.fixture(scope='module')
def trees():
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion as GD1
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D, RelativePositionBias
    from ddpm.diffusion_2d import GaussianDiffusion as GD2
    return dict(Unet2D=Unet2D, GD1=GD1, Unet3D=Unet3D_with_Conv3D, GD2=GD2, RPB=RelativePositionBias)


def test_library_exports_every_declared_symbol():
    import ctypes
    from wdno_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from wdno_amd.build import build_library
        build_library()
    header = open(os.path.join(ROOT, 'include', 'wdno_hip.h')).read()
    declared = set(re.findall(r'\b(wdno_[a-z0-9_]+)\s*\(', header))
    declared -= {'wdno_stream_t'}
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in wdno_hip.h but not exported'
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    l = _lib.load()
    assert l.wdno_version() >= 100 and l.wdno_strerror(-1) == b'invalid argument'


def test_state_dict_layout_matches_reference(trees):
    with torch.device('meta'):
        a = trees['Unet2D'](dim=128, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
        b = trees['Unet3D'](dim=64, dim_mults=(1, 2, 4), channels=42)
    assert {k: list(v.shape) for k, v in a.state_dict().items()} == M['state_dict_unet2d_full']
    assert {k: list(v.shape) for k, v in b.state_dict().items()} == M['state_dict_unet3d_full']
    assert list(a.state_dict().keys()) == list(M['state_dict_unet2d_full'].keys())      # same order too
    assert list(b.state_dict().keys()) == list(M['state_dict_unet3d_full'].keys())
    assert sum(p.numel() for p in a.parameters()) == M['n_params_unet2d_full'] == 140748553
    assert sum(p.numel() for p in b.parameters() if p.requires_grad) == M['n_params_unet3d_full'] == 23837482
    assert a.channels == 9 and a.out_dim == 9 and a.self_condition is False and b.channels == 42


def test_schedule_buffers_match_reference(trees):
    g = load_npz('ref_schedules.npz')
    net2 = trees['Unet2D'](dim=8, dim_mults=(1,), channels=2)
    for sched in ('cosine', 'linear'):
        m = trees['GD1'](net2, seq_length=(8, 8), beta_schedule=sched, ori_shape=[8, 8])
        sd = {k: v for k, v in m.state_dict().items() if not k.startswith('model.')}
        assert list(sd.keys()) == M['diffusion_buffers']
        for k, v in sd.items():
            assert torch.allclose(v, torch.from_numpy(g[f'burgers_{sched}::{k}']), rtol=1e-6, atol=1e-7), (sched, k)
    net3 = trees['Unet3D'](dim=8, dim_mults=(1,), channels=2)
    for sched in ('sigmoid', 'cosine', 'linear'):
        m = trees['GD2'](net3, None, False, False, True, False, 'bior1.3', 'zero', None, None, image_size=8, frames=2, beta_schedule=sched)
        for k, v in m.state_dict().items():
            if not k.startswith('model.'):
                assert torch.allclose(v, torch.from_numpy(g[f'smoke_{sched}::{k}']), rtol=1e-6, atol=1e-7), (sched, k)
    with pytest.raises(ValueError):
        trees['GD2'](net3, None, False, False, True, False, 'bior1.3', 'zero', None, None, image_size=8, frames=2, beta_schedule='nope')
    with pytest.raises(NotImplementedError):
        trees['GD2'](net3, None, False, False, False, False, 'bior1.3', 'zero', None, None, image_size=8, frames=2)


def test_reference_checkpoint_keys_load(trees):
    g = load_npz('ref_smoke_diffusion.npz')
    c = M['smoke_diffusion']
    net = trees['Unet3D'](dim=c['unet']['dim'], dim_mults=tuple(c['unet']['dim_mults']), channels=42, resnet_groups=c['unet']['resnet_groups'])
    d = dict(c['diffusion'])
    dif = trees['GD2'](net, loss_layer_weight=torch.from_numpy(g['lw']), **d)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('w::')}
    res = dif.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys


def test_relative_position_buckets_match_oracle(trees):
    from oracle.unet_ref import relative_position_bucket
    rpb = trees['RPB'](heads=4, max_distance=32)
    for n in (4, 24, 48):
        pos = torch.arange(n)
        assert torch.equal(rpb.bucket_table(n, 'cpu'), relative_position_bucket(pos[None, :] - pos[:, None]))


def test_no_cpu_fallback(trees):
    from wdno_amd import ops
    net = trees['Unet2D'](dim=8, dim_mults=(1, 2), channels=3)
    with pytest.raises(RuntimeError, match='GPU'):
        net(torch.randn(1, 3, 8, 8), torch.tensor([1]))
    with pytest.raises(RuntimeError, match='GPU'):
        ops.silu(torch.randn(4))


def test_packing_helpers_match_oracle(trees):
    import wave_trans
    import wave_trans_2d
    from oracle import dwt_ref as R
    rng = np.random.default_rng(0)
    yl, yh = rng.standard_normal((2, 2, 41, 60)).astype(np.float32), rng.standard_normal((2, 2, 3, 41, 60)).astype(np.float32)
    t = wave_trans.coef_to_tensor(torch.from_numpy(yl), [torch.from_numpy(yh)], pad=True)
    assert np.array_equal(t.numpy(), R.burgers_coef_to_tensor(yl, yh, pad=True))
    a, b = wave_trans.tensor_to_coef(t.reshape(2, 8, 64, 64), (41, 60))
    ra, rb = R.burgers_tensor_to_coef(t.reshape(2, 8, 64, 64).numpy(), (41, 60))
    assert np.array_equal(a.numpy(), ra) and np.array_equal(b[0].numpy(), rb)
    packed = rng.standard_normal((2, 42, 24, 40, 40)).astype(np.float32)
    for up in (None, 'time', 'space'):
        yl3, det = wave_trans_2d.tensor_to_coef(torch.from_numpy(packed), (18, 34, 34), up)
        ryl, rdet = R.smoke_tensor_to_coef(packed, (18, 34, 34), up)
        assert np.array_equal(yl3.numpy(), ryl) and all(np.array_equal(det[k].numpy(), rdet[k]) for k in rdet)
    c = wave_trans_2d.coef_to_tensor([torch.from_numpy(ryl), {k: torch.from_numpy(v) for k, v in rdet.items()}])
    assert np.array_equal(c.numpy(), R.smoke_coef_to_tensor(ryl, rdet))


def test_packing_helpers_match_reference_golden(trees):
    """P1-P3 against outputs of the reference's own functions (tests/golden/make_ref_packing_golden.py), incl. J = 3."""
    import wave_trans
    import wave_trans_2d
    from ddpm_burgers.wavelet_utils import upsample_coef as up_b
    from ddpm.wave_utils import upsample_coef as up_s
    from oracle import dwt_ref as R
    from tests.helpers import load_npz
    P = load_npz('ref_packing.npz')
    T = torch.from_numpy
    assert np.array_equal(wave_trans.coef_to_tensor(T(P['b_j1_yl']), [T(P['b_j1_yh0'])]).numpy(), P['b_j1_out'])
    assert np.array_equal(wave_trans.coef_to_tensor(T(P['b_j1_yl'][..., :40, :]), [T(P['b_j1_yh0'][..., :40, :])], pad=True).numpy(), P['b_j1_out_pad'])
    yh3 = [P[f'b_j3_yh{i}'] for i in range(3)]
    assert np.array_equal(wave_trans.coef_to_tensor(T(P['b_j3_yl']), [T(h) for h in yh3]).numpy(), P['b_j3_out'])
    assert np.array_equal(R.burgers_coef_to_tensor(P['b_j3_yl'], yh3), P['b_j3_out'])                       # pins the oracle too
    for fn, key, shp in ((wave_trans.tensor_to_coef, 'b_t2c', (41, 60)), (wave_trans.tensor_to_coef_super, 'b_t2cs', (40, 60))):
        a, b = fn(T(P['b_t2c_in']), shp)
        assert np.array_equal(a.numpy(), P[f'{key}_yl']) and np.array_equal(b[0].numpy(), P[f'{key}_yh'])
    det = {k: T(P['s_c2t_det'][:, i]) for i, k in enumerate(wave_trans_2d.BANDS)}
    assert np.array_equal(wave_trans_2d.coef_to_tensor([T(P['s_c2t_lll']), det]).numpy(), P['s_c2t_out'])
    for ut in (None, 'time', 'space'):
        a, b = wave_trans_2d.tensor_to_coef(T(P['s_t2c_in']), (6, 7, 7), ut)
        assert np.array_equal(a.numpy(), P[f's_t2c_{ut}_yl'])
        assert np.array_equal(np.stack([b[k].numpy() for k in wave_trans_2d.BANDS], axis=1), P[f's_t2c_{ut}_yh'])
        ra, rb = R.smoke_tensor_to_coef(P['s_t2c_in'], (6, 7, 7), ut)
        assert np.array_equal(ra, P[f's_t2c_{ut}_yl'])
    assert np.array_equal(R.upsample_coef_2d(P['up_b_in']), P['up_b_out'])
    assert np.array_equal(R.upsample_coef_3d(P['up_s_in'], 'time'), P['up_s_time'])
    assert np.array_equal(R.upsample_coef_3d(P['up_s_in'], 'space'), P['up_s_space'])
    assert up_b.__code__.co_varnames[:2] == ('w_sub', 'shape') and up_s.__code__.co_varnames[:3] == ('w_sub', 'shape', 'type')


def test_pywt_shim_and_filters():
    from wdno_amd import tree_path
    sys.path.insert(0, tree_path('third_party'))
    import pywt
    w = pywt.Wavelet('bior2.4')
    assert w.dec_len == 10 and abs(sum(w.dec_lo) - math.sqrt(2)) < 1e-12
    assert pywt.dwt_max_level(80, 'bior2.4') == 3 and pywt.dwt_max_level(32, 'bior1.3') == 2
    assert pywt.dwt_coeff_len(81, 10, 'periodization') == 41 and pywt.dwt_coeff_len(64, 6, 'zero') == 34
    g = load_npz('dwt_pywt.npz')
    assert str(g['dwt2per_ref_wave']) == 'bior2.4'


def test_lr_schedules_match_torch():
    from wdno_amd.trainer import cosine_annealing_lr, multistep_lr
    p = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.SGD(p, lr=1e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10000)
    for step in range(5):
        assert abs(opt.param_groups[0]['lr'] - cosine_annealing_lr(1e-4, step, 10000)) < 1e-15
        opt.step(); sch.step()
    assert multistep_lr(1e-3, 49999) == 1e-3 and abs(multistep_lr(1e-3, 50000) - 1e-4) < 1e-18 and abs(multistep_lr(1e-3, 200000) - 1e-5) < 1e-18


def test_ddim_time_pairs():
    from wdno_amd.diffusion_core import ddim_time_pairs
    from oracle.diffusion_ref import ddim_times
    for T, S in ((1000, 4), (1000, 50), (1000, 100), (1000, 250)):
        assert ddim_time_pairs(T, S) == ddim_times(T, S)
    assert ddim_time_pairs(1000, 4)[0] == (999, 749) and ddim_time_pairs(1000, 4)[-1][1] == -1


def test_trainer_checkpoint_layout_matches_torch_adam():
    """T3 row: the 'opt' entry of a checkpoint has the layout torch.optim.Adam.state_dict() has (indices over *all*
    model.parameters(), state only for trainable ones) and round-trips through adam_state_from_torch."""
    import types
    from wdno_amd import trainer as T
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
    net[0].bias.requires_grad_(False)                       # frozen parameter in the middle (like rotary freqs)
    ref = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.99))
    for _ in range(2):
        net(torch.randn(5, 3)).square().sum().backward()
        ref.step(); ref.zero_grad()
    rsd = ref.state_dict()
    train = [p for p in net.parameters() if p.requires_grad]
    n = sum(p.numel() for p in train)

    class Buf:
        params = train

        def _spans(self):
            off = 0
            for p in train:
                yield off, p.numel()
                off += p.numel()
    opt = types.SimpleNamespace(buf=Buf(), exp_avg=torch.zeros(n), exp_avg_sq=torch.zeros(n), step_count=0, lr=0.0, betas=(0, 0), eps=0.0)
    T.adam_state_from_torch(net, opt, rsd)
    assert opt.step_count == 2 and opt.lr == 1e-4 and opt.betas == (0.9, 0.99) and opt.eps == 1e-8
    out = T.adam_state_to_torch(net, opt)
    assert sorted(out['state'].keys()) == sorted(rsd['state'].keys()) == [0, 2, 3]
    assert out['param_groups'][0]['params'] == rsd['param_groups'][0]['params'] == [0, 1, 2, 3]
    for i in rsd['state']:
        for k in ('exp_avg', 'exp_avg_sq'):
            assert torch.equal(out['state'][i][k], rsd['state'][i][k])
        assert float(out['state'][i]['step']) == float(rsd['state'][i]['step'])
    assert set(rsd['param_groups'][0].keys()) <= set(out['param_groups'][0].keys()) | {'decoupled_weight_decay'}
    fresh = torch.optim.Adam(net.parameters())
    fresh.load_state_dict(out)                                # a reference Trainer can resume from our checkpoint


def test_package_fallthrough_to_reference_modules(tmp_path):
    """Modules / names that are not on the WDNO path resolve to the reference tree when it sits behind ours on sys.path
    (ddpm.utils, ddpm_burgers.result_io, the 2-D `Unet` of ddpm.diffusion_2d, the raw-field `Smoke` dataset ...). A stand-in tree is used here."""
    import subprocess
    from wdno_amd import tree_path
    fake = tmp_path / 'ref_smoke'
    (fake / 'ddpm').mkdir(parents=True)                      # namespace package, like smoke/ddpm in the reference
    (fake / 'ddpm' / 'utils.py').write_text('load_data = "reference utils"\n')
    (fake / 'ddpm' / 'data_2d.py').write_text('Smoke = "reference raw dataset"\nSmoke_wave = "must not win"\n')
    (fake / 'ddpm' / 'diffusion_2d.py').write_text('Unet = "reference 2-D Unet"\nGaussianDiffusion = "must not win"\n')
    fakeb = tmp_path / 'ref_burgers'
    (fakeb / 'ddpm_burgers').mkdir(parents=True)
    (fakeb / 'ddpm_burgers' / '__init__.py').write_text('')
    (fakeb / 'ddpm_burgers' / 'result_io.py').write_text('def merge_save_dict():\n    return "reference io"\n')
    code = (
        'import sys\n'
        f'sys.path[:0] = [{tree_path("third_party")!r}, {tree_path("smoke")!r}, {tree_path("burgers")!r}, {str(fake)!r}, {str(fakeb)!r}]\n'
        'from ddpm.diffusion_2d import Unet, GaussianDiffusion, Trainer\n'
        'from ddpm.data_2d import Smoke, Smoke_wave\n'
        'from ddpm.utils import load_data\n'
        'from ddpm_burgers.result_io import merge_save_dict\n'
        'from ddpm_burgers.unet import Unet2D\n'
        'import ddpm.diffusion_2d as m\n'
        'assert Unet == "reference 2-D Unet" and load_data == "reference utils" and merge_save_dict() == "reference io"\n'
        'assert Smoke == "reference raw dataset" and isinstance(Smoke_wave, type)\n'
        'assert isinstance(GaussianDiffusion, type) and "wdno_amd" in m.__file__ and "wdno_amd" in sys.modules[Unet2D.__module__].__file__\n'
        'try:\n'
        '    m.DoesNotExist\n'
        'except AttributeError as e:\n'
        '    print("OK", type(e).__name__)\n')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and 'OK AttributeError' in out.stdout, out.stderr[-2000:]


def test_trainer_signatures_match_reference(trees):
    """Constructor keywords of both Trainer classes (train_diffusion.py:41-66, diffusion_2d.py:1061-1087)."""
    import inspect
    from ddpm_burgers.train_diffusion import Trainer as TB
    from ddpm.diffusion_2d import Trainer as TS
    kb = list(inspect.signature(TB.__init__).parameters)
    assert kb[:3] == ['self', 'diffusion_model', 'dataset']
    for k in ('is_super_model', 'wave_type', 'pad_mode', 'rescaler', 'exp_name', 'train_batch_size', 'gradient_accumulate_every', 'train_lr',
              'train_num_steps', 'ema_update_every', 'ema_decay', 'adam_betas', 'test_every', 'save_and_sample_every', 'num_samples',
              'results_folder', 'amp', 'mixed_precision_type', 'split_batches', 'max_grad_norm'):
        assert k in kb, k
    ks = list(inspect.signature(TS.__init__).parameters)
    assert ks[:4] == ['self', 'diffusion_model', 'dataset', 'dataset_path']
    for k in ('N_downsample', 'train_batch_size', 'gradient_accumulate_every', 'augment_horizontal_flip', 'train_lr', 'train_num_steps',
              'ema_update_every', 'ema_decay', 'adam_betas', 'save_and_sample_every', 'num_samples', 'results_path', 'amp', 'fp16',
              'split_batches', 'convert_image_to', 'calculate_fid', 'inception_block_idx', 'is_schedule', 'resume', 'resume_step'):
        assert k in ks, k
    for T in (TB, TS):
        for m in ('save', 'load', 'train', 'device'):
            assert hasattr(T, m)


def test_smoke_dataset_matches_reference_golden(trees, tmp_path):
    """ddpm.data_2d.Smoke_wave (file-per-simulation coefficient format -> U-Net state, base and both
    super-resolution variants) against the outputs of the reference class on the same synthetic files."""
    from ddpm.data_2d import Smoke_wave, SuperDataLoader, pack_smoke_state
    from tests.helpers import load_npz
    G = load_npz('ref_data_smoke.npz')
    T = torch.from_numpy
    for kind in ('time', 'space'):
        d = tmp_path / 'train' / 'bior1.3_zero' / f'{kind}_downsample'
        d.mkdir(parents=True)
        torch.save({'coef': [T(G[f'{kind}_coef{i}']) for i in range(2)], 'init_coef': [T(G[f'{kind}_init{i}']) for i in range(2)],
                    'smokeout': [T(G[f'{kind}_smokeout{i}']) for i in range(2)], 'shape': [G[f'{kind}_coef{i}'].shape[-3:] for i in range(2)],
                    'ori_shape': torch.Size(G[f'{kind}_ori_shape'].tolist())}, str(d / '000000'))
    for tag, kw in (('base', dict(is_super_model=False)), ('super_time', dict(is_super_model=True, downsample_type='time', N_downsample=0)),
                    ('super_space', dict(is_super_model=True, downsample_type='space', N_downsample=0))):
        ds = Smoke_wave(str(tmp_path), 'bior1.3', 'zero', **kw)
        state, shape, ori_shape, sim_id = ds[0]
        assert np.array_equal(ds.RESCALER.numpy(), G[f'out_{tag}_rescaler'])
        assert state.shape == G[f'out_{tag}_state'].shape and np.array_equal(state.numpy(), G[f'out_{tag}_state']), tag
        assert list(shape) == G[f'out_{tag}_shape'].tolist() and list(ori_shape) == G[f'out_{tag}_ori_shape'].tolist() and sim_id == 0
    assert len(ds) == 20000 and hasattr(SuperDataLoader, '__iter__')
    # the packer itself is device-agnostic: same result when called directly
    st = pack_smoke_state(T(G['time_coef0']), T(G['time_init0']), T(G['time_smokeout0']), torch.from_numpy(G['out_base_rescaler']))
    assert np.array_equal(st.numpy(), G['out_base_state'])


def test_no_kernel_spills_registers():
    """hipcc's resource report from the last build (wdno_amd/build/kernel_resources.json): a kernel that falls into scratch
    memory still computes the right thing, only several times slower (the linear-attention backward once went from 0.26 to
    1.5 ms that way), so it is treated as a build failure here."""
    import json
    from wdno_amd import build
    build.build_library(verbose=False)
    if not os.path.exists(build.RESOURCES):
        pytest.skip('library was built without the resource report')
    res = json.load(open(build.RESOURCES))
    assert len(res) > 50
    bad = {k: v for k, v in res.items() if v.get('vgpr_spill', 0) or v.get('scratch_bytes', 0)}
    assert not bad, bad


# ----------------------------------------------------------------------------------------------------- round-2 boundary checks
def _signature_rows(obj):
    import inspect
    out = []
    for name, p in inspect.signature(obj).parameters.items():
        dv = p.default
        if dv is inspect.Parameter.empty:
            rep = '<required>'
        elif isinstance(dv, (int, float, str, bool, type(None), tuple, list, dict)):
            rep = repr(dv)
        elif torch.is_tensor(dv):
            rep = 'tensor' + repr(dv.tolist())
        else:
            rep = '<object>'
        out.append([name, str(p.kind), rep])
    return out


def test_signatures_equal_the_reference():
    """Every class / method / function the drop-in trees re-implement has the reference's parameter list: names, order, kinds
    and defaults (tests/golden/ref_signatures.json, written by make_ref_round2_golden.py from the imported reference)."""
    import importlib
    import json
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    with open(os.path.join(GOLDEN, 'ref_signatures.json')) as f:
        ref = json.load(f)
    assert len(ref) >= 60
    bad = []
    for key, want in ref.items():
        parts, obj = key.split('.'), None
        for cut in range(len(parts) - 1, 0, -1):
            try:
                obj = importlib.import_module('.'.join(parts[:cut]))
                for a in parts[cut:]:
                    obj = getattr(obj, a)
                break
            except (ImportError, AttributeError):
                obj = None
        if obj is None:
            bad.append((key, 'missing'))
        elif _signature_rows(obj) != want:
            bad.append((key, _signature_rows(obj), want))
    assert not bad, bad


def test_guidance_combiners_match_reference_outputs():
    from wdno_amd import tree_path
    p = tree_path('burgers')
    if p not in sys.path:
        sys.path.insert(0, p)
    from ddpm_burgers import model_utils as MU
    g = np.load(os.path.join(GOLDEN, 'ref_round2.npz'))
    ep, nj = torch.from_numpy(g['proj::ep']), torch.from_numpy(g['proj::nj'])
    for norm in ('F', '1D_x'):
        assert torch.equal(MU.get_proj_ep_orthogonal_func(norm)(ep, nj), torch.from_numpy(g['proj::out_' + norm])), norm
    assert torch.equal(MU.get_proj_ep_orthogonal_func('1D_t')(ep[0, 0], nj[0, 0]), torch.from_numpy(g['proj::out_1D_t']))
    sched = torch.from_numpy(g['gb::sched_values'])
    assert torch.equal(torch.stack([MU.get_scheduler('cosine')(t) for t in range(1000)]), sched)


def test_oracle_smoke_train_loop_matches_reference_t2():
    """T2 row, CPU side: oracle p_losses + torch Adam(1e-3, (0.9, 0.99)) + clip + MultiStepLR reproduces the reference's 3 steps."""
    import json
    from oracle import diffusion_ref as D, unet_ref as U
    g = np.load(os.path.join(GOLDEN, 'ref_round2.npz'))
    with open(os.path.join(GOLDEN, 'ref_round2_manifest.json')) as f:
        m = json.load(f)['t2']
    sd = {k[len('t2::w0::model.'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('t2::w0::model.')}
    sd = {k: (v.requires_grad_(True) if v.is_floating_point() and not k.endswith('freqs') else v) for k, v in sd.items()}
    params = [v for v in sd.values() if v.requires_grad]
    u = m['unet']
    model = lambda x, t: U.unet3d_forward(sd, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_groups'])
    buf = D.make_buffers('sigmoid', 1000)
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.99))
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[50000, 150000, 300000], gamma=0.1)
    lw = torch.from_numpy(g['t2::lw'])
    for step in range(3):
        x0, t, noise = (torch.from_numpy(g[f't2::s{step}_{k}']) for k in ('x0', 't', 'noise'))
        loss = D.smoke_p_losses(model, buf, x0, t, noise, padded_shape=m['diffusion']['padded_shape'], loss_layer_weight=lw)
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step(); sch.step()
        assert abs(loss.item() - float(g[f't2::s{step}_loss'])) < 1e-5 * abs(float(g[f't2::s{step}_loss']))
        assert abs(gn.item() - float(g[f't2::s{step}_gnorm'])) < 1e-4 * float(g[f't2::s{step}_gnorm'])


def test_planes_hand_over_predicate():
    """ops.conv_reads_planes: the host-side contract behind out_planes=... (a norm may leave its fp32 output unwritten only when
    the convolution behind it is certain to read fp16 planes)."""
    from wdno_amd import ops
    prev = ops.CONV_MATH, ops.PLANES_FWD
    try:
        ops.CONV_MATH, ops.PLANES_FWD = 'f16x3', True
        w3 = torch.empty(64, 64, 3, 3, 3)
        assert ops.conv_reads_planes(307200, w3)                                   # a level-0 ResnetBlock convolution
        assert not ops.conv_reads_planes(ops.H3_MIN_PIXELS - 1, w3)                # too few pixels: exact-fp32 kernel, reads fp32
        assert not ops.conv_reads_planes(307200, torch.empty(8, 4, 1, 1, 1))       # reduction below the split threshold
        assert not ops.conv_reads_planes(ops.LINEAR_ROWS_MAX, torch.empty(256, 64))   # few-rows linear kernel reads fp32
        assert ops.conv_reads_planes(307200, torch.empty(384, 64))                 # to_qkv on the whole level-0 tensor
        assert not ops.conv_reads_planes(307200, torch.empty(64, 42, 7, 7, 7))     # 42 -> 44 channels: rows are not 16-byte multiples of fp16
        ops.CONV_MATH = 'f32'
        assert not ops.conv_reads_planes(307200, w3)
        ops.CONV_MATH, ops.PLANES_FWD = 'bf16', True
        assert ops.conv_reads_planes(307200, w3)                                   # single bf16 plane: the same hand-over
        ops.PLANES_FWD = False
        assert not ops.conv_reads_planes(307200, w3)
    finally:
        ops.CONV_MATH, ops.PLANES_FWD = prev


def test_ddim_coefficients_follow_the_loaded_schedule(trees):
    """The scalar DDIM coefficients are evaluated from the module's CURRENT alphas_cumprod buffer: loading a checkpoint whose schedule
    buffers differ (even in the last bit -- computed on another host) must not leave a construction-time copy behind
    (wdno_amd.diffusion_core.ac_host; smoke/ddpm/diffusion_2d.py:905-909 reads self.alphas_cumprod at sampling time)."""
    from wdno_amd import diffusion_core as K
    net3 = trees['Unet3D'](dim=8, dim_mults=(1,), channels=2)
    m = trees['GD2'](net3, None, False, False, True, False, 'bior1.3', 'zero', None, None, image_size=8, frames=2)
    a0 = m._ac_host.clone()
    assert torch.equal(a0, m.alphas_cumprod)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    for _ in range(2):                            # two ulps up at one entry
        sd['alphas_cumprod'][749] = torch.nextafter(sd['alphas_cumprod'][749], torch.tensor(1.0))
    m.load_state_dict(sd)
    assert torch.equal(m._ac_host, sd['alphas_cumprod']) and not torch.equal(m._ac_host, a0)
    s0, c0, _ = K.ddim_coefficients(a0, 999, 749, 1.0)
    s1, c1, _ = K.ddim_coefficients(m._ac_host, 999, 749, 1.0)
    assert abs(c1 - c0) > 1e-3 * abs(c0)         # c = sqrt(1 - a' - sigma^2) cancels: c^2 ~ 1e-6 from terms ~ 1, so ulps of the schedule are visible in c
    net2 = trees['Unet2D'](dim=8, dim_mults=(1,), channels=2)
    b = trees['GD1'](net2, seq_length=(8, 8), ori_shape=[8, 8])
    sdb = {k: v.clone() for k, v in b.state_dict().items()}
    sdb['alphas_cumprod'] = sdb['alphas_cumprod'] * 0.5
    b.load_state_dict(sdb)
    assert torch.equal(b._ac_host, sdb['alphas_cumprod'])


def test_ddim_coefficients_are_host_independent_and_match_the_reference_where_it_matters():
    """K.ddim_coefficients does every fp32 operation correctly rounded (double arithmetic + rounding): equal, bit for bit, to the
    oracle's numpy-float32 evaluation for every step pair of the three schedules, and equal to what the reference computed in the
    build container at the one place where an ulp matters (sigmoid schedule, eta = 1, t = 999 -> 749: c^2 = 1.43e-6 left from
    terms ~1; torch's CPU sqrt on the GPU box's host rounds sigma the other way and gets c = 1.2449e-3: tools/ddim_coef_probe.py)."""
    from oracle import diffusion_ref as D
    from wdno_amd import diffusion_core as K
    same = lambda x, y: x == y or (x != x and y != y)
    n = 0
    for sched in (K.sigmoid_beta_schedule, K.cosine_beta_schedule, K.linear_beta_schedule):
        ac = torch.cumprod(1 - sched(1000), 0).float()
        for steps in (3, 4, 10, 100):
            for eta in (0.0, 0.3, 1.0):
                for t, tn in K.ddim_time_pairs(1000, steps):
                    if tn >= 0:
                        a, b = K.ddim_coefficients(ac, t, tn, eta), D.ddim_scalars({'alphas_cumprod': ac}, t, tn, eta)
                        assert all(same(x, y) for x, y in zip(a, b)), (sched.__name__, steps, eta, t, tn, a, b)
                        n += 1
    assert n > 1000
    ac = torch.cumprod(1 - K.sigmoid_beta_schedule(1000), 0).float()
    assert K.ddim_coefficients(ac, 999, 749, 1.0) == (0.9224164485931396, 0.0011960399569943547, 0.3861948251724243)
    ac = torch.cumprod(1 - K.cosine_beta_schedule(1000), 0).float()
    assert math.isnan(K.ddim_coefficients(ac, 999, 665, 1.0)[1])          # Burgers DDIM-3 at eta = 1: NaN in the reference too


def test_batched_smoke_packer_equals_the_per_sample_packer(trees):
    """pack_smoke_batch (the GPU-side packer used by the bench's fields -> DWT -> pack -> step pipeline) is index work: identical to
    stacking pack_smoke_state (the reference's per-item packing, data_2d.py:156-221) over the batch."""
    from ddpm.data_2d import pack_smoke_batch, pack_smoke_state
    g = torch.Generator().manual_seed(3)
    coef = torch.randn(3, 5, 8, 18, 34, 34, generator=g)
    init_coef = torch.randn(3, 5, 4, 34, 34, generator=g)
    smokeout = torch.randn(3, 2, 18, generator=g)
    resc = torch.rand(1, 42, 1, 1, generator=g) + 0.5
    one = torch.stack([pack_smoke_state(coef[i], init_coef[i], smokeout[i], resc) for i in range(3)])
    many = pack_smoke_batch(coef, init_coef[:, 0], smokeout, resc)
    assert many.shape == (3, 24, 42, 40, 40) and torch.equal(one, many)
