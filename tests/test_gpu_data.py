"""SURVEY 8f rank 1 on the GPU: the offline transforms (field -> coefficient file) and the Burgers packer, whose condition
channel needs the HIP IDWT / DWT. Smoke_wave (pure indexing) is covered on the CPU in test_host.py."""
import sys

import numpy as np
import pytest
import torch

from tests.helpers import load_npz, rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def trees():
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    return True


def test_burgers_packer_matches_reference_golden(trees):
    """get_wavelet_super_preprocess against the reference function (run on the oracle DWT): base, u0-only condition, and both
    super-resolution levels. Packed coefficients are copies (bit-exact); the condition stripes go through IDWT + DWT (1e-5)."""
    import ddpm_burgers.data_burgers_1d as DB
    G = load_npz('ref_data_burgers.npz')
    coef = [torch.from_numpy(G[f'coef{i}']) for i in range(4)]
    db = lambda: {'coef': [c.clone() for c in coef], 'shape': [c.shape[2:] for c in coef], 'ori_shape': torch.Size((81, 120))}
    resc = torch.tensor([10, 3, 3, 1, 21, 5, 5, 1, 10]).view(1, 9, 1, 1).float()
    resc2 = torch.cat((resc[:, :8].repeat(1, 2, 1, 1), resc[:, 8:]), dim=1)
    for tag, kw in (('base', dict(rescaler=resc)), ('base_u0only', dict(rescaler=resc, is_condition_uT=False)),
                    ('super0', dict(rescaler=resc2, is_super_model=True, N_downsample=0)), ('super1', dict(rescaler=resc2, is_super_model=True, N_downsample=1))):
        pre = DB.get_wavelet_super_preprocess(mode='periodization', wave_type='bior2.4', **kw)
        data, shape, ori_shape = pre(db())
        ref = G[f'out_{tag}_data']
        assert data.device.type == 'cpu' and tuple(data.shape) == ref.shape, tag
        assert list(shape) == G[f'out_{tag}_shape'].tolist() and list(ori_shape) == G[f'out_{tag}_ori_shape'].tolist()
        assert np.array_equal(data[:, :-1].numpy(), ref[:, :-1]), tag                       # coefficient channels: pure copies
        err = np.abs(data[:, -1].numpy() - ref[:, -1]).max() / np.abs(ref[:, -1]).max()
        assert err < 1e-5, (tag, err)
    ds = DB.DiffusionDataset(db(), preprocess=DB.get_wavelet_super_preprocess(rescaler=resc, mode='periodization', wave_type='bior2.4'))
    assert len(ds) == 3 and ds[1].shape == (9, 64, 64) and ds.shape == [41, 60] and ds.ori_shape == [81, 120]


def test_burgers_offline_transform_vs_oracle(trees):
    import wave_trans
    from oracle import dwt_ref as R
    rng = np.random.default_rng(5)
    u = rng.standard_normal((2, 2, 81, 120)).astype(np.float32)
    out = wave_trans.transform_dataset(torch.from_numpy(u).to(DEV), 'bior2.4', 'periodization', N_downsample=4)
    assert [tuple(s) for s in out['shape']] == [(4, 41, 60), (4, 21, 30), (4, 11, 15), (4, 6, 8)] and tuple(out['ori_shape']) == (81, 120)
    for i in range(4):
        yl, yh = R.dwt2(u[:, :, ::2 ** i, ::2 ** i].astype(np.float64), 'bior2.4', 'periodization')
        ref = R.burgers_coef_to_tensor(yl, yh)
        assert out['coef'][i].device.type == 'cpu' and np.abs(out['coef'][i].numpy() - ref).max() < 5e-6


def test_smoke_offline_transform_vs_oracle_and_dataset_roundtrip(trees, tmp_path):
    """transform_simulation -> files -> Smoke_wave: the whole preprocessing chain of the smoke task, against the oracle."""
    import wave_trans_2d
    from ddpm.data_2d import Smoke_wave
    from oracle import dwt_ref as R
    rng = np.random.default_rng(6)
    X = rng.standard_normal((5, 32, 64, 64)).astype(np.float32)
    s = rng.random(32).astype(np.float32)
    ft, fs = wave_trans_2d.transform_simulation(torch.from_numpy(X).to(DEV), torch.from_numpy(s).to(DEV), N_downsample=3)
    assert [tuple(v) for v in ft['shape']] == [(18, 34, 34), (10, 34, 34), (6, 34, 34)]
    assert [tuple(v) for v in fs['shape']] == [(18, 34, 34), (18, 18, 18), (18, 10, 10)] and tuple(ft['ori_shape']) == (32, 64, 64)
    for kind, f in (('time', ft), ('space', fs)):
        for i in range(3):
            xs = X[:, ::2 ** i] if kind == 'time' else X[:, :, ::2 ** i, ::2 ** i]
            lll, det = R.dwt3(xs.astype(np.float64), 'bior1.3')
            assert np.abs(f['coef'][i].numpy() - R.smoke_coef_to_tensor(lll, det)).max() < 5e-6
            yl, yh = R.dwt2(xs[:, None, 0].astype(np.float64), 'bior1.3', 'zero')
            assert np.abs(f['init_coef'][i].numpy() - np.concatenate([yl, yh[:, 0]], axis=1)).max() < 5e-6
            ss = s.reshape(1, 1, -1)[:, :, ::2 ** i] if kind == 'time' else s.reshape(1, 1, -1)
            lo, hi = R.dwt1d(ss.astype(np.float64), 'bior1.3', 'zero')
            assert np.abs(f['smokeout'][i].numpy() - np.concatenate([lo, hi], axis=1)[0]).max() < 5e-6
    for kind, f in (('time', ft), ('space', fs)):
        d = tmp_path / 'train' / 'bior1.3_zero' / f'{kind}_downsample'
        d.mkdir(parents=True)
        torch.save(f, str(d / '000003'))
    state, shape, ori_shape, sim_id = Smoke_wave(str(tmp_path), 'bior1.3', 'zero')[3]
    assert state.shape == (24, 42, 40, 40) and shape == [18, 34, 34] and ori_shape == [32, 64, 64] and sim_id == 3
    assert float(state[18:, :40].abs().max()) == 0 and float(state[:, :40, 34:].abs().max()) == 0     # zero padding of the coefficient channels
    st_sr, _, _, _ = Smoke_wave(str(tmp_path), 'bior1.3', 'zero', is_super_model=True, downsample_type='space', N_downsample=0)[3]
    assert st_sr.shape == (24, 82, 40, 40)


def test_smoke_guidance_gradient_vs_oracle_finite_difference(trees):
    """dJ/dx of the control objective (inference_2d.py:30-66) through the HIP IDWT adjoints, checked against
    central differences of the same objective evaluated with the numpy oracle in fp64 (oracle/guidance_ref.py, pinned against a run of the
    reference's guidance_fn by tests/test_oracle_dwt.py; J is quadratic: the difference is exact)."""
    from wdno_amd.smoke import guidance as Gd
    from oracle import guidance_ref as G
    rng = np.random.default_rng(9)
    shape, ori = (18, 34, 34), (32, 64, 64)
    x = np.zeros((1, 24, 42, 40, 40))
    x[:, :18, :, :34, :34] = rng.standard_normal((1, 18, 42, 34, 34)) * 0.3
    x[:, :18, -1] = rng.standard_normal((1, 18, 40, 40)) * 0.3
    resc = np.linspace(1.0, 9.0, 42).reshape(1, 1, 42, 1, 1)
    init_u = rng.standard_normal((1, 64, 64))
    w_e, w_i = 0.7, 1.3
    g = Gd.guidance_fn(torch.from_numpy(x).float().to(DEV), shape, ori, torch.from_numpy(resc).float().to(DEV), w_energy=w_e, w_init=w_i,
                       init_u=torch.from_numpy(init_u).float().to(DEV)).double().cpu().numpy()
    xs = x * resc
    for seed in range(3):
        v = np.random.default_rng(100 + seed).standard_normal(xs.shape)
        fd = G.directional_derivative(xs, v, shape, ori, init_u, w_e, w_i)
        assert abs((g * v).sum() - fd) < 2e-5 * max(1.0, abs(fd)), (seed, (g * v).sum(), fd)
    # conditioned on the control: only the initial-density term remains
    g2 = Gd.guidance_fn(torch.from_numpy(x).float().to(DEV), shape, ori, torch.from_numpy(resc).float().to(DEV), is_condition_control=True,
                        w_energy=w_e, w_init=w_i, init_u=torch.from_numpy(init_u).float().to(DEV))
    assert float(g2[:, :, 8:].abs().max()) == 0 and float(g2[:, :, :8].abs().max()) > 0


@pytest.mark.parametrize('name', ['full', 'full_control', 'small_b2', 'small_no_weights'])
def test_smoke_guidance_gradient_vs_reference_run(trees, name):
    """wdno_amd/smoke/guidance.py against gradients the REFERENCE's own guidance_fn returned (tests/golden/ref_guidance.npz,
    generated by tests/golden/make_ref_guidance_golden.py from smoke/inference_2d.py:30-66 on inputs tests/helpers.guidance_input regenerates):
    the autograd form through the HIP transforms and the closed form the graph-captured guided sampler uses, 1e-5 rel-L2, and exactly zero
    where the reference's gradient is zero (the padding)."""
    from wdno_amd.smoke import guidance as Gd
    from tests.test_oracle_dwt import guidance_case
    x, resc, init_u, g_ref, kw = guidance_case(name)
    shape, ori = kw.pop('shape'), kw.pop('ori_shape')
    xd, rd, ud = x.to(DEV), resc.to(DEV), init_u.to(DEV)
    tc, hc, wc = shape
    for fn in (Gd.guidance_fn, Gd.guidance_fn_explicit):
        g = fn(xd, shape, ori, rd, init_u=ud, **kw)
        assert rel_l2(g, g_ref) < 1e-5, (name, fn.__name__, rel_l2(g, g_ref))
        rest = g.clone()
        rest[:, :tc, :40, :hc, :wc] = 0
        rest[:, :tc, -1] = 0
        assert float(rest.abs().max()) == 0.0, (name, fn.__name__)
    # through the callable the sampler receives (design_fn(x, low=, init=, init_u=), inference_2d.py:81-93)
    dfn = Gd.GuidanceFn(shape, ori, rd, **kw)
    assert rel_l2(dfn(xd, low=None, init=None, init_u=ud), g_ref) < 1e-5


# ------------------------------------------------------------------------------------------------ round 6: packing kernel + resident loaders
def _smoke_files(root, n, seed=0):
    """n synthetic simulations in the offline transform's on-disk format (wave_trans_2d.py:172-185): level 0 only."""
    g = torch.Generator().manual_seed(seed)
    d = root / 'train' / 'bior1.3_zero' / 'time_downsample'
    d.mkdir(parents=True, exist_ok=True)
    for i in range(n):
        f = {'coef': [torch.randn(5, 8, 18, 34, 34, generator=g)], 'init_coef': [torch.randn(5, 4, 34, 34, generator=g)], 'smokeout': [torch.rand(2, 18, generator=g)],
             'shape': [(18, 34, 34)], 'ori_shape': (32, 64, 64)}
        torch.save(f, str(d / f'{i:06d}'))


def test_pack_smoke_kernel_is_bit_identical_to_the_index_formulation(trees):
    """wdno_pack_smoke_state (csrc/pack.hip; f1, data_2d.py:156-221) against the torch formulation of the same packing -- which
    tests/test_host.py pins to the reference's golden -- for 5 and 4 fields, with and without an index list into larger resident stores, the
    [N, 5, 4, nx, nx] form of the initial coefficients, and other pad sizes: torch.equal (IEEE division in both)."""
    from ddpm.data_2d import pack_smoke_batch, pack_smoke_gpu, _RESCALERS
    g = torch.Generator().manual_seed(1)
    for nf, nt, nx, pad_t, pad_x in ((5, 18, 34, 24, 40), (4, 18, 34, 24, 40), (5, 10, 18, 12, 20), (2, 3, 7, 4, 8)):
        n = 6
        coef = torch.randn(n, nf, 8, nt, nx, nx, generator=g)
        init5 = torch.randn(n, 5, 4, nx, nx, generator=g)
        so = torch.rand(n, 2, nt, generator=g)
        r = torch.tensor(_RESCALERS['bior1.3'], dtype=torch.float32)
        r = torch.cat((r[:8 * nf], r[-2:]))
        want = pack_smoke_batch(coef, init5[:, 0].contiguous(), so, r.reshape(1, -1, 1, 1), pad_t, pad_x)          # host tensors: the index formulation
        got = pack_smoke_batch(coef.to(DEV), init5[:, 0].contiguous().to(DEV), so.to(DEV), r.reshape(1, -1, 1, 1).to(DEV), pad_t, pad_x)
        assert got.is_cuda and torch.equal(got.cpu(), want)
        idx = torch.tensor([4, 0, 0, 5], device=DEV)
        got2 = pack_smoke_gpu(coef.to(DEV), init5.to(DEV), so.to(DEV), r, idx, pad_t, pad_x)                      # stores as the files hold them
        assert torch.equal(got2.cpu(), want[[4, 0, 0, 5]])


def test_resident_smoke_loader_equals_the_dataset(trees, tmp_path):
    """ResidentSmokeLoader: every epoch is a permutation of the simulations, every batch equals Smoke_wave's own items (torch.load + host packing
    per sample) bit for bit; files are read exactly once (the second epoch reads none); two ranks see disjoint shards that reshuffle per epoch."""
    from ddpm.data_2d import Smoke_wave
    from wdno_amd.loader import ResidentSmokeLoader
    _smoke_files(tmp_path, 11)
    ds = Smoke_wave(str(tmp_path), 'bior1.3', 'zero')
    ds.n_simu = 11
    loads = []
    raw0 = ds.raw
    ds.raw = lambda i: (loads.append(i), raw0(i))[1]
    ld = ResidentSmokeLoader(ds, 4, DEV, shuffle=True, seed=3, num_workers=0)
    for epoch in range(2):
        ld.sampler.set_epoch(epoch)
        seen = []
        for state, shape, _, ids in ld:
            assert state.is_cuda and shape == [18, 34, 34]
            for j, i in enumerate(ids.tolist()):
                assert torch.equal(state[j].cpu(), ds[i][0])
            seen += ids.tolist()
        assert sorted(seen) == list(range(11))
        assert len(loads) == 11                       # the second epoch is served from HBM
    assert ld.resident_bytes() == 11 * __import__('wdno_amd.loader', fromlist=['x']).SMOKE_SIM_BYTES
    a = ResidentSmokeLoader(ds, 4, DEV, rank=0, world=2, seed=0, num_workers=0)
    b = ResidentSmokeLoader(ds, 4, DEV, rank=1, world=2, seed=0, num_workers=0)
    ia, ib = a.sampler.indices(), b.sampler.indices()
    assert len(ia) == len(ib) == 6 and sorted(set(ia) | set(ib)) == list(range(11))
    a.sampler.set_epoch(1)
    assert a.sampler.indices() != ia
    # the same order as torch's DistributedSampler gives for this seed / epoch
    from torch.utils.data.distributed import DistributedSampler
    s = DistributedSampler(ds, num_replicas=2, rank=0, shuffle=True, seed=0)
    s.set_epoch(1)
    assert list(s) == a.sampler.indices()


def test_resident_tensor_loader_and_trainer_pick_it_up(trees, tmp_path):
    """ResidentTensorLoader (DiffusionDataset.x in HBM, a batch = one gather) and TrainerCore.make_loader choosing the resident pipelines for the
    datasets they serve, the reference's DataLoader for everything else and when Trainer.resident_data is off."""
    import ddpm_burgers.data_burgers_1d as DB
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion
    from ddpm_burgers.train_diffusion import Trainer
    from wdno_amd.loader import ResidentTensorLoader, ResidentSmokeLoader
    from wdno_amd.trainer import TrainerCore
    x = torch.randn(10, 9, 8, 8)

    class DS(torch.utils.data.Dataset):
        def __init__(self):
            self.x = x

        def __len__(self):
            return 10

        def __getitem__(self, i):
            return self.x[i]
    ld = ResidentTensorLoader(DS(), 4, DEV, seed=5)
    got = torch.cat([b.cpu() for b in ld])
    assert got.shape == x.shape and sorted(got.reshape(10, -1).sum(1).tolist()) == sorted(x.reshape(10, -1).sum(1).tolist())
    torch.manual_seed(0)
    net = Unet2D(dim=8, dim_mults=(1, 2), channels=9, resnet_block_groups=1)
    dif = GaussianDiffusion(net, seq_length=(8, 8), padded_shape=[6, 7], ori_shape=[10, 14], loss_layer_weight=torch.ones(1, 9, 1, 1),
                            is_condition_pad=True, is_condition_u0=True, is_condition_f=True)
    prev = TrainerCore.num_workers
    TrainerCore.num_workers = 0
    try:
        tr = Trainer(dif, DS(), rescaler=torch.ones(1), train_batch_size=4, train_num_steps=3, results_folder=str(tmp_path / 'a'))
        assert isinstance(tr.dl.gi_frame.f_locals['dl'], ResidentTensorLoader)
        for _ in range(3):                              # crosses an epoch boundary (10 samples, batches of 4)
            tr.optimisation_step(lambda: next(tr.dl).to(tr.device, non_blocking=True))
        assert next(tr.dl).is_cuda

        class NoRes(Trainer):
            resident_data = False
        tr2 = NoRes(dif, DS(), rescaler=torch.ones(1), train_batch_size=4, train_num_steps=3, results_folder=str(tmp_path / 'b'))
        assert isinstance(tr2.dl.gi_frame.f_locals['dl'], torch.utils.data.DataLoader)
    finally:
        TrainerCore.num_workers = prev
    torch.cuda.synchronize()


def test_pack_smoke_fields_equals_the_transform_then_pack_chain(trees):
    """wdno_pack_smoke_fields: fields + smoke-out curve -> states in two launches (3-D analysis + a packing launch that transforms the two condition
    channels itself) against the chain it replaces -- offline-style transforms (3-D, 2-D of rho(t = 0), 1-D of the curve: csrc/dwt.hip, pinned to
    PyWavelets by tests/test_gpu_ops.py) followed by the packing of data_2d.py:156-221. Coefficient channels are the same launches' bits; the condition
    channels agree to fp32 rounding (another order of the same fused multiply-adds)."""
    from ddpm.data_2d import pack_smoke_batch, pack_smoke_fields, _RESCALERS
    from wdno_amd import wavelets as Wv
    g = torch.Generator().manual_seed(2)
    for nf in (5, 4):
        b = 3
        fields = torch.randn(b, nf, 32, 64, 64, generator=g).to(DEV)
        curve = torch.rand(b, 32, generator=g).to(DEV)
        r = torch.tensor(_RESCALERS['bior1.3'], dtype=torch.float32)
        r = torch.cat((r[:8 * nf], r[-2:])).to(DEV)
        got = pack_smoke_fields(fields, curve, r)
        coef = Wv.dwt_packed(fields.reshape(b * nf, 32, 64, 64), 'bior1.3', 'zero', 3).reshape(b, nf, 8, 18, 34, 34)
        init = Wv.dwt_packed(fields[:, 0, 0].reshape(b, 1, 64, 64).contiguous(), 'bior1.3', 'zero', 2).reshape(b, 4, 34, 34)
        lo, hi = Wv.DWT1DForward(J=1, mode='zero', wave='bior1.3')(curve.unsqueeze(1).contiguous())
        want = pack_smoke_batch(coef, init, torch.cat((lo, hi[0]), dim=1).contiguous(), r.reshape(1, -1, 1, 1))
        assert got.shape == want.shape == (b, 24, 8 * nf + 2, 40, 40)
        assert torch.equal(got[:, :, :8 * nf], want[:, :, :8 * nf])
        for ch in (-2, -1):
            assert float(want[:, :, ch].abs().max()) > 0.05
            assert float((got[:, :, ch] - want[:, :, ch]).abs().max()) < 2e-6 * float(want[:, :, ch].abs().max()) + 1e-7


@pytest.mark.parametrize('wave, shape, pads', [('bior1.3', (3, 5, 32, 64, 64), (24, 40)), ('bior1.3', (2, 4, 32, 64, 64), (24, 40)),
                                               ('db2', (2, 5, 32, 64, 64), (24, 40)), ('haar', (2, 3, 16, 32, 32), (8, 16)),
                                               ('bior1.3', (1, 2, 14, 30, 30), (12, 20)), ('db3', (9, 1, 20, 40, 40), (16, 24))])
def test_transform_store_into_the_state_is_bit_identical_to_the_two_tensor_form(trees, wave, shape, pads):
    """wdno_dwt_fwd_packed + wdno_pack_smoke_fields(coef = NULL): the 3-D analysis stores sub-bands as channels of the padded state, already divided by
    RESCALER, and the packing launch fills the padding and the condition channels -- against the form it replaces (analysis into a coefficient tensor,
    packing launch reads it): the same fused multiply-adds and the same IEEE division, so every bit agrees, also where a row's last float4 straddles
    the box edge (nx = 33, 34), where the box fills the padded row (nx = pad_x) and for one image per sample. The padded tensor is NaN-poisoned first:
    every element is written by exactly one of the two launches."""
    from ddpm.data_2d import pack_smoke_fields
    g = torch.Generator().manual_seed(5)
    b, nf, t0, h0, w0 = shape
    fields = torch.randn(*shape, generator=g).to(DEV)
    curve = torch.rand(b, t0, generator=g).to(DEV)
    r = (torch.rand(8 * nf + 2, generator=g) * 3 + 0.25).to(DEV)
    real_empty = torch.empty
    try:
        torch.empty = lambda *a, **k: real_empty(*a, **k).fill_(float('nan')) if k.get('dtype') == torch.float32 else real_empty(*a, **k)
        got = pack_smoke_fields(fields, curve, r, wave=wave, pad_t=pads[0], pad_x=pads[1])
    finally:
        torch.empty = real_empty
    want = pack_smoke_fields(fields, curve, r, wave=wave, pad_t=pads[0], pad_x=pads[1], direct=False)
    assert not torch.isnan(got).any()
    assert torch.equal(got, want)
    assert float(got[:, :, :8 * nf].abs().max()) > 0.1
