"""HIP-graph replay of the unguided sampling step (diffusion_core.sampling_loop / _StepGraph): a loop whose steps are replays of
ONE captured graph must give bit-identical results to the same loop issued launch by launch, for both trees, ancestral and DDIM
sampling, and for the full-width split-fp16 convolution path (whose amax records are re-zeroed inside the graph). GPU box only.

Reference loops: burgers/ddpm_burgers/diffusion_1d.py:310-460, smoke/ddpm/diffusion_2d.py:788-933."""
import sys

import pytest
import torch

from tests.helpers import load_npz, manifest, weights

pytestmark = pytest.mark.gpu
DEV = 'cuda'
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
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion as GD2
    return dict(Unet2D=Unet2D, GD1=GD1, Unet3D=Unet3D_with_Conv3D, GD2=GD2)


def _run(dif, use_graph, seed, **kw):
    g = torch.Generator(device=DEV).manual_seed(seed)
    dif.sample_noise = lambda shape, device: torch.randn(tuple(shape), device=device, generator=g)
    dif.use_graph = use_graph
    out = dif.sample(**kw)
    torch.cuda.synchronize()
    return out


def _smoke(trees, **over):
    gz = load_npz('ref_smoke_diffusion.npz')
    c = M['smoke_diffusion']
    u, d = c['unet'], dict(c['diffusion'])
    d['padded_shape'] = tuple(d['padded_shape']); d['ori_shape'] = tuple(d['ori_shape'])
    d.update(over)
    net = trees['Unet3D'](dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_groups=u['resnet_groups'])
    dif = trees['GD2'](net, loss_layer_weight=torch.from_numpy(gz['lw']), **d)
    dif.load_state_dict({k: v for k, v in weights(gz, 'w::').items() if k.startswith('model.')}, strict=False)
    return gz, dif.to(DEV)


@pytest.mark.parametrize('over', [dict(timesteps=20, sampling_timesteps=None), dict(timesteps=1000, sampling_timesteps=12, ddim_sampling_eta=1.0),
                                  dict(timesteps=1000, sampling_timesteps=9, ddim_sampling_eta=0.0)])
def test_smoke_graph_replay_equals_eager(trees, over):
    gz, dif = _smoke(trees, **over)
    init, control = torch.from_numpy(gz['ddim_init']).to(DEV), torch.from_numpy(gz['ddim_control']).to(DEV)
    a = _run(dif, False, 11, batch_size=2, init=init, control=control)
    b = _run(dif, True, 11, batch_size=2, init=init, control=control)
    assert torch.equal(a, b)
    # second call re-uses the cached graph with new conditions and new noise
    a2 = _run(dif, False, 12, batch_size=2, init=init * 0.5, control=control * 2.0)
    b2 = _run(dif, True, 12, batch_size=2, init=init * 0.5, control=control * 2.0)
    assert torch.equal(a2, b2) and not torch.equal(a, a2)


def test_sampling_loop_graph_equals_eager_for_an_unconditioned_initial_draw(trees):
    """in the smoke order the caller is expected to have imposed the conditions on the initial draw; sampling_loop imposes them
    itself in BOTH forms (on the graph's static buffer / on a copy for the eager launches), so a caller that did not still gets graph == eager,
    and the caller's tensor is not modified. Also the zero-box tag: a launch under a descriptor without a pad box removes an older tag."""
    from wdno_amd import diffusion_core as K, ops
    gz, dif = _smoke(trees, timesteps=1000, sampling_timesteps=10, ddim_sampling_eta=1.0)
    b = 2
    shape = (b, dif.frames, dif.channels, dif.image_size, dif.image_size)
    init, control = torch.from_numpy(gz['ddim_init']).to(DEV), torch.from_numpy(gz['ddim_control']).to(DEV)
    desc = dif._desc(shape, dif.padded_shape)
    src = dif._condition_source(shape, torch.device(DEV), init, control, None)
    x0 = torch.randn(shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))      # conditions NOT imposed
    keep = x0.clone()
    pairs = K.ddim_time_pairs(dif.num_timesteps, 10)
    outs = []
    with torch.no_grad():
        for ug in (False, True):
            g = torch.Generator(device=DEV).manual_seed(77)
            dif.sample_noise = lambda sh, device: torch.randn(tuple(sh), device=device, generator=g)
            outs.append(K.sampling_loop(dif, x0, src, desc, ddim_pairs=pairs, eta=1.0, cond_first=False, use_graph=ug))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(x0, keep)
    # the tag follows the LAST launch's descriptor
    y = K.apply_cond(x0.clone(), src, desc)
    assert ops.zero_box_of(y) is not None
    nopad = dif._desc(shape, dif.padded_shape)
    nopad.cond_pad = 0
    K.apply_cond(y, src, nopad)
    assert ops.zero_box_of(y) is None


@pytest.mark.parametrize('over', [dict(timesteps=16, sampling_timesteps=None), dict(timesteps=1000, sampling_timesteps=10, ddim_sampling_eta=1.0)])
def test_burgers_graph_replay_equals_eager(trees, over):
    gz = load_npz('ref_burgers_diffusion.npz')
    c = M['burgers_diffusion']
    u, d = c['unet'], dict(c['diffusion'])
    d.update(over)
    d['seq_length'] = tuple(d['seq_length'])
    net = trees['Unet2D'](dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_block_groups=u['resnet_block_groups'])
    dif = trees['GD1'](net, loss_layer_weight=torch.from_numpy(gz['lw']), **d)
    dif.load_state_dict({k: v for k, v in weights(gz, 'w::').items() if k.startswith('model.')}, strict=False)
    dif = dif.to(DEV)
    u_init, f = torch.from_numpy(gz['ddim_u_init']).to(DEV), torch.from_numpy(gz['ddim_f']).to(DEV)
    a = _run(dif, False, 21, batch_size=2, u_init=u_init, f=f)
    b = _run(dif, True, 21, batch_size=2, u_init=u_init, f=f)
    assert torch.equal(a, b)


def test_full_width_split_path_graph_replay_equals_eager(trees):
    """dim = 64 on a grid large enough for the split-fp16 convolutions (amax records + split planes inside the graph)."""
    from wdno_amd import ops
    torch.manual_seed(0)
    net = trees['Unet3D'](dim=64, dim_mults=(1, 2, 4), channels=42)
    dif = trees['GD2'](net, torch.ones(1, 1, 42, 1, 1), True, True, True, False, 'bior1.3', 'zero', (5, 12, 12), (8, 20, 20),
                       image_size=16, frames=6, timesteps=1000, sampling_timesteps=10, ddim_sampling_eta=1.0).to(DEV)
    init = torch.randn(1, 6, 16, 16, device=DEV) * 0.3
    control = torch.randn(1, 6, 16, 16, 16, device=DEV) * 0.3
    ops.PROFILE = {}
    a = _run(dif, False, 31, batch_size=1, init=init, control=control)
    used, ops.PROFILE = set(ops.PROFILE), None
    assert any('h3' in k for k in used), used          # the split-fp16 kernels really ran
    b = _run(dif, True, 31, batch_size=1, init=init, control=control)
    assert torch.equal(a, b)
    assert torch.isfinite(a).all()


def test_explicit_guidance_gradient_and_graph_captured_guided_sampling(trees):
    """8f-2: the closed-form design gradient (one IDWT + one adjoint-IDWT launch, no autograd) equals the autograd one, and a guided
    DDIM / ancestral loop whose steps are replays of one captured graph equals the same loop issued launch by launch."""
    from wdno_amd.smoke import guidance as Gd
    torch.manual_seed(0)
    shape, ori = (3, 6, 6), (2, 8, 8)           # bior1.3 / zero: 2 * 3 - 4 = 2 frames, 2 * 6 - 4 = 8 pixels
    resc = (torch.linspace(1.0, 5.0, 42).reshape(1, 1, 42, 1, 1)).to(DEV)
    x = (torch.randn(2, 4, 42, 8, 8) * 0.3).to(DEV)
    init_u = torch.randn(2, 8, 8, device=DEV)
    for cc in (False, True):
        kw = dict(is_condition_control=cc, w_energy=0.7, w_init=1.3)
        a = Gd.guidance_fn(x, shape, ori, resc, init_u=init_u, **kw)
        b = Gd.guidance_fn_explicit(x, shape, ori, resc, init_u=init_u, **kw)
        err = ((a - b).norm() / a.norm()).item()
        assert err < 2e-6, (cc, err)
    gz, dif = _smoke(trees, timesteps=1000, sampling_timesteps=10, ddim_sampling_eta=1.0, is_condition_control=False)
    dif.standard_fixed_ratio = 0.05
    fn = Gd.GuidanceFn(shape, ori, resc, w_energy=0.7, w_init=1.3)
    init = torch.from_numpy(gz['ddim_init']).to(DEV)
    a = _run(dif, False, 41, batch_size=2, design_fn=fn, design_guidance='standard', init=init, init_u=init_u)
    b = _run(dif, True, 41, batch_size=2, design_fn=fn, design_guidance='standard', init=init, init_u=init_u)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    # the same chain through the reference-style autograd callback stays close
    auto = lambda xx, low=None, init=None, init_u=None: Gd.guidance_fn(xx, shape, ori, resc, init_u=init_u, w_energy=0.7, w_init=1.3)
    c = _run(dif, False, 41, batch_size=2, design_fn=auto, design_guidance='standard', init=init, init_u=init_u)
    assert ((a - c).norm() / c.norm()).item() < 1e-4
    gz, dif6 = _smoke(trees, timesteps=12, sampling_timesteps=None, is_condition_control=False)
    a = _run(dif6, False, 42, batch_size=2, design_fn=fn, design_guidance='standard-alpha', init=init, init_u=init_u)
    b = _run(dif6, True, 42, batch_size=2, design_fn=fn, design_guidance='standard-alpha', init=init, init_u=init_u)
    assert torch.equal(a, b)


def test_guided_graph_reads_the_per_call_tensors_of_every_call(trees):
    """inference_2d.py calls sample() once per batch with the SAME design_fn and a NEW init_u (= state[:, 0, 0]): a cached graph must
    replay on the new values (static buffers refreshed before the replays), never on the first call's tensors."""
    from wdno_amd.smoke import guidance as Gd
    from wdno_amd import diffusion_core as K
    torch.manual_seed(1)
    shape, ori = (3, 6, 6), (2, 8, 8)
    resc = (torch.linspace(1.0, 5.0, 42).reshape(1, 1, 42, 1, 1)).to(DEV)
    gz, dif = _smoke(trees, timesteps=1000, sampling_timesteps=10, ddim_sampling_eta=1.0, is_condition_control=False)
    dif.standard_fixed_ratio = 0.05
    fn = Gd.GuidanceFn(shape, ori, resc, w_energy=0.7, w_init=40.0)
    init = torch.from_numpy(gz['ddim_init']).to(DEV)
    u1 = torch.randn(2, 8, 8, device=DEV)
    u2 = torch.randn(2, 8, 8, device=DEV) * 3.0
    g1 = _run(dif, True, 51, batch_size=2, design_fn=fn, design_guidance='standard', init=init, init_u=u1)
    n_graphs = len(K._graph_cache[dif])
    u1.fill_(float('nan'))                       # the first call's tensor is gone (freed / overwritten) by the time of the second call
    g2 = _run(dif, True, 51, batch_size=2, design_fn=fn, design_guidance='standard', init=init * 0.5, init_u=u2)
    assert len(K._graph_cache[dif]) == n_graphs          # cache hit: the same graph replayed
    e2 = _run(dif, False, 51, batch_size=2, design_fn=fn, design_guidance='standard', init=init * 0.5, init_u=u2)
    assert torch.isfinite(g2).all() and torch.equal(g2, e2) and not torch.equal(g1, g2)
    # a python scalar the captured launches bake in is part of the key
    dif.standard_fixed_ratio = 0.02
    g3 = _run(dif, True, 51, batch_size=2, design_fn=fn, design_guidance='standard', init=init * 0.5, init_u=u2)
    e3 = _run(dif, False, 51, batch_size=2, design_fn=fn, design_guidance='standard', init=init * 0.5, init_u=u2)
    assert torch.equal(g3, e3) and not torch.equal(g3, g2)


def test_graph_cache_sees_a_plain_load_state_dict(trees):
    """A load_state_dict (or any in-place weight edit) between two sample() calls does not pass through the trainer's weight epoch:
    the cache key carries the parameters' version counters, so the stale graph (and its packed weight operands) is not replayed."""
    import gc
    import weakref
    gz, dif = _smoke(trees, timesteps=1000, sampling_timesteps=9, ddim_sampling_eta=0.0)
    init, control = torch.from_numpy(gz['ddim_init']).to(DEV), torch.from_numpy(gz['ddim_control']).to(DEV)
    a = _run(dif, True, 61, batch_size=2, init=init, control=control)
    sd = {k: (v * 1.01 if v.is_floating_point() and k.startswith('model.') and 'freqs' not in k else v) for k, v in dif.state_dict().items()}
    dif.load_state_dict(sd)
    b = _run(dif, True, 61, batch_size=2, init=init, control=control)
    e = _run(dif, False, 61, batch_size=2, init=init, control=control)
    assert torch.equal(b, e) and not torch.equal(a, b)
    ref = weakref.ref(dif)                       # the cached graphs do not keep the module (and its activations) alive
    del dif
    gc.collect()
    assert ref() is None


# ------------------------------------------------------------------------------------------------ the TRAINING step as one graph (round 3)
def _train_pair(trees, tree):
    from wdno_amd.trainer import TrainStep, multistep_lr
    def make():
        torch.manual_seed(4)
        if tree == 'smoke':          # full width on a grid large enough for the split-fp16 kernels, planes hand-over, amax records
            net = trees['Unet3D'](dim=64, dim_mults=(1, 2, 4), channels=42)
            dif = trees['GD2'](net, torch.ones(1, 1, 42, 1, 1), True, True, True, False, 'bior1.3', 'zero', (5, 12, 12), (8, 20, 20), image_size=16, frames=6)
        else:
            net = trees['Unet2D'](dim=32, dim_mults=(1, 2, 4), channels=9, resnet_block_groups=1)
            dif = trees['GD1'](net, seq_length=(32, 32), padded_shape=[21, 28], ori_shape=[41, 56], loss_layer_weight=torch.ones(1, 9, 1, 1),
                               is_condition_pad=True, is_condition_u0=True, is_condition_f=True)
        return TrainStep(dif.to(DEV), lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True, ema_update_every=2)
    g = torch.Generator().manual_seed(8)
    x = (torch.randn((2, 6, 42, 16, 16) if tree == 'smoke' else (4, 9, 32, 32), generator=g) * 0.5).to(DEV)
    return make, x


@pytest.mark.parametrize('tree', ['smoke', 'burgers'])
def test_training_step_graph_replay_equals_eager(trees, tree):
    """TrainStep.capture: loss -> backward -> gradient gather replayed from ONE captured HIP graph (random draws, clip + Adam, EMA stay
    outside). Five optimisation steps (two eager warm-up steps inside capture(), then three replays) must leave bit-identical losses,
    gradient norms, weights, Adam moments and EMA weights as five eager steps consuming the same generator."""
    from wdno_amd import ops
    make, x = _train_pair(trees, tree)
    xs = [x * (1.0 + 0.1 * i) for i in range(5)]
    torch.manual_seed(99)
    te = make()
    ops.PROFILE = {}
    eager = [te.step(xi) for xi in xs]
    used, ops.PROFILE = set(ops.PROFILE), None
    if tree == 'smoke':
        assert any('h3' in k for k in used), used
    torch.cuda.synchronize()
    torch.manual_seed(99)
    tg = make()
    # capture() warms up on its example: feed the same two batches the eager run saw first
    tg.step(xs[0])
    tg.capture(xs[1], warmup=1)
    graphed = [None, None] + [tg.step(xi) for xi in xs[2:]]
    torch.cuda.synchronize()
    assert tg._graph is not None
    for i in (2, 3, 4):
        assert torch.equal(eager[i][0], graphed[i][0]) and torch.equal(eager[i][1], graphed[i][1]), i
    assert torch.equal(te.opt.buf.flat_param, tg.opt.buf.flat_param)
    assert torch.equal(te.opt.exp_avg, tg.opt.exp_avg) and torch.equal(te.opt.exp_avg_sq, tg.opt.exp_avg_sq)
    assert torch.equal(te.ema.flat, tg.ema.flat) and te.step_idx == tg.step_idx == 5
    # a batch of another shape falls back to the eager path
    out = tg.step(xs[0][:1])
    assert torch.isfinite(out[0])
