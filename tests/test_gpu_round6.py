"""Round 6: the window weight-gradient kernel's split-pair items, the deferred multi-tensor split reduction, and the drop-in surfaces that came with
them. GPU box only.

Reference lines: every nn.Conv3d / Conv2d of the U-Nets (smoke/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:189-230,
burgers/ddpm_burgers/unet.py:129-181) -- autograd's weight gradient of them is what these kernels compute."""
import ctypes as C
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


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


WG_CASES = [('3d 64->64', (2, 24, 40, 40, 64), (64, 64, 3, 3, 3)), ('3d 128->64', (1, 24, 40, 40, 128), (64, 128, 3, 3, 3)),
            ('2d 128->128', (16, 1, 64, 64, 128), (128, 128, 1, 3, 3)), ('3d 64->64 odd split count', (3, 6, 40, 40, 64), (64, 64, 3, 3, 3))]


@pytest.mark.parametrize('name,xs,ws', WG_CASES, ids=[c[0] for c in WG_CASES])
@pytest.mark.parametrize('math', ['f16x3', 'bf16'])
def test_split_pair_plan_equals_the_round5_plan(name, xs, ws, math):
    """The same weight gradient under the split-pair item plan (default) and under round 5's (library debug mode 70: the odd tap row paired with an
    empty window): other split boundaries, so fp32 sums in another order -- equal to 1e-5 relative (measured ~1e-6), each plan bit-reproducible;
    and, through the two-step API (partial sums, then wdno_wgrad_reduce_multi), bit-identical to the one-call API."""
    from wdno_amd import ops, _lib
    lib = ops._lib_()
    prev = ops.CONV_MATH
    ops.CONV_MATH = math
    try:
        g = torch.Generator(device=DEV).manual_seed(3)
        x = torch.randn(*xs, device=DEV, generator=g)
        k, c, ks = ws[0], ws[1], tuple(ws[2:])
        dy = torch.randn(*xs[:4], k, device=DEV, generator=g)
        xpl, ypl = ops.split_f16(x.reshape(-1, c)), ops.split_f16(dy.reshape(-1, k))
        pd = tuple(v // 2 for v in ks)
        f = lambda: ops.conv_wgrad_h3(xpl, tuple(xs[:4]), ypl, tuple(xs[1:4]), ks, (1, 1, 1), pd, param_kc=(k, c))
        res = {}
        for mode in (0, 70):
            lib.wdno_set_debug(mode)
            try:
                res[mode] = f().clone()
                assert torch.equal(res[mode], f())
            finally:
                lib.wdno_set_debug(0)
        rel = ((res[0] - res[70]).norm() / res[70].norm()).item()
        print(name, math, 'split-pair vs round-5 plan', rel)
        assert 0 < rel < 1e-5                               # the plans really differ (another split count), the sums agree
        dst = torch.full_like(res[0], float('nan'))
        with ops.flat_wgrad_scope():                        # deferred: partial sums now, the ordered reduction at the end of the scope
            dw = ops.conv_wgrad_h3(xpl, tuple(xs[:4]), ypl, tuple(xs[1:4]), ks, (1, 1, 1), pd, param_kc=(k, c), out=dst)
            assert len(ops._WGRAD_PENDING) == 1 and dw is dst
            torch.cuda.synchronize()
            assert torch.isnan(dst).all()                   # nothing has been reduced yet
        assert ops._WGRAD_PENDING is None and torch.equal(dst, res[0]) and ops.take_deferred_dsts() == {dst.data_ptr()}
    finally:
        ops.CONV_MATH = prev


def test_deferred_weight_gradient_reductions_leave_a_training_step_bit_identical(trees):
    """ops.DEFER_WGRAD_REDUCE: inside a trainer's backward the split reductions of all weight gradients run as ONE launch when the backward has
    returned (wdno_wgrad_reduce_multi, items by value in the kernel arguments) instead of one launch behind every weight-gradient kernel. Same
    additions in the same order: two optimisation steps of the full-width smoke model end in the same bits, eager and graph-replayed.
    ops.DEFER_ROW_SUMS likewise for the row sums that end the backward (bias gradients from column-sum partials, GroupNorm d(gamma) / d(beta):
    wdno_rows_sum_multi) -- third arm: weight gradients deferred, row sums not."""
    from wdno_amd import ops
    from wdno_amd.trainer import TrainStep
    out = {}
    for defer in (True, False, 'wgrad only'):
        ops.DEFER_WGRAD_REDUCE = bool(defer)
        ops.DEFER_ROW_SUMS = defer is True
        queued = []
        real_flush = ops.flush_wgrad_reduces

        def counting_flush():
            queued.append((len(ops._WGRAD_PENDING or ()), len(ops._ROWSUM_PENDING)))
            real_flush()
        ops.flush_wgrad_reduces = counting_flush
        try:
            torch.manual_seed(0)
            net = trees['Unet3D'](dim=64, dim_mults=(1, 2, 4), channels=42)
            dif = trees['GD2'](net, torch.ones(1, 1, 42, 1, 1), True, True, True, False, 'bior1.3', 'zero', (9, 17, 17), (16, 32, 32), image_size=20, frames=12).to(DEV)
            ts = TrainStep(dif, lr=1e-3, use_ema=False)
            x = (torch.randn(2, 12, 42, 20, 20, generator=torch.Generator().manual_seed(5)) * 0.5).to(DEV)
            torch.manual_seed(9)
            launches = {}
            ops.PROFILE = launches
            try:
                l0, _ = ts.step(x)
            finally:
                ops.PROFILE = None
            assert any('wgrad_h3' in k for k in launches)
            ts.capture(x, warmup=1)
            l2, _ = ts.step(x)
            torch.cuda.synchronize()
            out[defer] = (float(l0), float(l2), ts.opt.buf.flat_param.clone())
            del ts
            if defer is True:                # the backward really queued both kinds (GroupNorm tails, bias sums of split dy sweeps)
                assert max(q[0] for q in queued) > 10 and max(q[1] for q in queued) > 10, queued
            elif defer == 'wgrad only':
                assert max(q[1] for q in queued) == 0, queued
        finally:
            ops.DEFER_WGRAD_REDUCE = True
            ops.DEFER_ROW_SUMS = True
            ops.flush_wgrad_reduces = real_flush
    for arm in (False, 'wgrad only'):
        assert out[True][0] == out[arm][0] and out[True][1] == out[arm][1] and torch.equal(out[True][2], out[arm][2]), arm


def test_reduce_multi_takes_more_items_than_one_launch_holds():
    """wdno_wgrad_reduce_multi chunks its host list into launches of WDNO_WGRAD_REDUCE_MAX (40) items: 90 small items, every destination right."""
    from wdno_amd import ops, _lib
    lib = ops._lib_()
    n_items, splits, K, Cc = 90, 3, 8, 8
    n = 1 * K * 3 * Cc                                  # ntap = 1, kw = 3
    ws = torch.randn(n_items, splits, n, device=DEV)
    dw = torch.zeros(n_items, K, Cc, 1, 1, 3, device=DEV)
    items = (_lib.WgradReduceItem * n_items)()
    for i in range(n_items):
        items[i] = _lib.WgradReduceItem(ws[i].data_ptr(), dw[i].data_ptr(), n, splits, K, Cc, 3, 1, K, Cc, i % 2, 0)
    _lib.check(lib.wdno_wgrad_reduce_multi(C.cast(items, C.c_void_p), n_items, ops._stream()), 'wgrad_reduce_multi')
    torch.cuda.synchronize()
    want = ws.sum(1).reshape(n_items, 1, K, 3, Cc).permute(0, 2, 4, 1, 3).reshape(n_items, K, Cc, 1, 1, 3)      # [tap][k][dx][c] -> [k][c][tap][dx]
    bad = [i for i in range(n_items) if not torch.allclose(dw[i], want[i], rtol=1e-6, atol=1e-6)]
    assert not bad, bad


@pytest.mark.parametrize('xs', [(1, 24, 40, 40), (2, 6, 40, 40), (2, 5, 9, 7)], ids=['batch1', '19200px', 'tiny'])
def test_stem_forward_on_192_pixel_tiles_equals_the_256_pixel_tiles(xs):
    """The 7 x 7 x 7 stem (42 -> 64 channels, video_diffusion_pytorch_conv3d.py:405-407 init_conv) when the launch is less than one round of
    256-pixel tiles: 192-pixel tiles (csrc/conv_h3t.hip, the batch-1 sampling step) -- every output element is the same chain of products in
    the same order, so the bits equal those of the 256-pixel tiles (library debug mode 72)."""
    from wdno_amd import ops
    lib = ops._lib_()
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(*xs, 44, device=DEV, generator=g)
    x[..., 42:] = 0
    w = torch.randn(64, 42, 7, 7, 7, device=DEV, generator=g) * 0.05
    b = torch.randn(64, device=DEV, generator=g)
    res = {}
    with torch.no_grad():
        for mode in (0, 72):
            lib.wdno_set_debug(mode)
            try:
                ops.PROFILE = {}
                res[mode] = ops.conv_cl(x, w, b, stride=1, padding=3).clone()
                names, ops.PROFILE = list(ops.PROFILE), None
            finally:
                lib.wdno_set_debug(0)
            if mode == 0 and xs[1] == 24:                    # (fewer than 100 tiles: the chunked kernel, both modes)
                assert names == ['conv_fwd_h3t_kernel<192,64>'], names
    assert torch.equal(res[0], res[72])
    ref = torch.nn.functional.conv3d(x[..., :42].permute(0, 4, 1, 2, 3).double(), w.double(), b.double(), padding=3).permute(0, 2, 3, 4, 1).float()
    assert float((res[0] - ref).abs().max()) < 2e-5 * float(ref.abs().max())


def test_rows_sum_multi_windows_dtypes_and_more_items_than_one_launch_holds():
    """wdno_rows_sum_multi: out[j] = sum_r part[r * stride + col0 + j] for a host list of items -- float and double partial matrices, column windows
    of wider matrices (GroupNorm's [N][2 C] pieces: d(gamma) = columns 0 .. C-1, d(beta) = C .. 2C-1), 1 .. 2048 rows, 150 items = three launches of
    WDNO_ROWS_SUM_MAX (64); against torch's fp64 sum to fp32 rounding, and bit-identical to the single-tensor launch it replaces (wdno_colsum's finish
    = partial_rows_sum over the same rows)."""
    from wdno_amd import ops, _lib
    lib = ops._lib_()
    g = torch.Generator(device=DEV).manual_seed(4)
    items, wants, outs, keep = [], [], [], []
    for i in range(150):
        rows = [1, 7, 32, 33, 257, 2048][i % 6]
        ncols = [5, 32, 64, 100, 256][i % 5]
        stride = ncols * (2 if i % 2 else 1) + (3 if i % 3 == 0 else 0)
        col0 = (stride - ncols) if i % 2 else 0
        dbl = i % 4 < 2
        part = torch.randn(rows, stride, device=DEV, generator=g, dtype=torch.float64 if dbl else torch.float32)
        out = torch.full((ncols,), float('nan'), device=DEV)
        keep.append(part)
        outs.append(out)
        wants.append(part[:, col0:col0 + ncols].double().sum(0).float())
        items.append(_lib.RowsSumItem(part.data_ptr(), out.data_ptr(), rows, stride, col0, ncols, int(dbl), 0))
    arr = (_lib.RowsSumItem * len(items))(*items)
    _lib.check(lib.wdno_rows_sum_multi(C.cast(arr, C.c_void_p), len(items), ops._stream()), 'rows_sum_multi')
    torch.cuda.synchronize()
    bad = [i for i, (o, w) in enumerate(zip(outs, wants)) if not torch.allclose(o, w, rtol=2e-6, atol=1e-5)]
    assert not bad, bad
    # the job form used by the backward passes, run at once (no trainer scope) == the queued form above, bit for bit
    job = ops._RowsJob(keep[5], keep[5].data_ptr(), 2048, items[5].stride, items[5].col0, items[5].ncols, items[5].is_double)
    assert torch.equal(ops.rows_sum_for_param(job, None), outs[5])
