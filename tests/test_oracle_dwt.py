"""Pins oracle/dwt_ref.py against PyWavelets 1.1.1 outputs (tests/golden/dwt_pywt.npz)."""
import numpy as np
import pytest

from oracle import dwt_ref as R
from tests.helpers import load_npz

G = load_npz('dwt_pywt.npz')
TOL = 1e-12


def _err(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max()


@pytest.mark.parametrize('tag', ['ref', 'small', 'db4', 'sym4', 'haar', 'b13'])
def test_dwt2_periodization(tag):
    w = str(G[f'dwt2per_{tag}_wave'])
    yl, yh = R.dwt2(G[f'dwt2per_{tag}_x'], w, 'periodization')
    assert _err(yl, G[f'dwt2per_{tag}_yl']) < TOL
    assert _err(yh, G[f'dwt2per_{tag}_yh']) < TOL
    rec = R.idwt2(G[f'idwt2per_{tag}_yl'], G[f'idwt2per_{tag}_yh'], w, 'periodization')
    assert _err(rec, G[f'idwt2per_{tag}_x']) < TOL


def test_dwt2_reference_sizes():
    yl, yh = R.dwt2(G['dwt2per_ref_x'], 'bior2.4', 'periodization')
    assert yl.shape == (1, 2, 41, 60) and yh.shape == (1, 2, 3, 41, 60)       # 81x120 -> 41x60
    assert R.idwt2(yl, yh, 'bior2.4', 'periodization').shape == (1, 2, 82, 120)  # callers crop to 81
    t = R.burgers_coef_to_tensor(yl, yh, pad=True)
    assert t.shape == (1, 2, 4, 64, 64) and np.all(t[..., 41:, :] == 0) and np.all(t[..., 60:] == 0)


@pytest.mark.parametrize('tag', ['per', 'perodd', 'zero', 'zeroodd', 'zdb4'])
def test_dwt1(tag):
    w, m = str(G[f'dwt1_{tag}_wave']), str(G[f'dwt1_{tag}_mode'])
    lo, hi = R.dwt1d(G[f'dwt1_{tag}_x'], w, m)
    assert _err(lo, G[f'dwt1_{tag}_lo']) < TOL and _err(hi, G[f'dwt1_{tag}_hi']) < TOL
    assert _err(R.idwt1d(G[f'idwt1_{tag}_lo'], G[f'idwt1_{tag}_hi'], w, m), G[f'idwt1_{tag}_x']) < TOL


@pytest.mark.parametrize('tag', ['ref', 'odd'])
def test_dwt2_zero(tag):
    yl, yh = R.dwt2(G[f'dwt2zero_{tag}_x'], str(G[f'dwt2zero_{tag}_wave']), 'zero')
    assert _err(yl, G[f'dwt2zero_{tag}_yl']) < TOL and _err(yh, G[f'dwt2zero_{tag}_yh']) < TOL


@pytest.mark.parametrize('tag', ['mid', 'odd', 'db2'])
def test_dwt3_zero(tag):
    w = str(G[f'dwt3_{tag}_wave'])
    lll, det = R.dwt3(G[f'dwt3_{tag}_x'], w)
    assert _err(R.smoke_coef_to_tensor(lll, det), G[f'dwt3_{tag}_coef']) < TOL
    c = G[f'idwt3_{tag}_coef']
    rec = R.idwt3(c[:, 0], {k: c[:, i + 1] for i, k in enumerate(R.BANDS3[1:])}, w)
    assert _err(rec, G[f'idwt3_{tag}_x']) < TOL


def test_perfect_reconstruction_and_packing_roundtrip():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 32, 64, 64))
    lll, det = R.dwt3(x, 'bior1.3')
    assert lll.shape == (3, 18, 34, 34)
    assert np.abs(R.idwt3(lll, det, 'bior1.3') - x).max() < 1e-12
    t = R.smoke_coef_to_tensor(lll, det)                      # [3, 8, 18, 34, 34]
    packed = np.zeros((1, 24, 24, 40, 40))
    packed[0, :, :18, :34, :34] = t.reshape(24, 18, 34, 34)
    yl, yh = R.smoke_tensor_to_coef(packed, (18, 34, 34), nfields=3)
    assert np.array_equal(yl, lll) and all(np.array_equal(yh[k], det[k]) for k in det)
    x2 = rng.standard_normal((2, 2, 80, 120))
    yl, yh = R.dwt2(x2, 'bior2.4', 'periodization')
    assert np.abs(R.idwt2(yl, yh, 'bior2.4', 'periodization') - x2).max() < 1e-12
    t = R.burgers_coef_to_tensor(yl, yh, pad=True).reshape(2, 8, 64, 64)
    yl2, yh2 = R.burgers_tensor_to_coef(t, (40, 60))
    assert np.array_equal(yl2, yl) and np.array_equal(yh2, yh)


def test_upsample_coef():
    a = np.arange(2 * 3 * 2 * 2, dtype=np.float64).reshape(2, 3, 2, 2)
    u = R.upsample_coef_2d(a)
    assert u.shape == (2, 3, 4, 4) and u[0, 0, 1, 1] == a[0, 0, 0, 0] and u[1, 2, 3, 2] == a[1, 2, 1, 1]
    b = np.arange(1 * 2 * 3 * 2 * 2, dtype=np.float64).reshape(1, 2, 3, 2, 2)
    assert R.upsample_coef_3d(b, 'time').shape == (1, 4, 3, 2, 2)
    assert R.upsample_coef_3d(b, 'space').shape == (1, 2, 3, 4, 4)
    assert np.array_equal(R.upsample_coef_3d(b, 'time')[0, 1], b[0, 0])


# ----------------------------------------------------------------------------- multi-level (J > 1)
@pytest.mark.parametrize('tag', ['ref3', 'even2', 'db4odd'])
def test_wavedec2_multilevel(tag):
    w, J = str(G[f'wavedec2_{tag}_wave']), int(G[f'wavedec2_{tag}_J'])
    yl, yh = R.wavedec2(G[f'wavedec2_{tag}_x'], w, 'periodization', J)
    assert _err(yl, G[f'wavedec2_{tag}_yl']) < TOL
    for lvl in range(J):
        assert _err(yh[lvl], G[f'wavedec2_{tag}_yh{lvl}']) < TOL
    rec = R.waverec2(G[f'waverec2_{tag}_yl'], [G[f'waverec2_{tag}_yh{lvl}'] for lvl in range(J)], w, 'periodization')
    assert _err(rec, G[f'waverec2_{tag}_x']) < TOL


@pytest.mark.parametrize('tag', ['per3', 'zero2'])
def test_wavedec1_multilevel(tag):
    w, m, J = str(G[f'wavedec1_{tag}_wave']), str(G[f'wavedec1_{tag}_mode']), int(G[f'wavedec1_{tag}_J'])
    lo, his = R.wavedec1(G[f'wavedec1_{tag}_x'], w, m, J)
    assert _err(lo, G[f'wavedec1_{tag}_lo']) < TOL
    for lvl in range(J):
        assert _err(his[lvl], G[f'wavedec1_{tag}_hi{lvl}']) < TOL
    assert _err(R.waverec1(lo, his, w, m), G[f'waverec1_{tag}_x']) < TOL


@pytest.mark.parametrize('tag', ['l2', 'l2odd'])
def test_wavedec3_multilevel(tag):
    w, J = str(G[f'wavedec3_{tag}_wave']), int(G[f'wavedec3_{tag}_J'])
    c = R.wavedec3(G[f'wavedec3_{tag}_x'], w, 'zero', J)
    assert _err(c[0], G[f'wavedec3_{tag}_lll']) < TOL
    for lvl in range(J):
        d = G[f'wavedec3_{tag}_d{lvl}']
        for i, k in enumerate(R.BANDS3[1:]):
            assert _err(c[1 + lvl][k], d[:, i]) < TOL
    assert _err(R.waverec3(c, w, 'zero'), G[f'waverec3_{tag}_x']) < TOL


def test_multilevel_packing_reference_sizes():
    """burgers/wave_trans.py:43-62 with J = 3 at the reference size 81 x 120: bands live on a 44 x 60 grid."""
    yl, yh = R.wavedec2(G['wavedec2_ref3_x'], 'bior2.4', 'periodization', 3)
    assert [h.shape[-2:] for h in yh] == [(41, 60), (21, 30), (11, 15)] and yl.shape[-2:] == (11, 15)
    t = R.burgers_coef_to_tensor(yl, yh)
    assert t.shape == (1, 2, 10, 44, 60)
    assert np.array_equal(t[:, :, 0, ::4, ::4], yl) and np.array_equal(t[:, :, 0, 3::4, 3::4], yl)
    assert np.array_equal(t[:, :, 1:4, :41], yh[0]) and np.array_equal(t[:, :, 1:4, 43], yh[0][:, :, :, 40])
    assert np.array_equal(t[:, :, 4:7, :42:2, ::2], yh[1]) and np.array_equal(t[:, :, 4:7, 43, 1::2], yh[1][:, :, :, 20])
    assert np.array_equal(t[:, :, 7:10, ::4, ::4], yh[2])


# ----------------------------------------------------------------------------------------- the smoke control objective (8f rank 2)
GUIDANCE_CASES = ('full', 'full_control', 'small_b2', 'small_no_weights')


def guidance_case(name):
    """(x, rescaler, init_u, reference gradient [B, F, 42, H, W] fp64, keyword arguments) of a case of tests/golden/ref_guidance.npz."""
    from tests.helpers import guidance_input, load_npz
    gz = load_npz('ref_guidance.npz')
    meta = [int(v) for v in gz[f'{name}::meta']]
    tshape, shape, ori, control = tuple(meta[:5]), tuple(meta[5:8]), tuple(meta[8:11]), bool(meta[11])
    w_e, w_i = (float(v) for v in gz[f'{name}::weights'])
    x, resc, init_u = guidance_input(tshape, shape, ori, seed=sum(name.encode()))
    g = np.zeros(tshape)
    tc, hc, wc = shape
    g[:, :tc, :40, :hc, :wc] = gz[f'{name}::g_coef']
    g[:, :tc, -1] = gz[f'{name}::g_smokeout']
    assert float(gz[f'{name}::g_rest_absmax']) == 0.0          # the reference's gradient is zero in the padding
    return x, resc, init_u, g, dict(shape=shape, ori_shape=ori, w_energy=w_e, w_init=w_i, is_condition_control=control)


@pytest.mark.parametrize('name', GUIDANCE_CASES)
def test_guidance_objective_vs_reference_run(name):
    """oracle/guidance_ref.py against gradients the REFERENCE's guidance_fn returned (smoke/inference_2d.py:30-66 run by
    tests/golden/make_ref_guidance_golden.py): J is at most quadratic, so <g_ref, v> must equal the central difference of the restated J."""
    from oracle import guidance_ref as G
    x, resc, init_u, g, kw = guidance_case(name)
    xs = x.double().numpy() * resc.double().numpy()
    gn = float(np.sqrt((g ** 2).sum()))
    for seed in range(3):
        v = np.random.default_rng(100 + seed).standard_normal(xs.shape)
        if seed == 2:                   # a direction along the gradient itself: a missing term cannot hide in an orthogonal direction
            v = g / gn * np.sqrt(v.size)
        fd = G.directional_derivative(xs, v, kw['shape'], kw['ori_shape'], init_u.double().numpy(), kw['w_energy'], kw['w_init'], kw['is_condition_control'])
        tol = 1e-6 * gn * np.sqrt(v.size) if seed == 2 else 1e-5 * gn          # 1e-6 of <g, v> along g; 1e-5 of its typical size otherwise (the fixture is fp32)
        assert abs((g * v).sum() - fd) < tol, (name, seed, (g * v).sum(), fd)
