"""The arbiter evaluations (tests/arbiter.py) pinned: run in fp32 they reproduce the reference-generated fixtures (the oracle IS the
reference's arithmetic on the same torch CPU kernels); their fp64 runs then measure how far the reference's own fp32 result is from the
exact chain -- the yardstick of the GPU gates. CPU only."""
import numpy as np
import torch

from tests import arbiter as A
from tests.helpers import rel_l2


def test_guided_chains_fp32_oracle_reproduces_reference():
    gz = A.sub('gs')
    o = A.smoke_guided(torch.float32)
    assert rel_l2(o['ddim'], gz['ddim_out']) < 1e-6 and rel_l2(o['ddpm6'], gz['ddpm6_out']) < 1e-6
    e = A.smoke_guided(torch.float64)
    ref_ddim, ref_ddpm = rel_l2(gz['ddim_out'], e['ddim']), rel_l2(gz['ddpm6_out'], e['ddpm6'])
    print('guided: reference vs exact  ddim4', ref_ddim, ' ddpm6', ref_ddpm)
    assert ref_ddim < 1e-4 and ref_ddpm < 1e-5          # (2.1e-5 and 9e-7: the DDIM chain from t = 999 is the ill-conditioned one)


def test_cascade_fp32_oracle_reproduces_reference():
    gz = A.sub('sr')
    o = A.smoke_cascade(torch.float32)
    assert rel_l2(o['wave0'], gz['wave0']) < 1e-6
    assert np.array_equal(A.smoke_cascade(torch.float32, wave0=torch.from_numpy(gz['wave0']))['low'].numpy(), gz['low'])      # index work: bit-exact
    assert rel_l2(o['wave1'], gz['wave1']) < 1e-6
    e = A.smoke_cascade(torch.float64)
    print('cascade: reference vs exact  base', rel_l2(gz['wave0'], e['wave0']), ' end to end', rel_l2(gz['wave1'], e['wave1']))


def test_three_step_trainer_oracles_reproduce_reference():
    from tests.helpers import load_npz
    gb = load_npz('ref_train_burgers.npz')
    losses, gns, fin = A.burgers_train3(torch.float32)
    for i in range(3):
        assert abs(losses[i] - float(gb[f's{i}_loss'])) < 2e-6 * abs(float(gb[f's{i}_loss'])) and abs(gns[i] - float(gb[f's{i}_gnorm'])) < 2e-5 * float(gb[f's{i}_gnorm'])
    w0 = {k[len('w0::model.'):]: torch.from_numpy(gb[k]) for k in gb.files if k.startswith('w0::model.')}
    ref = {k[len('w3::model.'):]: torch.from_numpy(gb[k]) for k in gb.files if k.startswith('w3::model.')}
    _, tot = A.update_errors(w0, fin, ref)
    print('burgers: fp32 oracle updates vs reference', tot)
    assert tot < 1e-3
    gs = A.sub('t2')
    losses, gns, fin = A.smoke_train3(torch.float32)
    for i in range(3):
        assert abs(losses[i] - float(gs[f's{i}_loss'])) < 2e-6 * abs(float(gs[f's{i}_loss'])) and abs(gns[i] - float(gs[f's{i}_gnorm'])) < 2e-5 * float(gs[f's{i}_gnorm'])
    w0 = {k[len('w0::model.'):]: torch.from_numpy(v) for k, v in gs.items() if k.startswith('w0::model.')}
    ref = {k[len('w3::model.'):]: torch.from_numpy(v) for k, v in gs.items() if k.startswith('w3::model.')}
    _, tot = A.update_errors(w0, fin, ref)
    print('smoke: fp32 oracle updates vs reference', tot)
    assert tot < 1e-3
    for name, fn, g, pre in (('burgers', A.burgers_train3, {k: gb[k] for k in gb.files}, ''), ('smoke', A.smoke_train3, gs, '')):
        le, ge, fe = fn(torch.float64)
        w0 = {k[len('w0::model.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w0::model.')}
        ref = {k[len('w3::model.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w3::model.')}
        per, tot = A.update_errors(w0, ref, fe)
        print(name, 'reference vs exact: loss', [abs(float(g[f's{i}_loss']) - le[i]) / le[i] for i in range(3)], 'gnorm',
              [abs(float(g[f's{i}_gnorm']) - ge[i]) / ge[i] for i in range(3)], 'updates total', tot, 'worst parameter', max(per.values()))
