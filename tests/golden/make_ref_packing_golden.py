"""Golden vectors for the packing helpers (rows P1-P3 of SURVEY.md section 8) by importing the reference.

Build-container only (needs /root/reference):   python tests/golden/make_ref_packing_golden.py
Writes tests/golden/ref_packing.npz -- data only (random inputs, the reference functions' outputs).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_ref_golden as M  # noqa: E402

M.STUBS['matplotlib/__init__.py'] = ''
M.STUBS['matplotlib/pyplot.py'] = ''
M.install_stubs()

import wave_trans as B        # noqa: E402  burgers/wave_trans.py
import wave_trans_2d as S     # noqa: E402  smoke/wave_trans_2d.py
from ddpm_burgers.wavelet_utils import upsample_coef as up_b   # noqa: E402
from ddpm.wave_utils import upsample_coef as up_s               # noqa: E402

g = torch.Generator().manual_seed(11)
d = {}

# P1 burgers coef_to_tensor J = 1 (pad and no pad), J = 3 at the reference level sizes (81x120 -> 41x60, 21x30, 11x15)
yl = torch.randn(1, 2, 41, 60, generator=g)
yh = [torch.randn(1, 2, 3, 41, 60, generator=g)]
d['b_j1_yl'], d['b_j1_yh0'] = yl.numpy(), yh[0].numpy()
d['b_j1_out'] = B.coef_to_tensor(yl, yh).numpy()
d['b_j1_out_pad'] = B.coef_to_tensor(yl[..., :40, :], [yh[0][..., :40, :]], pad=True).numpy()
yl3 = torch.randn(1, 2, 11, 15, generator=g)
yh3 = [torch.randn(1, 2, 3, 41, 60, generator=g), torch.randn(1, 2, 3, 21, 30, generator=g), torch.randn(1, 2, 3, 11, 15, generator=g)]
d['b_j3_yl'] = yl3.numpy()
for i, h in enumerate(yh3):
    d[f'b_j3_yh{i}'] = h.numpy()
d['b_j3_out'] = B.coef_to_tensor(yl3, yh3).numpy()
# tensor_to_coef / tensor_to_coef_super
ct = torch.randn(1, 9, 64, 64, generator=g)
a, b = B.tensor_to_coef(ct, (41, 60))
d['b_t2c_in'], d['b_t2c_yl'], d['b_t2c_yh'] = ct.numpy(), a.numpy(), b[0].numpy()
a, b = B.tensor_to_coef_super(ct, (40, 60))
d['b_t2cs_yl'], d['b_t2cs_yh'] = a.numpy(), b[0].numpy()

# P2 smoke
lll = torch.randn(5, 6, 7, 7, generator=g)
det = {k: torch.randn(5, 6, 7, 7, generator=g) for k in ('aad', 'ada', 'add', 'daa', 'dad', 'dda', 'ddd')}
d['s_c2t_lll'] = lll.numpy()
d['s_c2t_det'] = np.stack([det[k].numpy() for k in det], axis=1)
d['s_c2t_out'] = S.coef_to_tensor([lll, det]).numpy()
ct = torch.randn(2, 42, 8, 10, 10, generator=g)
d['s_t2c_in'] = ct.numpy()
for ut in (None, 'time', 'space'):
    a, b = S.tensor_to_coef(ct, (6, 7, 7), ut)
    d[f's_t2c_{ut}_yl'] = a.numpy()
    d[f's_t2c_{ut}_yh'] = np.stack([b[k].numpy() for k in ('aad', 'ada', 'add', 'daa', 'dad', 'dda', 'ddd')], axis=1)

# P3 upsample_coef
w = torch.randn(2, 8, 5, 6, generator=g)
d['up_b_in'], d['up_b_out'] = w.numpy(), up_b(w, None).numpy()
w = torch.randn(2, 3, 4, 5, 5, generator=g)
d['up_s_in'] = w.numpy()
d['up_s_time'] = up_s(w, None, 'time').numpy()
d['up_s_space'] = up_s(w, None, 'space').numpy()

dst = os.path.join(HERE, 'ref_packing.npz')
np.savez_compressed(dst, **d)
print('wrote', dst, os.path.getsize(dst))
