"""Scalar DDIM coefficients of the sigmoid schedule as THIS host's torch CPU evaluates them (diffusion_2d.py:905-909), with every
intermediate, to compare hosts: c = sqrt(1 - a' - sigma^2) cancels to ~1e-6 from terms ~1, so a one-ulp difference in any scalar op
moves c by 2 %."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wdno_amd import diffusion_core as K  # noqa: E402

b = K.sigmoid_beta_schedule(1000)
ac = torch.cumprod(1 - b, 0).float()
h = lambda t: hex(t.view(torch.int32).item() & 0xffffffff)
for time, nxt in ((999, 749), (749, 499), (499, 249)):
    a, an = ac[time], ac[nxt]
    r = a / an
    p = (1 - r) * (1 - an)
    q = p / (1 - a)
    s = 1.0 * q.sqrt()
    s2 = s ** 2
    c2 = 1 - an - s2
    c = c2.sqrt()
    print(time, nxt, 'a', h(a), 'an', h(an), 'a/an', h(r), 'p', h(p), 'q', h(q), 'sigma', h(s), 'sigma^2', h(s2), 'c^2', h(c2), 'c', h(c), float(c), 'sqrt(an)', h(an.sqrt()))
    print('   K.ddim_coefficients', K.ddim_coefficients(ac, time, nxt, 1.0))
