"""Which torch-CPU elementwise functions return different bits on this host than on another one? Prints CRCs of fixed evaluations
(run in the build container and on the GPU box and diff): vectorised libm functions (Sleef) are only accurate to ~1 ulp and the code
path depends on the CPU's ISA, so fp32 results of the REFERENCE itself differ between hosts wherever such a function feeds an
ill-conditioned step (the t = 999 DDIM step multiplies the U-Net's output by 1.8e3)."""
import math
import zlib

import torch

torch.manual_seed(0)
x = torch.linspace(-8, 8, 100001)
big = torch.linspace(0, 1000, 100001)


def crc(t):
    return zlib.crc32(t.contiguous().numpy().tobytes())


print('cpu capability', torch.backends.cpu.get_cpu_capability(), 'threads', torch.get_num_threads())
for dim in (8, 64, 128):
    half = dim // 2
    f = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    print('freq table', dim, crc(f), 'correctly rounded:', int((f == torch.exp((torch.arange(half) * -(math.log(10000) / (half - 1))).double()).float()).sum()), '/', half)
    arg = torch.tensor([999.0, 749.0, 499.0, 249.0])[:, None] * f[None, :]
    print('  sin/cos of t*f', crc(arg.sin()), crc(arg.cos()))
for name, fn in (('exp', torch.exp), ('sigmoid', torch.sigmoid), ('silu', torch.nn.functional.silu), ('gelu', torch.nn.functional.gelu), ('erf', torch.erf),
                 ('tanh', torch.tanh), ('rsqrt', torch.rsqrt), ('sqrt', lambda v: torch.sqrt(v.abs())), ('log', lambda v: torch.log(v.abs() + 1e-3)),
                 ('softmax', lambda v: torch.softmax(v[:99995].reshape(-1, 7), -1))):
    print(name, crc(fn(x)))
print('sin(big)', crc(big.sin()), 'cos(big)', crc(big.cos()))
w = torch.randn(16, 8, 3, 3, 3)
a = torch.randn(2, 8, 6, 10, 10)
print('conv3d', crc(torch.nn.functional.conv3d(a, w, padding=1)), 'linear', crc(torch.nn.functional.linear(torch.randn(64, 256), torch.randn(128, 256))),
      'groupnorm', crc(torch.nn.functional.group_norm(a, 4)), 'sum', crc(a.sum((2, 3, 4))), 'var', crc(a.var(dim=1, unbiased=False)))
