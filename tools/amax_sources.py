"""Which kernel produced each tensor that the split-fp16 convolutions take an amax of? (one train step of the bench model)
Tags the outputs of every autograd Function in wdno_amd.ops and tallies split_f16 calls by (direction, producer tag)."""
import collections
import sys
import torch
import wdno_amd.ops as ops

stats = collections.defaultdict(lambda: [0, 0])
cur = ['?']


def tag_outputs(out, tag):
    for t in (out if isinstance(out, tuple) else (out,)):
        if isinstance(t, torch.Tensor):
            try:
                t._prod = (tag, t._version)
            except Exception:
                pass
    return out


def src(t):
    p = getattr(t, '_prod', None)
    if p is None:
        return 'untagged'
    return p[0] if p[1] == t._version else p[0] + '+inplace'


for name in dir(ops):
    cls = getattr(ops, name)
    if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
        f0, b0 = cls.forward, cls.backward

        def mk(f0, b0, name):
            def fwd(ctx, *a):
                if name in ('_Conv', '_ConvT'):
                    cur[0] = 'fwd<-' + src(a[0])
                return tag_outputs(f0(ctx, *a), name + '.fwd')

            def bwd(ctx, *g):
                if name in ('_Conv', '_ConvT'):
                    cur[0] = 'bwd<-' + src(g[0])
                return tag_outputs(b0(ctx, *g), name + '.bwd')
            return fwd, bwd
        f, b = mk(f0, b0, name)
        cls.forward = staticmethod(f)
        cls.backward = staticmethod(b)

for fn in ('split_f16', 'split_f16_colsum'):
    o = getattr(ops, fn)

    def mk2(o):
        def w(x2d, amax=None):
            s = stats[cur[0] + (' (record)' if amax is not None else '')]
            s[0] += 1
            s[1] += x2d.numel() * 4
            return o(x2d, amax)
        return w
    setattr(ops, fn, mk2(o))

import bench
sys.argv = ['bench.py', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--sample-steps', '0']
bench.main()
tot = sum(v[1] for v in stats.values())
for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f'{k:40s} calls {v[0]:4d}  MB {v[1] / 1e6:9.1f}  {100 * v[1] / tot:5.1f}%')
