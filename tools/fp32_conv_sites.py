"""Which convolutions of the bench model run on the exact-fp32 kernels (not the split-fp16 path), and how big are they?"""
import collections, sys
import torch
import wdno_amd.ops as ops
stats = collections.Counter()
orig = ops.conv_fwd_raw
def raw(x, wp, bias_p, residual, ks, st, pd, kp):
    stats[('fwd/dgrad', tuple(x.shape), kp, ks, st)] += 1
    return orig(x, wp, bias_p, residual, ks, st, pd, kp)
ops.conv_fwd_raw = raw
ow = ops.conv_wgrad_raw
def wraw(x, dy, ks, st, pd):
    stats[('wgrad', tuple(x.shape), dy.shape[-1], ks, st)] += 1
    return ow(x, dy, ks, st, pd)
ops.conv_wgrad_raw = wraw
ot = ops.conv_transpose_raw
def traw(x, wt, bias_p, cout_p):
    stats[('transpose', tuple(x.shape), cout_p)] += 1
    return ot(x, wt, bias_p, cout_p)
ops.conv_transpose_raw = traw
import bench
sys.argv = ['bench.py', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--sample-steps', '0']
bench.main()
for k, v in sorted(stats.items(), key=lambda kv: -kv[1]):
    print(v, k)
