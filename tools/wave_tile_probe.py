"""Round-4 experiment: 128 x 64 accumulator tile per compute wave (512 x 64 block tiles on 16-channel stages, debug 45)
against the 64 x 64 wave tile of the 256 x 64 blocks, on the level-0 3x3x3 layers: time, shader cycles and clock of the slowest compute
wave (--stamps of tools/bench_conv.py), result vs the default kernel.   python tools/wave_tile_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wdno_amd import ops

lib = ops._lib_()


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, xs, ws in (('l0 3x3x3 64->64', (8, 24, 40, 40, 64), (64, 64, 3, 3, 3)), ('l0 3x3x3 128->64', (8, 24, 40, 40, 128), (64, 128, 3, 3, 3)),
                     ('l1 3x3x3 256->64', (8, 24, 20, 20, 256), (64, 256, 3, 3, 3)), ('b2 [2,24,40,40] 64->64', (2, 24, 40, 40, 64), (64, 64, 3, 3, 3))):
    x = torch.randn(*xs, device='cuda')
    w = torch.randn(*ws, device='cuda') * 0.05
    ks, st, pd = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    cp, kp = xs[-1], 64
    xpl = ops.split_f16(x.reshape(-1, cp))
    f = lambda: ops.conv_fwd_h3(xpl, tuple(xs[:4]), w, ops.pack_fwd, 'f', None, None, ks, st, pd, kp)
    P = xs[0] * xs[1] * xs[2] * xs[3]
    flops = 2.0 * P * 64 * xs[-1] * 27
    res = {}
    for dbg in (0, 45):
        lib.wdno_set_debug(dbg)
        y = f().clone()
        t = timeit(f)
        vals = {}
        for mode in (23, 24, 26):
            lib.wdno_set_debug(100 * 0 + mode if dbg == 0 else mode)      # stamps run on the default selection only (debug is one integer)
            if dbg == 45:
                break
            rec = ops._amax_slot(x.device)
            for _ in range(5):
                rec.zero_()
                ops.conv_fwd_h3(xpl, tuple(xs[:4]), w, ops.pack_fwd, 'f', None, None, ks, st, pd, kp, amax_rec=rec)
            torch.cuda.synchronize()
            vals[mode] = float(rec.max())
        lib.wdno_set_debug(0)
        res[dbg] = (t, y, vals)
    d = (res[45][1] - res[0][1]).abs().max().item() / res[0][1].abs().max().item()
    tiles = {0: -(-P // 256), 45: -(-P // 512)}
    print(f'{name:24s} 64x64 per wave ({tiles[0]} tiles of 256x64): {res[0][0]:.4f} ms {flops / res[0][0] / 1e9:6.1f} TF/s | '
          f'128x64 per wave ({tiles[45]} tiles of 512x64): {res[45][0]:.4f} ms {flops / res[45][0] / 1e9:6.1f} TF/s | max rel diff {d:.1e}', flush=True)
