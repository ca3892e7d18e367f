"""Times the softmax attention over the pixels of a frame (the mid spatial attention, conv3d.py:450-452: no rotation, no bias) on the
tiled MFMA forward kernel (csrc/attention.hip: attn_fwd_mfma_tiled_kernel) against the thread-per-row kernel (library debug mode 67), checks
that they agree and that repeated launches return the same bits. Usage: python tools/bench_spatial_attn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wdno_amd import ops  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    lib = ops._lib_()
    torch.manual_seed(0)
    for units, h, w in ((192, 10, 10), (96, 20, 20), (48, 24, 24), (192, 9, 9)):
        n = h * w
        qkv = torch.randn(units * n, 384, device='cuda')
        with torch.no_grad():
            f = lambda: ops.softmax_attention(qkv, 4, units, 1, n, n, 0, 1, 32 ** -0.5)
            y = f().clone()
            same = sum(int(not torch.equal(y, f())) for _ in range(20))
            t_new = timed(f)
            lib.wdno_set_debug(67)
            y_old = f().clone()
            t_old = timed(f)
            lib.wdno_set_debug(0)
        print(f'{units} units x {n} tokens: tiled MFMA {t_new:7.1f} us   thread per row {t_old:7.1f} us   rel. difference {float((y - y_old).norm() / y_old.norm()):.2e}'
              f'   repeats that differ {same}/20', flush=True)


def main_bwd():
    lib = ops._lib_()
    torch.manual_seed(1)
    for units, h, w in ((192, 10, 10), (96, 20, 20), (192, 9, 9)):
        n = h * w
        qkv = torch.randn(units * n, 384, device='cuda', requires_grad=True)
        go = torch.randn(units * n, 128, device='cuda')

        def fb():
            qkv.grad = None
            out = ops.softmax_attention(qkv, 4, units, 1, n, n, 0, 1, 32 ** -0.5)
            out.backward(go)
            return qkv.grad

        def t(reps=20):              # forward + backward replayed from one captured graph (launch by launch the host is the limit)
            for _ in range(3):
                fb()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(reps):
                    fb()
            g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e3 / reps
        g_new = fb().clone()
        same = sum(int(not torch.equal(g_new, fb())) for _ in range(20))
        t_new = t()
        lib.wdno_set_debug(68)
        g_old = fb().clone()
        t_old = t()
        lib.wdno_set_debug(0)
        print(f'forward + backward, {units} units x {n} tokens: tiled MFMA backward {t_new:7.1f} us   thread per row {t_old:7.1f} us   '
              f'rel. difference of dqkv {float((g_new - g_old).norm() / g_old.norm()):.2e}   repeats that differ {same}/20', flush=True)


if __name__ == '__main__':
    main()
    main_bwd()
