"""What limits the large convolution kernels' clock? (prove the limiter.)

Runs pure loops of one kernel for a few seconds each and samples, for the GPU that is actually working:
  * hwmon at ~20 Hz: socket power (PPT, W), shader clock (MHz), junction / HBM temperature   (/sys/class/drm/card*/device/hwmon)
  * amd-smi throttle accumulators before / after the loop: PPT, PROCHOT, socket / VR / HBM thermal, and the per-XCD
    "gfx clock below host limit" accumulators split by cause (power / thermal)   (amd-smi metric --violation)
Loops: idle; level-0 3x3x3 64->64 forward (tap-resident split kernel) on random operands and on ALL-ZERO operands (same instruction
stream, no data toggling); its weight gradient; an HBM-bound kernel (fp16 split pass) for contrast.
    python tools/conv_limiter_probe.py [seconds_per_phase] > profiles/r04_conv_limiter.md"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from wdno_amd import ops  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
HW = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'))


def rd(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


def hw_sample():
    return [(rd(h + '/power1_input'), rd(h + '/freq1_input'), rd(h + '/temp2_input'), rd(h + '/temp3_input')) for h in HW]


def smi_violation():
    try:
        out = subprocess.run(['amd-smi', 'metric', '--violation', '--json'], capture_output=True, text=True, timeout=30).stdout
        data = json.loads(out)
        return data if isinstance(data, list) else data.get('gpu_data', data)
    except Exception as e:
        return {'error': repr(e)}


def flat(d, pre=''):
    out = {}
    if isinstance(d, dict):
        for k, v in d.items():
            out.update(flat(v, pre + k + '.'))
    elif isinstance(d, list):
        if all(isinstance(v, (int, float)) for v in d):
            out[pre[:-1]] = sum(d)
        else:
            for i, v in enumerate(d):
                out.update(flat(v, pre + str(i) + '.'))
    elif isinstance(d, (int, float)):
        out[pre[:-1]] = d
    return out


def phase(name, fn):
    before = smi_violation()
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append((time.perf_counter(), hw_sample()))
            time.sleep(0.05)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < SECS:
        if fn is None:
            time.sleep(0.05)
        else:
            for _ in range(50):
                fn()
            n += 50
            torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / max(n, 1)
    stop[0] = True
    th.join()
    after = smi_violation()
    return dict(name=name, launches=n, ms_per_launch=ms, samples=samples, before=before, after=after)


dev = 'cuda'
xs, ws_ = (8, 24, 40, 40, 64), (64, 64, 3, 3, 3)
x = torch.randn(*xs, device=dev)
w = torch.randn(*ws_, device=dev) * 0.05
ks, st, pd = (3, 3, 3), (1, 1, 1), (1, 1, 1)
xpl = ops.split_f16(x.reshape(-1, 64))
y = ops.conv_fwd_h3(xpl, xs[:4], w, ops.pack_fwd, 'f', None, None, ks, st, pd, 64)
ypl = ops.split_f16(y.reshape(-1, 64))
zx = torch.zeros_like(x)
zpl = ops.split_f16(zx.reshape(-1, 64), amax=torch.ones(ops.AMAX_FLOATS, device=dev))
wz = torch.zeros_like(w)
flops = 2.0 * x.numel() // 64 * 64 * 64 * 27
res = []
res.append(phase('idle', None))
res.append(phase('conv fwd 64->64 level 0 (random operands)', lambda: ops.conv_fwd_h3(xpl, xs[:4], w, ops.pack_fwd, 'f', None, None, ks, st, pd, 64)))
# round 4: the 128 x 64 accumulator tile per compute wave (512 x 64 block tiles on 16-channel stages, debug 45: 12 fragment reads per
# 24 matrix instructions instead of 8 per 12) -- at batch 8 (600 tiles: 2.34 rounds of 256 CUs against 4.69 of the 256 x 64 tiles) and at
# batch 10 (750 / 1500 tiles: both 97.7 % of their last round)
lib = ops._lib_()
def with_debug(mode, fn):
    def g():
        lib.wdno_set_debug(mode)
        fn()
        lib.wdno_set_debug(0)
    return g
res.append(phase('conv fwd 64->64 level 0, 128 x 64 accumulator tile per wave (512 x 64 blocks, debug 45)',
                 with_debug(45, lambda: ops.conv_fwd_h3(xpl, xs[:4], w, ops.pack_fwd, 'f', None, None, ks, st, pd, 64))))
xs10 = (10, 24, 40, 40, 64)
x10 = torch.randn(*xs10, device=dev)
xpl10 = ops.split_f16(x10.reshape(-1, 64))
F10 = 10.0 / 8.0
res.append(phase('batch 10: 64 x 64 per wave (1500 tiles of 256 x 64) [TFLOP/s column: x 1.25]', lambda: ops.conv_fwd_h3(xpl10, xs10[:4], w, ops.pack_fwd, 'f', None, None, ks, st, pd, 64)))
res.append(phase('batch 10: 128 x 64 per wave (750 tiles of 512 x 64, debug 45) [TFLOP/s column: x 1.25]',
                 with_debug(45, lambda: ops.conv_fwd_h3(xpl10, xs10[:4], w, ops.pack_fwd, 'f', None, None, ks, st, pd, 64))))
res.append(phase('conv fwd 64->64 level 0 (all-zero operands)', lambda: ops.conv_fwd_h3(zpl, xs[:4], wz, ops.pack_fwd, 'f', None, None, ks, st, pd, 64)))
res.append(phase('conv wgrad 64->64 level 0', lambda: ops.conv_wgrad_h3(xpl, xs[:4], ypl, tuple(y.shape[1:4]), ks, st, pd)))
res.append(phase('fp16 split pass (HBM-bound, 79 MB in / 79 MB out)', lambda: ops.split_f16(x.reshape(-1, 64))))

# which hwmon entry is the working GPU: the one whose power moved most between idle and the first conv loop
def mean_power(r, i):
    v = [s[1][i][0] for s in r['samples'] if s[1][i][0] is not None]
    return sum(v) / max(1, len(v))
gi = max(range(len(HW)), key=lambda i: mean_power(res[1], i) - mean_power(res[0], i))
print('# Clock limiter of the large convolution kernels (tools/conv_limiter_probe.py)\n')
print(f'GPU under test: {HW[gi]}; power cap {rd(HW[gi] + "/power1_cap") / 1e6:.0f} W; {SECS:.0f} s per loop; hwmon sampled every ~50 ms\n')
print('| loop | ms / launch | TFLOP/s (fp32-equivalent) | power W mean (min-max) | sclk MHz mean (min-max) | T junction / HBM C |')
print('|---|---:|---:|---:|---:|---:|')
for r in res:
    col = lambda k: [s[1][gi][k] for s in r['samples'][len(r['samples']) // 4:] if s[1][gi][k] is not None]       # skip the ramp
    pw, ck, tj, tm = col(0), col(1), col(2), col(3)
    tf = f'{flops / (r["ms_per_launch"] * 1e-3) / 1e12:.0f}' if 'conv' in r['name'] else '-'
    print(f'| {r["name"]} | {r["ms_per_launch"]:.4f} | {tf} | {sum(pw) / len(pw) / 1e6:.0f} ({min(pw) / 1e6:.0f}-{max(pw) / 1e6:.0f}) | '
          f'{sum(ck) / len(ck) / 1e6:.0f} ({min(ck) / 1e6:.0f}-{max(ck) / 1e6:.0f}) | {max(tj) / 1e3:.0f} / {max(tm) / 1e3:.0f} |')
print('\nThrottle accumulators that moved during each loop (amd-smi metric --violation, after - before, summed over the listed GPUs / XCDs; '
      'the ACCUMULATION_COUNTER is the time base):\n')
for r in res:
    fb, fa = flat(r['before']), flat(r['after'])
    moved = {k: fa[k] - fb[k] for k in fa if k in fb and isinstance(fa[k], (int, float)) and fa[k] != fb[k] and 'ACTIVITY' not in k.upper()}
    short = {k.split('throttle.')[-1] if 'throttle.' in k else k: v for k, v in moved.items()}
    print(f'* **{r["name"]}**: ' + (', '.join(f'{k} +{v}' for k, v in sorted(short.items())) or 'none'))
