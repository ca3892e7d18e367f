"""BASELINE.json configs[4] on one GPU: super-resolution DDPM sampling of the smoke task followed by the IDWT reconstruction.
The cascade step of inference_2d.py:run_super_model (:155-232) for the space-SR model at level 1:
    low-resolution coefficients (base sample, nearest-upsampled) -> condition `low` -> DDIM steps of the 82-channel U-Net on
    [B, 24, 82, 80, 80] -> tensor_to_coef(upsample_type='space') -> waverec3 -> fields [B, 5, 32, 128, 128].
Random weights / inputs (no checkpoints offline); prints DDIM steps/s and the reconstruction time."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wdno_amd import _lib, tree_path
for t in ('third_party', 'smoke', 'burgers'):
    sys.path.insert(0, tree_path(t))
import ptwt, pywt
from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
from ddpm.diffusion_2d import GaussianDiffusion
from ddpm.wave_utils import upsample_coef
from wave_trans_2d import tensor_to_coef

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--steps', type=int, default=6)
args = ap.parse_args()
_lib.load()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
B = args.batch
shapes = [[18, 34, 34], [18, 66, 66]]            # coefficient shapes of the base level and of the 128 x 128 level
net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=82)
dif = GaussianDiffusion(net, torch.ones(1, 1, 82, 1, 1), True, True, True, True, 'bior1.3', 'zero', shapes, [[32, 64, 64], [32, 128, 128]],
                        image_size=80, frames=24, timesteps=1000, sampling_timesteps=args.steps, loss_type='l2').to(dev)
base = torch.randn(B, 24, 40, 40, 40, device=dev) * 0.3                      # coefficient channels of a base-resolution sample
with torch.no_grad():
    low = upsample_coef(base[:, :18, :, :34, :34].contiguous(), None, 'space')                       # [B, 18, 40, 68, 68]
    low = torch.nn.functional.pad(low, (0, 80 - 68, 0, 80 - 68, 0, 0, 0, 24 - 18))
    init = torch.randn(B, 24, 80, 80, device=dev)
    control = torch.randn(B, 24, 16, 80, 80, device=dev)
    for _ in range(2):                                                        # warm-up (kernel attributes, pixel tables, weight packs)
        out = dif.sample(batch_size=B, N_upsample=1, init=init, control=control, low=low)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = dif.sample(batch_size=B, N_upsample=1, init=init, control=control, low=low)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'SR sampling [B={B}, 24, 82, 80, 80]: {args.steps} DDIM steps in {dt*1e3:.1f} ms -> {args.steps/dt:.2f} steps/s ({args.steps*B/dt:.1f} sample-steps/s)')
    t0 = time.perf_counter()
    for _ in range(5):
        coef = tensor_to_coef(out[:, :, :40].permute(0, 2, 1, 3, 4).contiguous(), shapes[1], upsample_type='space')
        rec = ptwt.waverec3([coef[0].contiguous(), {k: v.contiguous() for k, v in coef[1].items()}], pywt.Wavelet('bior1.3'))
        fields = rec[:, :32, :128, :128].reshape(-1, 5, 32, 128, 128)
    torch.cuda.synchronize()
    print(f'IDWT reconstruction to {tuple(fields.shape)}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms')
