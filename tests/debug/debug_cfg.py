"""GPU debugging aid: Burgers p_losses HIP vs oracle over a grid of configurations."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from wdno_amd import tree_path
for t in ('third_party', 'smoke', 'burgers'):
    sys.path.insert(0, tree_path(t))
from ddpm_burgers.unet import Unet2D
from ddpm_burgers.diffusion_1d import GaussianDiffusion as GD1
from oracle import diffusion_ref as D, unet_ref as U

def run(b, hw, mults, padded, flags, groups=1):
    torch.manual_seed(0)
    net = Unet2D(dim=8, dim_mults=mults, channels=9, resnet_block_groups=groups)
    dif = GD1(net, seq_length=hw, padded_shape=list(padded), ori_shape=[10, 14], loss_layer_weight=torch.ones(1, 9, 1, 1),
              is_condition_pad=flags.get('pad', False), is_condition_u0=flags.get('u0', False), is_condition_uT=flags.get('uT', False), is_condition_f=flags.get('f', False))
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x0 = torch.randn(b, 9, *hw) * 0.5
    t = torch.randint(0, 1000, (b,))
    noise = torch.randn(b, 9, *hw)
    model = lambda x, tt: U.unet2d_forward(sd, x, tt, dim=8, dim_mults=mults, groups=groups)
    with torch.no_grad():
        ref = D.burgers_p_losses(model, D.make_buffers('cosine', 1000), x0, t, noise, padded_shape=padded, loss_layer_weight=torch.ones(1, 9, 1, 1), flags=flags)
        out_ref = model(x0, t)
        dif = dif.cuda()
        got = dif.p_losses(x0.cuda(), t.cuda(), noise=noise.cuda())
        out = net(x0.cuda(), t.cuda())
    e = ((out.cpu() - out_ref).norm() / out_ref.norm()).item()
    print(f'b={b} hw={hw} mults={mults} flags={flags}: loss {got.item():.6f} vs {ref.item():.6f}  unet rel {e:.2e}', flush=True)

F = dict(pad=True, u0=True, f=True)
run(4, (8, 8), (1, 2), (6, 7), F)
run(4, (8, 8), (1, 2), (6, 7), {})
run(2, (8, 8), (1, 2), (6, 7), F)
run(1, (8, 8), (1, 2), (6, 7), F)
run(4, (16, 16), (1, 2), (11, 14), F)
run(4, (8, 8), (1, 2, 4), (6, 7), F)
run(3, (8, 16), (1, 2), (6, 7), F)
run(8, (16, 8), (1, 2), (6, 7), F)
