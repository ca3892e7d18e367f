import sys, torch, json
sys.path.insert(0, ".")
import bench
from wdno_amd.trainer import TrainStep, multistep_lr
dev = torch.device("cuda", 0)
dif = bench.build_model(dev, 8)
ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
x = (torch.randn(8, 24, 42, 40, 40) * 0.5).to(dev)
for _ in range(2): ts.step(x)
print(json.dumps(bench.smoke_pipeline_leg(ts, dev, 8, 5)))
