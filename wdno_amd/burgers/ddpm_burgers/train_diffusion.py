"""Trainer (1-D Burgers) -- drop-in for burgers/ddpm_burgers/train_diffusion.py:40-237 on MI355X.

Same constructor keywords, attributes (`model`, `opt`, `ema`, `step`, `results_folder`, `device`), checkpoint file
naming (`{is_wavelet}-cos10000-model-{milestone}.pt`) and checkpoint dictionary keys
(`step / model / opt / ema / scaler / loss`, with `opt` in torch.optim.Adam layout) as the reference, so
`trainer.train()` / `trainer.load(milestone)` keep working for train_ddpm_burgers.py:187-200 and test_util.py:221-263.

What runs differently: the step is the flat-buffer path of wdno_amd.trainer (one RCCL all-reduce of the gradient,
clip + Adam in two HIP launches, EMA in one); `accelerate` is not used -- launch one process per GPU with torchrun and
the process group is picked up; tensorboard logging is replaced by stdout (same cadence, `test_every`).
"""
import os
from datetime import datetime
from multiprocessing import cpu_count
from pathlib import Path

import torch
import torch.distributed as dist

from wdno_amd.trainer import TrainerCore, cosine_annealing_lr
from ddpm_burgers.model_utils import cycle, exists, has_int_squareroot


class Trainer(TrainerCore):
    def __init__(
        self,
        diffusion_model,
        dataset,
        *,
        is_super_model=False,
        wave_type='db4',
        pad_mode='zero',
        rescaler=1,
        exp_name='',
        train_batch_size=16,
        gradient_accumulate_every=1,
        train_lr=1e-4,
        train_num_steps=100000,
        ema_update_every=10,
        ema_decay=0.995,
        adam_betas=(0.9, 0.99),
        test_every=1000,
        save_and_sample_every=1000,
        num_samples=25,
        results_folder='./results',
        amp=False,
        mixed_precision_type='fp16',
        split_batches=True,
        max_grad_norm=1.,
    ):
        assert has_int_squareroot(num_samples), 'number of samples must have an integer square root'
        super().__init__(diffusion_model, train_batch_size=train_batch_size, gradient_accumulate_every=gradient_accumulate_every,
                         train_lr=train_lr, train_num_steps=train_num_steps, ema_update_every=ema_update_every, ema_decay=ema_decay,
                         adam_betas=adam_betas, save_and_sample_every=save_and_sample_every, split_batches=split_batches,
                         max_grad_norm=max_grad_norm, results_dir=results_folder,
                         mixed_precision=mixed_precision_type if amp else 'no',                            # train_diffusion.py:71-74 (Accelerator)
                         lr_schedule=lambda base, step: cosine_annealing_lr(base, step, 10000, 0.0))      # train_diffusion.py:118
        self.is_super_model = is_super_model
        self.wave_type = wave_type
        self.pad_mode = pad_mode
        self.rescaler = rescaler.to(self.device) if torch.is_tensor(rescaler) else rescaler
        self.exp_name = exp_name
        self.num_samples = num_samples
        self.test_every = test_every
        self.results_folder = Path(results_folder)
        workers = min(cpu_count(), 16) if self.num_workers is None else self.num_workers
        if not is_super_model:
            dl = self.make_loader(dataset, self.local_batch_size, workers)
        else:                                      # a list of datasets, one group drawn at random per batch (data_burgers_1d.py SuperDataLoader)
            from ddpm_burgers.data_burgers_1d import SuperDataLoader
            dl = SuperDataLoader(dataset, batch_size=self.local_batch_size, shuffle=True, pin_memory=True, num_workers=workers,
                                 seed=self.data_seed if self.world > 1 else None)
        self.dl = self.cycle(dl)          # advances DistributedSampler epochs

    def _path(self, milestone):
        if type(milestone) is int:
            return str(self.results_folder / f'{self.model.is_wavelet}-cos10000-model-{milestone}.pt')
        return str(self.results_folder / milestone)

    def save(self, milestone):
        if not self.is_main_process:
            return
        data = self.checkpoint_dict()
        data['loss'] = self.total_loss
        torch.save(data, self._path(milestone))

    def load(self, milestone):
        data = torch.load(self._path(milestone), map_location=self.device, weights_only=False)
        self.load_checkpoint_dict(data)
        if 'version' in data:
            print(f"loading from version {data['version']}")

    def train(self):
        print(self.device)
        while self.step < self.train_num_steps:
            total_loss = self.optimisation_step(lambda: next(self.dl).to(self.device, non_blocking=True))
            if self.is_main_process:
                self.ema.update()
                if self.step != 0 and (self.step + 1) % self.save_and_sample_every == 0:
                    self.ema.ema_model.eval()
                    self.save(self.step // self.save_and_sample_every)
                if self.step % self.test_every == 0:
                    print(datetime.now(), f'Step: {self.step}, Total error: {total_loss}')
            self.step += 1
        if self.is_main_process:
            print('training completes')
