"""GaussianDiffusion (smoke, 3-D wavelet-coefficient videos) on MI355X -- drop-in for smoke/ddpm/diffusion_2d.py:568-1058.

Same constructor signature, attributes, buffer names and method surface (forward / p_losses / q_sample / p_sample /
p_sample_loop / ddim_sample / sample / model_predictions) as the reference. What changed is *how* a step runs:

  training : one fused launch builds the noisy state, imposes every conditioning overwrite (init density, control
             channels, zero padding, low-resolution channels) and produces the masked noise target
             (diffusion_2d.py:1003-1033); the U-Net runs on HIP kernels; one fused launch reduces the loss.
  sampling : per step, one launch for the posterior / DDIM update and one for re-imposing the conditions, instead of
             ~20 indexed assignments and ~10 broadcast arithmetic ops.

Only the wavelet parametrisation (is_wavelet=True) is implemented: that is the WDNO path (train_2d.py:104-121).
All draws go through `self.sample_noise`, so tests can inject the exact noise the reference consumed.
"""
import math
from collections import namedtuple

import torch
from torch import nn

import wdno_amd
from wdno_amd import diffusion_core as K
from wdno_amd.trainer import TrainerCore as _TrainerCore, multistep_lr as _multistep_lr

_reference_getattr = wdno_amd.reference_fallthrough('ddpm.diffusion_2d', __file__)

ModelPrediction = namedtuple('ModelPrediction', ['pred_noise', 'pred_x_start'])


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def identity(t, *args, **kwargs):
    return t


def extract(a, t, x_shape):
    b, *_ = t.shape
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


linear_beta_schedule = K.linear_beta_schedule
cosine_beta_schedule = K.cosine_beta_schedule
sigmoid_beta_schedule = K.sigmoid_beta_schedule


class GaussianDiffusion(nn.Module):
    def __init__(
        self,
        model,
        loss_layer_weight,
        is_condition_control,
        is_condition_pad,
        is_wavelet,
        is_super_model,
        wave_type,
        pad_mode,
        padded_shape,
        ori_shape,
        *,
        image_size,
        frames,
        timesteps=1000,
        sampling_timesteps=None,
        loss_type='l2',
        beta_schedule='sigmoid',
        schedule_fn_kwargs=dict(),
        ddim_sampling_eta=0.,
        min_snr_loss_weight=False,
        min_snr_gamma=5,
        standard_fixed_ratio=0.01,
        coeff_ratio=0.1,
    ):
        super().__init__()
        if not is_wavelet:
            raise NotImplementedError('wdno_amd implements the wavelet parametrisation only (is_wavelet=True)')
        if loss_type != 'l2':
            raise NotImplementedError("only loss_type='l2' is used on the WDNO path (train_2d.py:119)")
        self.model = model
        self.loss_layer_weight = loss_layer_weight
        self.is_condition_control = is_condition_control
        self.is_condition_pad = is_condition_pad
        self.channels = self.model.channels
        self.self_condition = self.model.self_condition
        self.image_size = image_size
        self.frames = frames
        self.is_wavelet = is_wavelet
        self.is_super_model = is_super_model
        self.wave_type = wave_type
        self.pad_mode = pad_mode
        self.padded_shape = padded_shape
        self.ori_shape = ori_shape
        self.standard_fixed_ratio = standard_fixed_ratio
        self.coeff_ratio = coeff_ratio
        self.loss_type = loss_type

        if beta_schedule == 'linear':
            fn = linear_beta_schedule
        elif beta_schedule == 'cosine':
            fn = cosine_beta_schedule
        elif beta_schedule == 'sigmoid':
            fn = sigmoid_beta_schedule
        else:
            raise ValueError(f'unknown beta schedule {beta_schedule}')
        betas = fn(timesteps, **schedule_fn_kwargs)
        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)
        self.sampling_timesteps = default(sampling_timesteps, timesteps)
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta

        def loss_weight(snr):
            clipped = snr.clone()
            if min_snr_loss_weight:
                clipped.clamp_(max=min_snr_gamma)
            return clipped / snr
        K.register_schedule(self, betas, loss_weight)
        self._lw_cache = None
        self.use_graph = None       # None: HIP-graph replay of the unguided sampling step when the loop is long enough (WDNO_SAMPLE_GRAPH)

    # ------------------------------------------------------------------ helpers
    def _coef_shape(self, shape, N_upsample=None):
        b, f, c, h, w = shape
        if not self.is_super_model:
            return self.padded_shape
        if N_upsample is None:      # training: infer the level from the tensor size (diffusion_2d.py:990-996)
            N_upsample = int(math.log2(40 / w)) if self.is_condition_control else int(math.log2(24 / f))
        ps = self.padded_shape[N_upsample]
        if self.is_condition_control:
            return [ps[0], ps[1] + 2, ps[2] + 2]
        return [ps[0] + 2, ps[1], ps[2]]

    def _desc(self, shape, coef_shape):
        return K.cond_desc(0, shape, coef_shape, self.is_condition_pad, self.is_condition_control, 0, 0, self.is_super_model)

    def _loss_weights(self, b, c, device):
        lw = self.loss_layer_weight
        key = (id(lw), c, b, str(device))
        if self._lw_cache is None or self._lw_cache[0] != key:
            mean = float(torch.as_tensor(lw, dtype=torch.float32).mean()) if lw is not None else 1.0
            wc = torch.full((c,), mean, device=device, dtype=torch.float32)
            wb = torch.ones((b,), device=device, dtype=torch.float32)
            self._lw_cache = (key, wc, wb)
        return self._lw_cache[1], self._lw_cache[2]

    # ------------------------------------------------------------------ closed-form pieces (API parity; thin torch glue)
    def predict_start_from_noise(self, x_t, t, noise):
        return extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise

    def predict_noise_from_start(self, x_t, t, x0):
        return (extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - x0) / extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape)

    def predict_v(self, x_start, t, noise):
        return extract(self.sqrt_alphas_cumprod, t, x_start.shape) * noise - extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * x_start

    def predict_start_from_v(self, x_t, t, v):
        return extract(self.sqrt_alphas_cumprod, t, x_t.shape) * x_t - extract(self.sqrt_one_minus_alphas_cumprod, t, x_t.shape) * v

    def q_posterior(self, x_start, x_t, t):
        mean = extract(self.posterior_mean_coef1, t, x_t.shape) * x_start + extract(self.posterior_mean_coef2, t, x_t.shape) * x_t
        return mean, extract(self.posterior_variance, t, x_t.shape), extract(self.posterior_log_variance_clipped, t, x_t.shape)

    @torch.no_grad()
    def interpolate(self, x1, x2, t=None, lam=0.5):
        """diffusion_2d.py:950-968: noise both ends to step t, blend, denoise back (no conditioning is imposed, as in the reference)."""
        b, device = x1.shape[0], x1.device
        t = default(t, self.num_timesteps - 1)
        assert x1.shape == x2.shape
        t_batched = torch.full((b,), t, device=device, dtype=torch.long)
        xt1, xt2 = (self.q_sample(v, t=t_batched) for v in (x1, x2))
        img = (1 - lam) * xt1 + lam * xt2
        x_start = None
        for i in reversed(range(0, t)):
            self_cond = x_start if self.self_condition else None
            img, x_start = self.p_sample(tuple(img.shape), img.contiguous(), i, self_cond)
        return img

    @property
    def _ac_host(self):
        return K.ac_host(self)          # host copy of the current alphas_cumprod buffer (scalar DDIM coefficients)

    def sample_noise(self, shape, device):
        return torch.randn(shape, device=device)

    # ------------------------------------------------------------------ sampling
    def model_predictions(self, shape, x, t, x_self_cond=None, clip_x_start=False, rederive_pred_noise=False, design_fn=None,
                          design_guidance='standard', low=None, init=None, init_u=None):
        pred_noise = self.model(x, t, x_self_cond)
        maybe_clip = (lambda v: v.clamp(-1., 1.)) if clip_x_start else identity
        x_start = maybe_clip(self.predict_start_from_noise(x, t, pred_noise))
        if design_fn is not None:       # guidance hook (inference_2d.py:30-66): user callback under autograd
            if getattr(design_fn, 'graph_safe', False):       # closed-form gradient (smoke/guidance.py GuidanceFn): no tape needed
                g = design_fn(x_start, low=low, init=init, init_u=init_u)
            else:
                with torch.enable_grad():
                    x_clone = x_start.clone().detach().requires_grad_()
                    g = design_fn(x_clone, low=low, init=init, init_u=init_u)
            if design_guidance == 'standard':
                grad_final = self.standard_fixed_ratio * g
            elif design_guidance == 'standard-alpha':
                grad_final = extract(self.coeff_ratio * self.betas.clone().flip(0), t, x.shape) * g
            else:
                raise ValueError(design_guidance)
            pred_noise = pred_noise + grad_final
            x_start = maybe_clip(self.predict_start_from_noise(x, t, pred_noise))
        if clip_x_start and rederive_pred_noise:
            pred_noise = self.predict_noise_from_start(x, t, x_start)
        return ModelPrediction(pred_noise, x_start)

    def p_mean_variance(self, shape, x, t, x_self_cond=None, clip_denoised=True, design_fn=None, design_guidance='standard', low=None, init=None,
                        init_u=None):
        preds = self.model_predictions(shape, x, t, x_self_cond, design_fn=design_fn, design_guidance=design_guidance, low=low, init=init, init_u=init_u)
        x_start = preds.pred_x_start
        if clip_denoised:
            x_start = x_start.clamp(-1., 1.)
        mean, var, logvar = self.q_posterior(x_start=x_start, x_t=x, t=t)
        return mean, var, logvar, x_start

    @torch.no_grad()
    def p_sample(self, shape, x, t: int, x_self_cond=None, clip_denoised=True, design_fn=None, design_guidance='standard',
                 low=None, init=None, init_u=None):
        b, device = x.shape[0], x.device
        bt = torch.full((b,), t, device=device, dtype=torch.long)
        noise = self.sample_noise(tuple(x.shape), device) if t > 0 else None
        if design_fn is None:
            eps = self.model(x, bt, x_self_cond)
            return K.p_sample_update(self, x, eps, noise, bt, clamp=clip_denoised)
        mean, _, logvar, x_start = self.p_mean_variance(shape, x=x, t=bt, x_self_cond=x_self_cond, clip_denoised=clip_denoised,
                                                        design_fn=design_fn, design_guidance=design_guidance, low=low, init=init, init_u=init_u)
        pred = mean if noise is None else mean + (0.5 * logvar).exp() * noise
        return pred, x_start

    def _condition_source(self, shape, device, init, control, low):
        """Clean values for every conditioned position, assembled once per sampling call."""
        src = torch.zeros(shape, device=device, dtype=torch.float32)
        assert init is not None
        src[:, :, -2] = init.to(device)
        if self.is_condition_control:
            src[:, :, 24:40] = control.to(device)
        if self.is_super_model:
            src[:, :, 40:80] = low.to(device)
        return src

    @torch.no_grad()
    def p_sample_loop(self, shape, N_upsample=0, design_fn=None, design_guidance='standard', return_all_timesteps=None, init=None,
                      init_u=None, control=None, low=None, device=None):
        device = self.betas.device
        shape = tuple(shape)
        desc = self._desc(shape, self._coef_shape(shape, N_upsample))
        src = self._condition_source(shape, device, init, control, low)
        x = K.apply_cond(self.sample_noise(list(shape), device).contiguous(), src, desc)
        if design_fn is None and not self.self_condition:          # unguided: fused launches, the step replayed from one HIP graph
            return K.sampling_loop(self, x, src, desc, cond_first=False, use_graph=self.use_graph)
        if getattr(design_fn, 'graph_safe', False) and not self.self_condition:
            return K.guided_sampling_loop_smoke(self, x, src, desc, design_fn, design_guidance, low=low, init=init, init_u=init_u, use_graph=self.use_graph)
        x_start = None
        for t in reversed(range(0, self.num_timesteps)):
            self_cond = x_start if self.self_condition else None
            x, x_start = self.p_sample(shape, x, t, self_cond, design_fn=design_fn, design_guidance=design_guidance, low=low, init=init, init_u=init_u)
            x = K.apply_cond(x.contiguous(), src, desc)
        return x

    @torch.no_grad()
    def ddim_sample(self, shape, N_upsample=0, design_fn=None, design_guidance='standard', init=None, init_u=None, control=None,
                    low=None, device=None):
        device, eta = self.betas.device, self.ddim_sampling_eta
        shape = tuple(shape)
        batch = shape[0]
        desc = self._desc(shape, self._coef_shape(shape, N_upsample))
        src = self._condition_source(shape, device, init, control, low)
        img = K.apply_cond(self.sample_noise(shape, device).contiguous(), src, desc)
        pairs = K.ddim_time_pairs(self.num_timesteps, self.sampling_timesteps)
        if design_fn is None and not self.self_condition:
            return K.sampling_loop(self, img, src, desc, ddim_pairs=pairs, eta=eta, cond_first=False, use_graph=self.use_graph)
        if getattr(design_fn, 'graph_safe', False) and not self.self_condition:
            return K.guided_sampling_loop_smoke(self, img, src, desc, design_fn, design_guidance, ddim_pairs=pairs, eta=eta, low=low, init=init,
                                                init_u=init_u, use_graph=self.use_graph)
        x_start = None
        for time, time_next in pairs:
            tc = torch.full((batch,), time, device=device, dtype=torch.long)
            self_cond = x_start if self.self_condition else None
            pred_noise, x_start = self.model_predictions(shape, img, tc, self_cond, clip_x_start=True, rederive_pred_noise=True,
                                                         design_fn=design_fn, design_guidance=design_guidance, init=init, init_u=init_u, low=low)
            if time_next < 0:
                img = x_start
                continue
            sigma, c, sqrt_an = K.ddim_coefficients(self._ac_host, time, time_next, eta)
            img = x_start * sqrt_an + c * pred_noise + sigma * self.sample_noise(shape, device)
            img = K.apply_cond(img.contiguous(), src, desc)
        return img

    @torch.no_grad()
    def sample(self, batch_size=16, N_upsample=0, design_fn=None, design_guidance='standard', init=None, init_u=None, control=None,
               low=None, device=None):
        assert batch_size == init.shape[0]
        if not self.is_super_model:
            size = (batch_size, self.frames, self.channels, self.image_size, self.image_size)
        else:
            size = (batch_size, low.shape[1], self.channels, low.shape[-2], low.shape[-1])
        fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        return fn(size, N_upsample, design_fn, design_guidance, init=init, init_u=init_u, control=control, low=low, device=device)

    # ------------------------------------------------------------------ training
    def q_sample(self, x_start, t, noise=None):
        noise = default(noise, lambda: torch.randn_like(x_start))
        x, _ = K.q_sample_cond(x_start, noise, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, K.plain_desc(x_start))
        return x

    def p_losses(self, state_start, t, noise=None):
        b, f, c, h, w = state_start.shape
        noise = default(noise, lambda: self.sample_noise(tuple(state_start.shape), state_start.device))
        desc = self._desc(tuple(state_start.shape), self._coef_shape(state_start.shape))
        state, target = K.q_sample_cond(state_start, noise, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, desc)
        model_out = self.model(state, t, None)
        wc, wb = self._loss_weights(b, c, state.device)
        # reference: mse(mean) * loss_layer_weight[1,1,C,1,1] -> .mean()  ==  mse * mean(loss_layer_weight)
        return K.weighted_mse(model_out, target, wc, wb, c, h * w)

    def forward(self, state, *args, **kwargs):
        b, device = state.shape[0], state.device
        t = torch.randint(0, self.num_timesteps, (b,), device=device).long()
        return self.p_losses(state, t, *args, **kwargs)


# ======================================================================================================================
# Trainer -- drop-in for smoke/ddpm/diffusion_2d.py:1060-1307
# ======================================================================================================================
def has_int_squareroot(num):
    return (math.sqrt(num) ** 2) == num


def cycle(dl):
    while True:
        for data in dl:
            yield data


class Trainer(_TrainerCore):
    """Same constructor keywords, attributes (`model`, `opt`, `ema`, `step`, `results_path`, `ds`, `device`), checkpoint
    naming (`model-{milestone}.pt`) and checkpoint keys (`step / model / opt / ema / scaler`) as the reference Trainer.
    The step itself is wdno_amd.trainer's flat-buffer path (one RCCL all-reduce, clip + Adam in two launches);
    `accelerate` is replaced by one process per GPU under torchrun.

    `dataset` is the string "Smoke" like in train_2d.py:123-126 (the dataset class is then taken from `ddpm.data_2d`,
    i.e. from whichever tree provides it), or -- an extension -- any map-style dataset object yielding
    `(state [F, C, H, W], shape, ori_shape, sim_id)` tuples."""

    mixed_precision_type = None       # None: fp32-equivalent steps; 'bf16': BASELINE configs[1]'s single-product mode for this Trainer's steps
                                      # (wdno_amd.trainer.conv_math_of). A class attribute: the constructor keeps the reference's parameter list,
                                      # whose only precision keyword, fp16=True, selects accelerate's fp16 + GradScaler mode (refused, see there)

    def __init__(
        self,
        diffusion_model,
        dataset,
        dataset_path,
        *,
        N_downsample=0,
        train_batch_size=16,
        gradient_accumulate_every=1,
        augment_horizontal_flip=True,
        train_lr=1e-4,
        train_num_steps=100000,
        ema_update_every=10,
        ema_decay=0.995,
        adam_betas=(0.9, 0.99),
        save_and_sample_every=1000,
        num_samples=25,
        results_path='./results',
        amp=False,
        fp16=False,
        split_batches=True,
        convert_image_to=None,
        calculate_fid=True,
        inception_block_idx=2048,
        is_schedule=True,
        resume=False,
        resume_step=0,
    ):
        # diffusion_2d.py:1093-1098: Accelerator(mixed_precision='fp16' if fp16 else 'no'); `amp` only sets accelerator.native_amp (no effect on
        # the arithmetic). The reference has no bf16 keyword here: the class attribute `mixed_precision_type` (None / 'bf16') selects the
        # single-product mode for this Trainer, like `use_graph` / `num_workers` without touching the constructor's parameter list.
        assert has_int_squareroot(num_samples), 'number of samples must have an integer square root'
        schedule = (lambda base, step: _multistep_lr(base, step, (50000, 150000, 300000), 0.1)) if is_schedule else (lambda base, step: base)
        super().__init__(diffusion_model, train_batch_size=train_batch_size, gradient_accumulate_every=gradient_accumulate_every,
                         train_lr=train_lr, train_num_steps=train_num_steps, ema_update_every=ema_update_every, ema_decay=ema_decay,
                         adam_betas=adam_betas, save_and_sample_every=save_and_sample_every, split_batches=split_batches,
                         max_grad_norm=1.0, results_dir=results_path, lr_schedule=schedule,
                         mixed_precision='fp16' if fp16 else (self.mixed_precision_type or 'no'))
        self.dataset = dataset
        self.num_samples = num_samples
        self.image_size = diffusion_model.image_size
        self.resume, self.resume_step = resume, resume_step
        from pathlib import Path
        self.results_path = Path(results_path)
        if isinstance(dataset, str):
            if dataset != 'Smoke':
                raise AssertionError(dataset)
            from ddpm.data_2d import Smoke_wave, SuperDataLoader                       # diffusion_2d.py:1117-1145
            if not self.model.is_wavelet:
                raise NotImplementedError('only the wavelet parametrisation is on the WDNO path')
            if not self.model.is_super_model:
                self.ds = Smoke_wave(dataset_path, self.model.wave_type, self.model.pad_mode, is_super_model=False, N_downsample=0)
            else:
                self.ds = [Smoke_wave(dataset_path, self.model.wave_type, self.model.pad_mode, is_super_model=True,
                                      downsample_type='space' if self.model.is_condition_control else 'time', N_downsample=i)
                           for i in range(N_downsample)]
        else:
            self.ds = dataset
        if isinstance(self.ds, (list, tuple)):
            from ddpm.data_2d import SuperDataLoader
            workers = 4 if self.num_workers is None else self.num_workers
            dl = SuperDataLoader(self.ds, batch_size=self.local_batch_size, shuffle=True, pin_memory=True, num_workers=workers,
                                 seed=self.data_seed if self.world > 1 else None)
        else:
            workers = 16 if self.num_workers is None else self.num_workers
            dl = self.make_loader(self.ds, self.local_batch_size, workers)
        self.dl = self.cycle(dl)          # advances DistributedSampler epochs

    def save(self, milestone):
        if not self.is_main_process:
            return
        torch.save(self.checkpoint_dict(), str(self.results_path / f'model-{milestone}.pt'))

    def load(self, milestone):
        path = str(self.results_path / f'model-{milestone}.pt')
        print('model path: ', path)
        data = torch.load(path, map_location=self.device, weights_only=False)
        self.load_checkpoint_dict(data)
        print('model loaded: ', path)

    def _next_state(self):
        state = next(self.dl)[0]                                   # (state, shape, ori_shape, sim_id)
        return state.to(self.device, non_blocking=True)

    def train(self):
        import logging
        import os
        if self.is_main_process:          # one writer: under torchrun every rank runs this method
            logging.basicConfig(filename=os.path.join(self.results_path, 'info.log'), level=logging.INFO,
                                format='%(asctime)s - %(levelname)s - %(message)s')
        while self.step < self.train_num_steps:
            lr = self.lr_schedule(self.train_lr, self.step)
            total_loss = self.optimisation_step(self._next_state)
            if self.step != 0 and self.step % 10 == 0 and self.is_main_process:
                logging.info(f'step: {self.step}, loss: {total_loss:.4f}, LR: {lr}')
            self.step += 1
            if self.is_main_process:
                self.ema.update()
                if self.step != 0 and self.step % self.save_and_sample_every == 0:
                    self.ema.ema_model.eval()
                    self.save(self.step // self.save_and_sample_every)
        if self.is_main_process:
            print('training complete')


def __getattr__(name):          # names that are not on the WDNO path (e.g. the 2-D `Unet`) come from the reference module, if present
    return _reference_getattr(name)
