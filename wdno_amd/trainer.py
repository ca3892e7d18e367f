"""Training-step machinery: flat parameter / gradient buffers, fused clip + Adam, EMA and the data-parallel all-reduce.

Counterpart of the hot loop of both reference Trainers (burgers/ddpm_burgers/train_diffusion.py:196-228,
smoke/ddpm/diffusion_2d.py:1268-1305), whose per-step work is
    loss = model(batch); backward; [DDP all-reduce(mean)]; clip_grad_norm_(1.0); Adam; lr-schedule; EMA every 10.
MI355X-first design:
  * all parameters live in ONE flat fp32 buffer and all gradients in another (nn.Parameters are views), so the
    gradient exchange is a single RCCL all-reduce over xGMI (95 MB smoke / 563 MB Burgers) instead of DDP's 25 MB
    buckets, and clip + Adam are two launches over the whole model (wdno_sumsq, wdno_adam_clip_step) with the clip
    coefficient computed on the device (no host sync);
  * one process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm; gloo is used by the CPU tests, which
    exercise exactly this exchange on host buffers.
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _lib, ops
from .ops import _lib_, _p, _stream, _ws


class FlatBuffers:
    """Re-homes every trainable parameter (and its .grad) of `params` as views into two flat fp32 tensors."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat_param = torch.empty(self.numel, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + n].view(p.shape)
                p.grad = self.flat_grad[off:off + n].view(p.shape)
                off += n

    def zero_grad(self):
        self.flat_grad.zero_()
        for p, (o, n) in zip(self.params, self._spans()):      # autograd may have replaced .grad with a fresh tensor
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + n].view(p.shape)

    def _spans(self):
        off = 0
        for p in self.params:
            yield off, p.numel()
            off += p.numel()

    def gather_grads(self):
        """Make sure every .grad is the flat view (autograd accumulates in place when .grad already exists)."""
        for p, (o, n) in zip(self.params, self._spans()):
            view = self.flat_grad[o:o + n].view(p.shape)
            if p.grad is None:
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view


def allreduce_mean_(flat, world_size=None, group=None):
    """DDP semantics on one flat buffer: sum over ranks, then divide by the world size (in place). Works for CUDA
    tensors over RCCL and for CPU tensors over gloo."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    ws = world_size or dist.get_world_size(group)
    if ws == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class FlatAdam:
    """clip_grad_norm_(max_grad_norm) + torch.optim.Adam(lr, betas, eps) over flat buffers; two HIP launches per step."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, max_grad_norm=1.0):
        self.buf = FlatBuffers(list(params))
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, betas, eps, max_grad_norm
        dev = self.buf.flat_param.device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdam runs on the GPU only (no CPU fallback)')
        self.exp_avg = torch.zeros_like(self.buf.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.buf.flat_param)
        self.step_count = 0
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        lib = _lib_()
        self._ws_bytes = lib.wdno_sumsq_ws_bytes(self.buf.numel)
        self._ws = _ws(self._ws_bytes, dev)

    @property
    def param_groups(self):       # minimal torch.optim surface used by LR schedulers / logging
        return [{'lr': self.lr, 'params': self.buf.params}]

    def zero_grad(self, set_to_none=False):
        self.buf.zero_grad()

    def step(self, lr=None, grad_scale=1.0):
        """Returns the (pre-clip) global gradient norm as a 0-d device tensor (no host sync)."""
        self.buf.gather_grads()
        lib = _lib_()
        n = self.buf.numel
        self.step_count += 1
        _lib.check(lib.wdno_sumsq(_p(self.buf.flat_grad), n, _p(self.sumsq), _p(self._ws), self._ws_bytes, _stream()), 'sumsq')
        _lib.check(lib.wdno_adam_clip_step(_p(self.buf.flat_param), _p(self.buf.flat_grad), _p(self.exp_avg), _p(self.exp_avg_sq), n,
                                           _p(self.sumsq), float(self.max_grad_norm or 0.0), float(grad_scale),
                                           float(self.lr if lr is None else lr), float(self.betas[0]), float(self.betas[1]),
                                           float(self.eps), int(self.step_count), _stream()), 'adam_clip_step')
        ops.bump_weight_epoch()
        return self.sumsq.sqrt()[0] * grad_scale

    def state_dict(self):
        return {'step': self.step_count, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'lr': self.lr}

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.lr = sd.get('lr', self.lr)


class FlatEMA:
    """ema = ema * beta + p * (1 - beta) over a flat copy of the parameters, every `update_every` calls
    (ema_pytorch.EMA(beta=0.995, update_every=10) as configured at train_diffusion.py:123-125)."""

    def __init__(self, flat_param, beta=0.995, update_every=10):
        self.flat = flat_param.clone()
        self.beta, self.update_every, self.calls = beta, update_every, 0

    def update(self, flat_param):
        self.calls += 1
        if self.calls % self.update_every:
            return
        _lib.check(_lib_().wdno_ema_update(_p(self.flat), _p(flat_param), flat_param.numel(), float(self.beta), _stream()), 'ema_update')


def cosine_annealing_lr(base_lr, step, t_max, eta_min=0.0):
    """torch.optim.lr_scheduler.CosineAnnealingLR closed form (train_diffusion.py:118)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * step / t_max)) / 2


def multistep_lr(base_lr, step, milestones=(50000, 150000, 300000), gamma=0.1):
    """MultiStepLR (diffusion_2d.py:1159-1161)."""
    return base_lr * gamma ** sum(1 for m in milestones if step >= m)


class TrainStep:
    """One data-parallel optimisation step of a GaussianDiffusion module.

    step(batch) = loss -> backward -> all-reduce(mean) of the flat gradient -> clip -> Adam -> (EMA on rank 0)."""

    def __init__(self, diffusion, lr, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=None, ema_decay=0.995, ema_update_every=10,
                 use_ema=True, group=None):
        self.model = diffusion
        self.opt = FlatAdam(diffusion.parameters(), lr=lr, betas=betas, max_grad_norm=max_grad_norm)
        self.base_lr = lr
        self.lr_schedule = lr_schedule or (lambda base, step: base)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.ema = FlatEMA(self.opt.buf.flat_param, ema_decay, ema_update_every) if (use_ema and self.rank == 0) else None
        self.step_idx = 0

    def step(self, batch, **loss_kwargs):
        self.opt.zero_grad()
        loss = self.model(batch, **loss_kwargs)
        loss.backward()
        self.opt.buf.gather_grads()
        if self.world > 1:
            allreduce_mean_(self.opt.buf.flat_grad, self.world, self.group)
        lr = self.lr_schedule(self.base_lr, self.step_idx)
        gnorm = self.opt.step(lr=lr, grad_scale=1.0 / self.world)
        self.step_idx += 1
        if self.ema is not None:
            self.ema.update(self.opt.buf.flat_param)
        return loss.detach(), gnorm

    def step_with(self, x0, t, noise):
        """Same as step() with injected timestep / noise (parity tests)."""
        self.opt.zero_grad()
        loss = self.model.p_losses(x0, t, noise=noise)
        loss.backward()
        self.opt.buf.gather_grads()
        if self.world > 1:
            allreduce_mean_(self.opt.buf.flat_grad, self.world, self.group)
        lr = self.lr_schedule(self.base_lr, self.step_idx)
        gnorm = self.opt.step(lr=lr, grad_scale=1.0 / self.world)
        self.step_idx += 1
        if self.ema is not None:
            self.ema.update(self.opt.buf.flat_param)
        return loss.detach(), gnorm

    # checkpoint dictionary with the reference's top-level keys (train_diffusion.py:148-185, diffusion_2d.py:1187-1230)
    def state_dict(self):
        return {'step': self.step_idx, 'model': self.model.state_dict(), 'opt': self.opt.state_dict(),
                'ema': None if self.ema is None else {'flat': self.ema.flat, 'calls': self.ema.calls}, 'scaler': None}

    def load_state_dict(self, sd):
        self.step_idx = int(sd['step'])
        self.model.load_state_dict(sd['model'])
        self.opt.load_state_dict(sd['opt'])
        if self.ema is not None and sd.get('ema') is not None:
            self.ema.flat.copy_(sd['ema']['flat'])
            self.ema.calls = int(sd['ema']['calls'])
        ops.bump_weight_epoch()
