"""Training-step machinery: flat parameter / gradient buffers, fused clip + Adam, EMA and the data-parallel all-reduce.

Counterpart of the hot loop of both reference Trainers (burgers/ddpm_burgers/train_diffusion.py:196-228,
smoke/ddpm/diffusion_2d.py:1268-1305), whose per-step work is
    loss = model(batch); backward; [DDP all-reduce(mean)]; clip_grad_norm_(1.0); Adam; lr-schedule; EMA every 10.
MI355X-first design:
  * all parameters live in ONE flat fp32 buffer and all gradients in another (nn.Parameters are views), so the
    gradient exchange is a single RCCL all-reduce over xGMI (95 MB smoke / 563 MB Burgers) instead of DDP's 25 MB
    buckets (optionally four contiguous spans, each started as soon as backward has filled it: OverlappedAllReduce,
    WDNO_DP_OVERLAP=1), and clip + Adam are two launches over the whole model (wdno_sumsq, wdno_adam_clip_step) with the clip
    coefficient computed on the device (no host sync);
  * one process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm; gloo is used by the CPU tests, which
    exercise exactly this exchange on host buffers.
"""
import ctypes as C
import math

import os

import torch
import torch.distributed as dist

from . import _lib, ops
from .ops import _lib_, _p, _stream, _ws


GRAD_GATHER = True        # gradients are stored by autograd and gathered into the flat buffer with one launch (test knob; False = round-1 in-place accumulation)


class _CopyItem(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('n', C.c_int64)]


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
        ops.drop_weight_caches()          # operands packed from the old storages would pin them (and can never hit again)
        self._untouched = None
        self.span_list = list(self._spans())
        self._gather_srcs, self._gather_table, self._gathered = None, None, False
        self._views = [self.flat_grad[o:o + n].view(p.shape) for p, (o, n) in zip(self.params, self.span_list)]
        self._view_ptrs = [v.data_ptr() for v in self._views]
        self.epoch_cell = [0]                             # ops._param_epoch: bumped by params_changed(), seen through every parameter of THIS buffer
        for prm, v in zip(self.params, self._views):      # ops._flat_grad_out: weight-gradient kernels write their span of flat_grad directly
            prm._wdno_flat_grad, prm._wdno_flat_busy = v, False
            prm._wdno_epoch_cell = self.epoch_cell

    def params_changed(self):
        """Call after ANY write to flat_param (optimiser step, broadcast, checkpoint load, EMA copy): the packed / split weight
        operands the convolutions cache are keyed on this epoch, parameter version counters do not see writes through the flat
        buffer. Every writer in this module goes through here."""
        self.epoch_cell[0] += 1
        ops.bump_weight_epoch(params=False)

    def watch_gradient_coverage(self):
        """Arms a one-step check that every parameter receives a gradient. The flat optimiser keeps zero-filled gradients
        (never None), so a parameter that backward does not reach still has its moments decayed and takes a momentum-driven
        update, whereas torch.optim.Adam after zero_grad(set_to_none=True) -- the reference -- skips it. Every trainable
        parameter of the WDNO models is reached in every step; anything else is refused rather than silently diverging."""
        self._untouched = set(range(len(self.params)))
        handles = []

        def make(i):
            def hook(_):
                if self._untouched is not None:
                    self._untouched.discard(i)
            return hook
        for i, p in enumerate(self.params):
            handles.append(p.register_post_accumulate_grad_hook(make(i)))
        self._watch_handles = handles

    def check_gradient_coverage(self):
        if self._untouched is None:
            return
        missing, self._untouched = sorted(self._untouched), None
        for h in self._watch_handles:
            h.remove()
        if missing:
            shapes = [tuple(self.params[i].shape) for i in missing[:8]]
            raise RuntimeError(f'wdno_amd FlatAdam: {len(missing)} trainable parameter(s) received no gradient in the first step '
                               f'(indices {missing[:8]}, shapes {shapes}); the flat optimiser would still update them, unlike '
                               'torch.optim.Adam with set_to_none gradients. Freeze them (requires_grad_(False)) before building the trainer.')

    def zero_grad(self):
        """Gradients are left as None: autograd then STORES each parameter's gradient instead of adding it into an existing
        tensor (AccumulateGrad issues one `add` launch per parameter when .grad exists: ~230 launches, 3.9 % of the smoke step),
        and gather_grads() moves all of them into the flat buffer with one launch."""
        self._gathered = False
        if not GRAD_GATHER:                  # A/B: the round-1 behaviour (zero-filled flat views, autograd adds into them)
            self.flat_grad.zero_()
            for p, v in zip(self.params, self._views):
                p.grad = v
            return
        for p in self.params:
            p.grad = None
            p._wdno_flat_busy = False

    def _spans(self):
        off = 0
        for p in self.params:
            yield off, p.numel()
            off += p.numel()

    def prepare_capture(self):
        """Pinned + device pointer tables of ONE upcoming graph capture (pinned host memory cannot be allocated while a stream is
        capturing); they stay alive, unmodified, as long as this object does: the captured upload re-reads the pinned buffer at
        every replay."""
        n = len(self.params)
        host = torch.empty((n, 3), dtype=torch.int64).pin_memory()
        table = torch.empty((n, 3), dtype=torch.int64, device=self.flat_grad.device)
        self._capture_tables = [(host, table)]        # the CapturedStep that uses it takes its own reference (discarded captures free theirs)

    def gather_grads(self, capture=False):
        """After backward: every p.grad -> its span of flat_grad (zeros where backward produced none), then p.grad becomes the
        flat view. One wdno_gather_items launch; the pointer table is re-uploaded only when an address changed (the caching
        allocator hands out the same blocks step after step). capture=True (inside a HIP-graph capture of the step): the table upload
        is part of the graph, from a pinned buffer that belongs to this capture alone and is never rewritten."""
        if not self.flat_grad.is_cuda:              # host buffers exist only in the gloo exchange tests (tests/test_distributed_cpu.py)
            for p, (o, n) in zip(self.params, self.span_list):
                view = self.flat_grad[o:o + n].view(p.shape)
                if p.grad is None:
                    view.zero_()
                elif p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
                p.grad = view
            return
        if self._gathered:
            return
        grads = [p.grad for p in self.params]
        ptrs = [0 if g is None else g.data_ptr() for g in grads]
        deferred = ops.take_deferred_dsts()           # spans that deferred split reductions wrote (ops.flush_wgrad_reduces) after autograd took the gradient
        if deferred:
            lost = [i for i, (a, v) in enumerate(zip(ptrs, self._view_ptrs)) if v in deferred and a != v]
            if lost:
                raise RuntimeError(f'wdno_amd FlatBuffers: autograd did not keep the flat-buffer view of {len(lost)} weight gradient(s) whose split reduction was '
                                   f'deferred (parameter indices {lost[:8]}): it holds a copy made before the reduction ran. Set ops.DEFER_WGRAD_REDUCE = False.')
        if ptrs == self._view_ptrs:                  # accumulated in place into the flat views (WDNO_GRAD_GATHER=0 / a second call)
            self._gathered = True
            return
        for i, g in enumerate(grads):
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                grads[i] = g = g.to(torch.float32).contiguous()
                ptrs[i] = g.data_ptr()
        if ptrs != self._gather_srcs:
            # (src, dst, n) rows = wdno_copy_item. The upload must not block the host (a pageable-memory copy waits for the stream and
            # ends the host's run-ahead: +2 ms per step): pinned staging buffers, rotated so that a buffer is not rewritten while an
            # earlier asynchronous copy of it may still be pending.
            if capture:
                host, table = self._capture_tables[-1]        # allocated by prepare_capture(): no host allocation inside a capture
                host[:, 0] = torch.tensor(ptrs, dtype=torch.int64)
                host[:, 1] = torch.tensor(self._view_ptrs, dtype=torch.int64)
                host[:, 2] = torch.tensor([sp[1] for sp in self.span_list], dtype=torch.int64)
                table.copy_(host, non_blocking=True)
                _lib.check(_lib_().wdno_gather_items(_p(table), len(ptrs), 48, _stream()), 'gather_items')
                for p, v in zip(self.params, self._views):
                    p.grad = v
                self._gathered = True
                self._gather_srcs = None                 # eager steps after a capture rebuild their own table
                return
            if self._gather_table is None:
                n = len(ptrs)
                self._gather_table = torch.empty((n, 3), dtype=torch.int64, device=self.flat_grad.device)
                self._gather_hosts = [torch.empty((n, 3), dtype=torch.int64).pin_memory() for _ in range(8)]
                for h in self._gather_hosts:
                    h[:, 1] = torch.tensor(self._view_ptrs, dtype=torch.int64)
                    h[:, 2] = torch.tensor([sp[1] for sp in self.span_list], dtype=torch.int64)
                self._gather_events = [None] * len(self._gather_hosts)
                self._gather_turn = 0
            slot = self._gather_turn % len(self._gather_hosts)
            h = self._gather_hosts[slot]
            self._gather_turn += 1
            if self._gather_events[slot] is not None:      # the asynchronous copy that last read this pinned buffer must have finished
                self._gather_events[slot].synchronize()    # (a no-op unless the host ran eight uploads ahead of the device)
            h[:, 0] = torch.tensor(ptrs, dtype=torch.int64)
            self._gather_table.copy_(h, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._gather_events[slot] = ev
            self._gather_srcs = ptrs
        _lib.check(_lib_().wdno_gather_items(_p(self._gather_table), len(ptrs), 48, _stream()), 'gather_items')
        for p, v in zip(self.params, self._views):
            p.grad = v
        self._gathered = True


def allreduce_sum_(flat, world_size=None, group=None, force=False):
    """SUM over ranks, in place (no-op without a process group). The 1/world factor of DDP's mean is NOT applied here: the
    callers fold it into the clip + Adam launch (FlatAdam.step(grad_scale=1/world)). Works for CUDA tensors over RCCL and, in
    the tests, for CPU / CUDA tensors over gloo."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    ws = world_size or dist.get_world_size(group)
    if ws == 1 and not force:          # force: a one-rank group still goes through the collective (first-run test of the RCCL path on one GPU)
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class OverlappedAllReduce:
    """The gradient all-reduce of a data-parallel step, started bucket by bucket while backward is still running.

    The flat gradient buffer is cut at parameter boundaries into `n_buckets` contiguous spans of about equal size. Every
    parameter carries a post-accumulate hook; when the last parameter of a bucket has received its gradient the bucket's
    sum all-reduce is launched asynchronously (RCCL runs it on its own stream, ordered after the kernels that wrote the
    span), so only the bucket that finishes last -- the first layers of the network -- is exposed after backward. All ranks
    run the same graph, hence launch the buckets in the same order. Buckets whose parameters got no gradient in this
    backward are launched by finish()."""

    def __init__(self, buf, group=None, n_buckets=4):
        self.buf, self.group = buf, group
        spans = list(buf._spans())
        target = max(1, buf.numel // max(1, n_buckets))
        self.bounds, self.bucket_of, self.members = [], [], []
        start, members = 0, []
        for i, (o, n) in enumerate(spans):
            members.append(i)
            self.bucket_of.append(len(self.bounds))
            if o + n - start >= target and len(self.bounds) < n_buckets - 1:
                self.bounds.append((start, o + n)); self.members.append(members)
                start, members = o + n, []
        if members or not self.bounds:
            self.bounds.append((start, buf.numel)); self.members.append(members)
        self.spans = spans
        self.active = False
        self.pending, self.launched, self.handles = [], [], []
        for i, p in enumerate(buf.params):
            p.register_post_accumulate_grad_hook(lambda t, i=i: self._ready(i))

    def begin(self):
        self.pending = [len(m) for m in self.members]
        self.launched = [False] * len(self.bounds)
        self.handles = []
        self.active = True

    def _ready(self, i):
        if not self.active:
            return
        b = self.bucket_of[i]
        self.pending[b] -= 1
        if self.pending[b] == 0 and not self.launched[b]:
            self._launch(b)

    def _launch(self, b):
        ops.flush_wgrad_reduces()                     # weight gradients whose split reductions are still pending are unwritten until this launch
        flat = self.buf.flat_grad
        for i in self.members[b]:                     # autograd adds in place into the flat views; anything else is copied in first
            p, (o, n) = self.buf.params[i], self.spans[i]
            if p.grad is None:                        # no gradient in this backward: its span must not carry last step's values into the sum
                flat[o:o + n].zero_()
            elif p.grad.data_ptr() != flat.data_ptr() + 4 * o:
                flat[o:o + n].view(p.shape).copy_(p.grad)
                p.grad = flat[o:o + n].view(p.shape)
        s, e = self.bounds[b]
        self.launched[b] = True
        if e > s:
            self.handles.append(dist.all_reduce(flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Launch what backward did not complete, then make the current stream wait for every bucket."""
        self.active = False
        for b in range(len(self.bounds)):
            if not self.launched[b]:
                self._launch(b)
        for h in self.handles:
            h.wait()
        self.handles = []


def init_distributed(backend=None):
    """One process per GPU: what `accelerate` did implicitly for the reference Trainers (train_diffusion.py:71-74,
    diffusion_2d.py:1093-1098). Under torchrun (RANK / WORLD_SIZE / LOCAL_RANK in the environment) this selects the rank's GPU
    BEFORE anything is allocated or launched on it (ops._stream() takes the current device's stream) and creates the process
    group -- backend "nccl" is RCCL on ROCm. Returns (rank, world, local_rank); (0, 1, LOCAL_RANK or 0) without torchrun."""
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        if os.environ.get('WDNO_DIST_SHARE_GPU') == '1':      # control-flow tests of the multi-rank paths on a one-GPU box (with WDNO_DIST_BACKEND=gloo)
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if dist.is_available() and not dist.is_initialized() and world_env > 1 and 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend or os.environ.get('WDNO_DIST_BACKEND', 'nccl'), rank=int(os.environ['RANK']), world_size=world_env)
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), local
    return 0, 1, local


def broadcast_parameters(buf, model=None, group=None):
    """Rank 0's parameters (and buffers) to every rank, as DistributedDataParallel does at construction: replicas that were built
    from different seeds would otherwise only share gradients, never weights."""
    forced = os.environ.get('WDNO_DP_FORCE_EXCHANGE', '0') == '1'
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not forced):
        return
    dist.broadcast(buf.flat_param, 0, group=group)
    if model is not None:
        for b in model.buffers():
            if b.is_floating_point() or b.dtype in (torch.int64, torch.int32):
                dist.broadcast(b, 0, group=group)
        for p in model.parameters():
            if not p.requires_grad:           # frozen parameters are not part of the flat buffer
                dist.broadcast(p.data, 0, group=group)
    buf.params_changed()


def cycle_loader(dl):
    """`cycle(dl)` of the reference (model_utils.py / diffusion_2d.py) that also advances the epoch of a DistributedSampler, so
    every pass reshuffles like accelerate's prepared loader does (without set_epoch each epoch replays the same permutation)."""
    epoch = 0
    while True:
        sampler = getattr(dl, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(epoch)
        for data in dl:
            yield data
        epoch += 1


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
        self.buf.check_gradient_coverage()
        lib = _lib_()
        n = self.buf.numel
        self.step_count += 1
        _lib.check(lib.wdno_sumsq(_p(self.buf.flat_grad), n, _p(self.sumsq), _p(self._ws), self._ws_bytes, _stream()), 'sumsq')
        _lib.check(lib.wdno_adam_clip_step(_p(self.buf.flat_param), _p(self.buf.flat_grad), _p(self.exp_avg), _p(self.exp_avg_sq), n,
                                           _p(self.sumsq), float(self.max_grad_norm or 0.0), float(grad_scale),
                                           float(self.lr if lr is None else lr), float(self.betas[0]), float(self.betas[1]),
                                           float(self.eps), int(self.step_count), _stream()), 'adam_clip_step')
        self.buf.params_changed()
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


def conv_math_of(mixed_precision):
    """accelerate's `mixed_precision` keyword -- what the reference Trainers build from `amp` / `mixed_precision_type` (train_diffusion.py:61-62,
    71-74) and `fp16` (diffusion_2d.py:1093-1098) -- -> the convolution arithmetic of wdno_amd.ops for the training step.
      'no'   : the process default (ops.CONV_MATH: the fp32-equivalent 3 x fp16 split; WDNO_CONV_MATH overrides);
      'bf16' : BASELINE configs[1] -- ONE bf16 plane per operand on v_mfma_f32_32x32x16_bf16, fp32 master weights, accumulators, norms, softmax
               and optimiser (what autocast(bfloat16) + fp32 Adam amounts to; tolerances: tests/test_gpu_bf16.py);
      'fp16' : refused. accelerate's fp16 mode is autocast(float16) + a GradScaler whose dynamic loss scale (skipped steps on overflow, growth
               every 2000 steps) is part of the training trajectory; there is no single-plane fp16 kernel family here (one fp16 plane needs the
               per-tensor scale of the split path AND loss scaling for the gradients) -- use 'bf16', which needs neither."""
    mp = 'no' if mixed_precision in (None, False, 'no') else str(mixed_precision)
    if mp == 'no':
        return ops.CONV_MATH
    if mp == 'bf16':
        if not ops.LOWP_AVAILABLE:
            raise RuntimeError('wdno_amd: the single-product bf16 kernels are not in this build')
        return 'bf16'
    if mp == 'fp16':
        raise ValueError("wdno_amd Trainer: mixed_precision 'fp16' (autocast(float16) + GradScaler) is not built on MI355X -- pass "
                         "mixed_precision_type='bf16' (Burgers: Trainer(amp=True, mixed_precision_type='bf16'); smoke: Trainer.mixed_precision_type = 'bf16'): "
                         'one bf16 plane per operand needs no loss scaling')
    raise ValueError(f'wdno_amd Trainer: unknown mixed_precision {mixed_precision!r}')


def is_capture_failure(err):
    """Is `err` a failure of HIP stream capture itself (survivable: the same step runs launch by launch) rather than an error of the step
    being captured? Out-of-memory for the graph's private pool, and RuntimeErrors that name the capture: HIP's
    hipErrorStreamCapture{Unsupported, Invalidated, ...} ("operation not permitted when stream is capturing"), torch's own capture checks."""
    if isinstance(err, torch.cuda.OutOfMemoryError):
        return True
    if not isinstance(err, RuntimeError):
        return False
    msg = str(err).lower()
    return 'captur' in msg or 'hipgraph' in msg or 'cudagraph' in msg


class CapturedStep:
    """loss -> backward -> gradient gather of `diffusion.p_losses` on static inputs, captured once in a HIP graph (ops.graph_capture) and
    replayed per step. Shared by TrainStep.capture and the drop-in Trainers (TrainerCore.use_graph). What a capture needs (each item
    found by a failed capture on the MI355X box): no host allocation / pageable upload while capturing (the pinned
    pointer table of the gather and the device tables of the weight refresh are made before); one side stream for the capture, with no
    autograd graph of an earlier step alive (AccumulateGrad nodes are bound to the stream they were created on); every packed weight
    operand marked stale so that the captured step starts with the two refresh launches."""

    def __init__(self, diffusion, buf, example_batch, overlap=None):
        """overlap: an OverlappedAllReduce -- its bucket all-reduces are started from the gradient hooks DURING the captured backward, so they
        become nodes of the graph (RCCL's stream forks from the capturing stream at each bucket and joins it again in finish()): replays then
        carry the exchange, overlapped exactly as in the eager step, and the caller must not reduce the flat buffer again (self.exchanged)."""
        import gc
        self.model, self.buf = diffusion, buf
        self.exchanged = overlap is not None
        gc.collect()                                   # drop autograd graphs of earlier steps that only the cycle collector frees
        ex = example_batch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            x = diffusion.normalize(ex) if hasattr(diffusion, 'normalize') else ex
            # static inputs (no random draw here: the generator must be consumed exactly as by eager steps)
            self.x, self.t, self.noise = x.clone(), torch.zeros((x.shape[0],), device=x.device, dtype=torch.long), torch.zeros_like(x)
            ops.prepare_graph_refresh()
        torch.cuda.current_stream().wait_stream(side)
        buf.prepare_capture()
        torch.cuda.synchronize()
        buf.zero_grad()
        self.graph = torch.cuda.CUDAGraph()
        epoch0 = ops.WEIGHT_EPOCH                      # operands the captured step touches carry used >= this
        with ops.graph_capture(self.graph, stream=side):
            loss = diffusion.p_losses(self.x, self.t, noise=self.noise)
            if overlap is not None:
                overlap.begin()
            with ops.flat_wgrad_scope():
                loss.backward()
            if overlap is not None:
                overlap.finish()
            buf.gather_grads(capture=True)
        self.loss, self.shape = loss.detach(), tuple(ex.shape)
        # the graph replays raw pointers into the operand caches of wdno_amd.ops, which evict on their own terms: hold what it reads
        self._keep, self._plans = ops.cache_snapshot(), ops.captured_plans(epoch0)
        self._capture_table = buf._capture_tables[-1]
        buf.params_changed()          # the capture only RECORDED the refresh of the packed weight operands: nothing may pass for fresh

    def draw(self, batch):
        """(x, t, noise) exactly as GaussianDiffusion.forward draws them (diffusion_2d.py:1052-1058, diffusion_1d.py:647-654): t first,
        then the noise, from the default generator -- so a graphed step consumes the same random numbers as an eager one."""
        m = self.model
        t = torch.randint(0, m.num_timesteps, (batch.shape[0],), device=batch.device).long()
        x = m.normalize(batch) if hasattr(m, 'normalize') else batch
        return x, t, m.sample_noise(tuple(x.shape), x.device)

    def run(self, batch):
        x, t, noise = self.draw(batch)
        self.x.copy_(x); self.t.copy_(t); self.noise.copy_(noise)
        self.graph.replay()
        self.buf._gathered = True                      # the replay left every gradient in the flat buffer
        ops.touch_plans(self._plans)
        return self.loss.clone()


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
        broadcast_parameters(self.opt.buf, diffusion, group)          # identical replicas, whatever the ranks' seeds were
        self.opt.buf.watch_gradient_coverage()
        self.ema = FlatEMA(self.opt.buf.flat_param, ema_decay, ema_update_every) if (use_ema and self.rank == 0) else None
        self.step_idx = 0
        self.time_comm, self.comm_events = False, []
        # The gradient exchange. Default: ONE all-reduce of the flat buffer after backward when it is small (smoke: 95 MB, ~1-2 ms exposed at 8
        # GPUs), bucket by bucket DURING backward (OverlappedAllReduce) when the buffer is >= 256 MB (Burgers: 563 MB = 0.9-6.4 ms on xGMI
        # against a 24 ms step). WDNO_DP_OVERLAP=1 / 0 forces either. WDNO_DP_FORCE_EXCHANGE=1 runs the exchange even in a one-rank process
        # group: tests/test_gpu_distributed.py drives the RCCL path (async bucket all-reduces on RCCL's stream, finish() ordering against
        # the HIP launch stream, broadcast) on the one GPU of the test box that way.
        self.exchange = self.world > 1 or (os.environ.get('WDNO_DP_FORCE_EXCHANGE', '0') == '1' and dist.is_available() and dist.is_initialized())
        ov = os.environ.get('WDNO_DP_OVERLAP')
        big = self.opt.buf.numel * 4 >= (256 << 20)
        self.overlap = (OverlappedAllReduce(self.opt.buf, group, int(os.environ.get('WDNO_DP_BUCKETS', '4')))
                        if self.exchange and (ov == '1' or (ov is None and big)) else None)

    def _backward_and_exchange(self, loss):
        if self.overlap is not None:
            self.overlap.begin()
            with ops.flat_wgrad_scope():
                loss.backward()
            self.overlap.finish()
            self.opt.buf.gather_grads()
            return
        with ops.flat_wgrad_scope():
            loss.backward()
        self.opt.buf.gather_grads()
        if self.exchange:
            if self.time_comm:                    # bench.py: the exposed part of the exchange (backward has finished), HIP events
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            allreduce_sum_(self.opt.buf.flat_grad, self.world, self.group, force=True)
            if self.time_comm:
                e1.record()
                self.comm_events.append((e0, e1))

    # ------------------------------------------------------------------ the step as ONE captured HIP graph
    def capture(self, example_batch, warmup=2):
        """Capture loss -> backward -> gradient gather of a step on batches shaped like `example_batch` in one HIP graph
        (CapturedStep); step() then replays it: ~1000 launches cost one graph launch of host time instead of ~18 ms of Python
        per step, the GPU work is unchanged (same kernels, same order: replays are bit-identical to eager steps,
        tests/test_gpu_graph.py). Left outside the graph on purpose: the random draws (two launches, so that eager and graphed steps
        consume the generator identically), the gradient exchange (RCCL) and clip + Adam (two launches whose learning rate and step
        count are host scalars). Runs `warmup` eager optimisation steps on the example first: packed weight operands, pixel tables
        and the gradient-coverage check have to exist before a capture."""
        # The overlapped bucket exchange INSIDE the graph is opt-in (WDNO_DP_GRAPH_OVERLAP=1, RCCL only): it has run on a one-rank RCCL group
        # only (no multi-GPU node in any round), and a mis-ordered fork / join of RCCL's stream under capture would leave gradients
        # un-reduced without an error. Default: the graph holds loss -> backward -> gather, and ONE all-reduce of the flat buffer follows
        # every replay (_step_graph) -- the hooks of OverlappedAllReduce stay inactive (begin() is not called). Opted in, the capture is
        # checked once against an eager overlapped step on the same inputs (_verify_overlap_capture) and dropped on any differing bit.
        in_graph = self.overlap is not None and os.environ.get('WDNO_DP_GRAPH_OVERLAP', '0') == '1'
        if in_graph and dist.get_backend(self.group) != 'nccl':
            raise RuntimeError('wdno_amd TrainStep.capture: only RCCL collectives can be recorded in a HIP graph (backend '
                               f'{dist.get_backend(self.group)!r} synchronises with the host); unset WDNO_DP_GRAPH_OVERLAP')
        for _ in range(max(1, warmup)):
            self.step(example_batch)
        # With the exchange in the graph the hooks run while the backward is being captured: the async all-reduces they start are recorded
        # as graph nodes on RCCL's stream (forked from / joined to the capturing stream), so a replay overlaps them with the rest of the backward
        # like the eager step does. A capture that RCCL refuses raises here; callers fall back to launch-by-launch steps (bench.py does).
        self._cap = CapturedStep(self.model, self.opt.buf, example_batch, overlap=self.overlap if in_graph else None)
        if in_graph and not self._verify_overlap_capture():
            import warnings
            warnings.warn('wdno_amd TrainStep.capture: a replay of the graph that holds the overlapped bucket all-reduces did not return the bits of the '
                          'eager overlapped step; the exchange stays outside the graph (one all-reduce after every replay)')
            self._cap = CapturedStep(self.model, self.opt.buf, example_batch, overlap=None)
        return self

    def _verify_overlap_capture(self):
        """One replay of the captured step (bucket all-reduces inside the graph) against the eager overlapped step on the SAME static inputs:
        the reduced flat gradients must agree bit for bit on every rank (same kernels, same order, RCCL's reduction order is fixed by the
        communicator). All ranks take the same decision (the verdict is min-reduced)."""
        cap, buf = self._cap, self.opt.buf
        cap.graph.replay()
        buf._gathered = True
        got = buf.flat_grad.clone()
        self.opt.zero_grad()
        loss = self.model.p_losses(cap.x, cap.t, noise=cap.noise)
        self._backward_and_exchange(loss)
        ok = torch.tensor([1.0 if torch.equal(got, buf.flat_grad) else 0.0], device=got.device)
        if self.world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        self.opt.zero_grad()
        return bool(ok.item() == 1.0)

    @property
    def _graph(self):
        cap = getattr(self, '_cap', None)
        return None if cap is None else cap.graph

    def _step_graph(self, batch):
        loss = self._cap.run(batch)
        if self.exchange and not self._cap.exchanged:
            if self.time_comm:                    # bench.py: the exposed exchange (the replayed backward has been enqueued in full), HIP events
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            allreduce_sum_(self.opt.buf.flat_grad, self.world, self.group, force=True)
            if self.time_comm:
                e1.record()
                self.comm_events.append((e0, e1))
        lr = self.lr_schedule(self.base_lr, self.step_idx)
        gnorm = self.opt.step(lr=lr, grad_scale=1.0 / self.world)
        self.step_idx += 1
        if self.ema is not None:
            self.ema.update(self.opt.buf.flat_param)
        return loss, gnorm

    def step(self, batch, **loss_kwargs):
        if getattr(self, '_cap', None) is not None and tuple(batch.shape) == self._cap.shape and not loss_kwargs:
            return self._step_graph(batch)
        self.opt.zero_grad()
        loss = self.model(batch, **loss_kwargs)
        self._backward_and_exchange(loss)
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
        self._backward_and_exchange(loss)
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


# ======================================================================================================================
# Reference-compatible checkpoint formats and the Trainer core (rows T1-T3 of SURVEY.md section 8a)
# ======================================================================================================================
def adam_state_to_torch(model, opt):
    """FlatAdam state -> the dictionary `torch.optim.Adam(model.parameters()).state_dict()` would hold
    (the 'opt' entry of the reference checkpoints, train_diffusion.py:155, diffusion_2d.py:1195): parameter indices
    count *every* parameter of model.parameters() (frozen ones included, they simply own no state)."""
    params = list(model.parameters())
    spans = {id(p): span for p, span in zip(opt.buf.params, opt.buf._spans())}
    state = {}
    for i, p in enumerate(params):
        if id(p) in spans and opt.step_count > 0:
            o, n = spans[id(p)]
            state[i] = {'step': torch.tensor(float(opt.step_count)), 'exp_avg': opt.exp_avg[o:o + n].view(p.shape).clone(),
                        'exp_avg_sq': opt.exp_avg_sq[o:o + n].view(p.shape).clone()}
    group = {'lr': opt.lr, 'betas': tuple(opt.betas), 'eps': opt.eps, 'weight_decay': 0, 'amsgrad': False, 'maximize': False,
             'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(len(params)))}
    return {'state': state, 'param_groups': [group]}


def adam_state_from_torch(model, opt, sd):
    """Inverse of adam_state_to_torch; accepts checkpoints written by the reference's torch.optim.Adam."""
    params = list(model.parameters())
    spans = {id(p): span for p, span in zip(opt.buf.params, opt.buf._spans())}
    steps = set()
    opt.exp_avg.zero_()
    opt.exp_avg_sq.zero_()
    for i, st in sd['state'].items():
        p = params[int(i)]
        if id(p) not in spans:
            continue
        o, n = spans[id(p)]
        opt.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
        opt.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
        steps.add(int(float(st['step'])))
    if len(steps) > 1:
        raise ValueError(f'per-parameter Adam step counters differ ({sorted(steps)}); the flat optimiser keeps one')
    opt.step_count = steps.pop() if steps else 0
    g = sd['param_groups'][0]
    opt.lr, opt.betas, opt.eps = float(g['lr']), tuple(g['betas']), float(g['eps'])


class ModelEMA:
    """ema_pytorch.EMA(model, beta, update_every) as the reference configures it (train_diffusion.py:123-125,
    diffusion_2d.py:1165-1167), on a flat buffer with one HIP launch per update.

    ema_pytorch is a third-party package that is not part of the reference tree (unpinned in env.sh); its published
    update rule is restated here: calls are counted in `step`; only every `update_every`-th call does work; until
    `update_after_step` (100) the online weights are copied; afterwards
        ema <- ema + (1 - d) * (online - ema),   d = clamp(1 - (1 + (step - update_after_step - 1) / inv_gamma)^-power, min_value, beta)
    with inv_gamma = 1, power = 2/3, min_value = 0. `ema_model` is a structural copy of the diffusion module whose
    parameters are views of the flat EMA buffer (so `trainer.ema.ema_model.sample(...)` works like in the reference).
    """

    def __init__(self, model, flat_online, beta=0.995, update_every=10, update_after_step=100, inv_gamma=1.0, power=2.0 / 3.0,
                 min_value=0.0):
        import copy
        self.online_model = model
        self.ema_model = copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.flat_online = flat_online
        self.flat = flat_online.detach().clone()
        off = 0
        for p_on, p_ema in zip(model.parameters(), self.ema_model.parameters()):
            if p_on.requires_grad:                       # same traversal order as FlatBuffers
                n = p_on.numel()
                p_ema.data = self.flat[off:off + n].view(p_on.shape)
                off += n
        assert off == self.flat.numel()
        self.beta, self.update_every, self.update_after_step = beta, update_every, update_after_step
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.step, self.initted = 0, False

    def to(self, device):
        return self

    def current_decay(self):
        epoch = max(self.step - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.0
        return min(max(1 - (1 + epoch / self.inv_gamma) ** -self.power, self.min_value), self.beta)

    def copy_params_from_model_to_ema(self):
        self.flat.copy_(self.flat_online)
        ops.bump_weight_epoch()

    def update(self):
        step = self.step
        self.step += 1
        if step % self.update_every:
            return
        if step <= self.update_after_step:
            self.copy_params_from_model_to_ema()
            return
        if not self.initted:
            self.copy_params_from_model_to_ema()
            self.initted = True
        n = self.flat.numel()
        _lib.check(_lib_().wdno_ema_update(_p(self.flat), _p(self.flat_online), n, float(self.current_decay()), _stream()), 'ema_update')
        ops.bump_weight_epoch()

    def state_dict(self):
        """Keys of ema_pytorch.EMA.state_dict(): initted, step, ema_model.*, online_model.*"""
        sd = {'initted': torch.tensor(self.initted), 'step': torch.tensor(self.step)}
        sd.update({f'online_model.{k}': v for k, v in self.online_model.state_dict().items()})
        sd.update({f'ema_model.{k}': v.clone() for k, v in self.ema_model.state_dict().items()})
        return sd

    def load_state_dict(self, sd):
        self.initted = bool(sd['initted'])
        self.step = int(sd['step'])
        self.ema_model.load_state_dict({k[len('ema_model.'):]: v for k, v in sd.items() if k.startswith('ema_model.')})
        ops.bump_weight_epoch()


class TrainerCore:
    """What both reference Trainers do, once (train_diffusion.py:40-237, diffusion_2d.py:1060-1307): data loader with
    split-batch semantics, Adam + LR schedule + global-norm clip, EMA on the main process, checkpoint dictionary
    {'step','model','opt','ema','scaler'}. `accelerate` is replaced by one process per GPU + torch.distributed
    (RCCL): if a process group is initialised (torchrun) the flat gradient is all-reduced once per step."""

    def __init__(self, diffusion_model, *, train_batch_size, gradient_accumulate_every, train_lr, train_num_steps, ema_update_every,
                 ema_decay, adam_betas, save_and_sample_every, split_batches, max_grad_norm, lr_schedule, results_dir, mixed_precision='no'):
        from pathlib import Path
        if not torch.cuda.is_available():
            raise RuntimeError('wdno_amd Trainer needs an MI355X (no CPU fallback on the hot path)')
        self.conv_math = conv_math_of(mixed_precision)           # the arithmetic of THIS trainer's steps (forward, backward, captured graph)
        self.rank, self.world, local = init_distributed()          # torchrun: set_device(LOCAL_RANK) + RCCL process group
        self._device = torch.device('cuda', local)
        self.model = diffusion_model.to(self._device)
        self.channels = diffusion_model.channels
        self.batch_size = train_batch_size
        self.local_batch_size = train_batch_size // self.world if split_batches else train_batch_size
        assert self.local_batch_size >= 1, 'train_batch_size smaller than the number of processes'
        self.gradient_accumulate_every = gradient_accumulate_every
        self.max_grad_norm = max_grad_norm
        self.train_num_steps = train_num_steps
        self.save_and_sample_every = save_and_sample_every
        self.train_lr = train_lr
        self.lr_schedule = lr_schedule
        self.opt = FlatAdam(self.model.parameters(), lr=train_lr, betas=adam_betas, max_grad_norm=max_grad_norm)
        broadcast_parameters(self.opt.buf, self.model)             # DDP semantics: every replica starts from rank 0's weights
        self.opt.buf.watch_gradient_coverage()
        self.data_seed = 1234 + 1000 * self.rank                   # loaders that are not sharded by a sampler draw per-rank streams
        if self.is_main_process:
            self.ema = ModelEMA(self.model, self.opt.buf.flat_param, beta=ema_decay, update_every=ema_update_every)
        self.results_dir = Path(results_dir)
        self.results_dir.mkdir(parents=True, exist_ok=True)
        self.step = 0
        self.total_loss = 0.0

    # ------------------------------------------------------------------ accelerate-like surface the scripts touch
    @property
    def device(self):
        return self._device

    @property
    def is_main_process(self):
        return self.rank == 0

    cycle = staticmethod(cycle_loader)
    num_workers = None          # DataLoader workers; None = the reference's choice (cpu_count() / 16). Class attribute, not a constructor
                                # argument: the constructors keep exactly the reference's parameter lists (tests/test_host.py)

    resident_data = True        # datasets that fit keep their tensors / raw coefficient files in HBM and form batches on the GPU (wdno_amd.loader:
                                # no worker processes, per-sample torch.load or host packing on the step's critical path). A class attribute like
                                # use_graph; False: the reference's DataLoader pipeline

    def make_loader(self, dataset, batch_size, num_workers, shuffle=True):
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        if self.resident_data:
            from . import loader as L
            free = torch.cuda.mem_get_info(self._device)[0]
            seed = 0 if self.world > 1 else int(torch.empty((), dtype=torch.int64).random_().item())      # ranks must agree on the permutation (DistributedSampler's seed)
            x = getattr(dataset, 'x', None)
            if torch.is_tensor(x) and x.dim() >= 2 and x.is_floating_point() and 4 * x.numel() < free // 4:
                return L.ResidentTensorLoader(dataset, batch_size, self._device, shuffle, self.rank, self.world, seed)
            if hasattr(dataset, 'raw') and not getattr(dataset, 'is_super_model', True) and len(dataset) * L.SMOKE_SIM_BYTES < free // 2:
                return L.ResidentSmokeLoader(dataset, batch_size, self._device, shuffle, self.rank, self.world, seed, num_workers=num_workers)
        sampler = DistributedSampler(dataset, num_replicas=self.world, rank=self.rank, shuffle=shuffle) if self.world > 1 else None
        return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle and sampler is None, sampler=sampler, pin_memory=True,
                          num_workers=num_workers, drop_last=False)

    # ------------------------------------------------------------------ one optimisation step (T1 / T2)
    use_graph = True            # the step (loss -> backward -> gradient gather) is replayed from one captured HIP graph per batch shape
                                # (CapturedStep; gradient_accumulate_every == 1 only; the first step of a shape runs launch by launch, the capture
                                # follows it). A class attribute like num_workers: the constructors keep the reference's parameter lists.
                                # Bit-identical to the launch-by-launch step (tests/test_gpu_trainer.py); False: every step launch by launch.

    def optimisation_step(self, next_batch):
        """next_batch() -> device tensor. Returns the python float loss of this rank (the reference logs it per rank). Runs under this
        trainer's arithmetic (conv_math_of): like accelerate's autocast around `self.model(data)` (train_diffusion.py:205-208), the mode covers
        the training step only -- sampling from the EMA copy between steps stays fp32-equivalent."""
        with ops._math(self.conv_math):
            return self._optimisation_step(next_batch)

    def _optimisation_step(self, next_batch):
        total = None
        if self.use_graph and self.gradient_accumulate_every == 1:
            batch = next_batch()
            caps = self.__dict__.setdefault('_caps', {})                 # one captured step per batch shape (the loader's last batch of an epoch is short)
            cap = caps.get(tuple(batch.shape))
            if cap is not None:
                caps[tuple(batch.shape)] = caps.pop(tuple(batch.shape))        # most recently used last: eviction below takes the least recently used
                total = float(cap.run(batch))
            else:
                next_batch = lambda _b=batch: _b            # this step runs launch by launch (it is the warm-up), the capture follows it
        if total is None:
            self.opt.zero_grad()
            total = 0.0
            for _ in range(self.gradient_accumulate_every):
                loss = self.model(next_batch()) / self.gradient_accumulate_every
                with ops.flat_wgrad_scope():
                    loss.backward()
                total += float(loss.detach())
            del loss
            self.opt.buf.gather_grads()
        if self.world > 1:
            allreduce_sum_(self.opt.buf.flat_grad, self.world)
        self.last_grad_norm = self.opt.step(lr=self.lr_schedule(self.train_lr, self.step), grad_scale=1.0 / self.world)
        self.total_loss = total
        if self.use_graph and self.gradient_accumulate_every == 1 and tuple(batch.shape) not in self._caps:
            if len(self._caps) >= 3:                         # a capture pins one step's activations: at most three shapes, the least recently used goes
                self._caps.pop(next(iter(self._caps)))
            try:
                self._caps[tuple(batch.shape)] = CapturedStep(self.model, self.opt.buf, batch)
            except Exception as e:
                # Only failures OF THE CAPTURE are survivable (an op that is illegal under stream capture, a host sync, no memory for the
                # private pool): the run continues launch by launch, as it did before graphs were the default. Anything else raised by the
                # captured forward / backward -- a shape error, the GroupNorm parameter-epoch guard, a WDNO_E* library error -- is a bug that
                # the eager step would hit as well and is re-raised; WDNO_STRICT_GRAPH=1 re-raises capture failures too (tests).
                if not is_capture_failure(e) or os.environ.get('WDNO_STRICT_GRAPH', '0') == '1':
                    self._capture_failed(e, warn=False)
                    raise
                self._capture_failed(e)
        return total

    def _capture_failed(self, err, warn=True):
        import gc
        import warnings
        if warn:
            warnings.warn(f'wdno_amd Trainer: capturing the training step in a HIP graph failed ({err!r:.300}); continuing launch by launch '
                          '(use_graph = False for this trainer)')
        self.use_graph = False
        self._caps.clear()                                   # frees the private pools and operand snapshots of the captures made so far
        ops._CAPTURE = None
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        self.opt.buf._gathered = False
        self.opt.buf.params_changed()                        # whatever the aborted capture recorded as refreshed is not

    @property
    def _cap(self):                   # the most recent capture (tests / diagnostics)
        caps = self.__dict__.get('_caps') or {}
        return next(reversed(caps.values()), None) if caps else None

    # ------------------------------------------------------------------ checkpoints (T3)
    def checkpoint_dict(self):
        return {'step': self.step, 'model': self.model.state_dict(), 'opt': adam_state_to_torch(self.model, self.opt),
                'ema': self.ema.state_dict(), 'scaler': None}

    def load_checkpoint_dict(self, data):
        self.model.load_state_dict(data['model'])
        self.step = data['step']
        adam_state_from_torch(self.model, self.opt, data['opt'])
        if self.is_main_process:
            self.ema.load_state_dict(data['ema'])
        ops.bump_weight_epoch()
