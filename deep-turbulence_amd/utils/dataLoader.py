"""Training / testing data of the TM-Glow examples (reference utils/dataLoader.py).

On-disk format (reference :68-99): one `.npz` per simulation, key `data` = [T, 4 (u_x, u_y, u_z, p), H, W]; u_z is
dropped.  Pipeline of the reference, kept step for step: bilinear pre-upscale of the low-fidelity field
(align_corners=True) -> division by the inlet velocity (u, u, u^2) and inlet velocity as 4th input channel (backward
step only) -> per-channel z-score -> split of every series into `tSplit` sub-series -> one LSTM seed per sub-series.

MI355X-first: the normalised set is moved to HBM once (it is a few GB; the device has 288) and a batch is an index
gather plus the additive noise drawn ON the device, so nothing is copied or randomised on the host inside the step.
`DeviceLoader` iterates like the reference's torch DataLoader (same tuples, same len, same drop_last rules)."""
import os

import numpy as np
import torch
import torch.nn.functional as F


class DeviceLoader(object):
    """Batches of (input, target, third) gathered on `device`; `third` is the LSTM seed (training) or u0 (testing).
    Noise: input += input_noise_std * N(0,1), target += target_noise_std * N(0,1), drawn on the device per batch
    (reference TrainingDataset.__getitem__ :40-45 draws it per item on the host)."""

    def __init__(self, inputs, targets, third, batch_size, shuffle, drop_last, input_noise_std=0.0, target_noise_std=0.0,
                 device=None):
        assert inputs.size(0) == targets.size(0) == third.size(0), 'tensors must share the batch dimension'
        dev = device if device is not None else inputs.device
        self.inputs, self.targets, self.third = inputs.to(dev), targets.to(dev), third.to(dev)
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.input_noise_std, self.target_noise_std = input_noise_std, target_noise_std
        self.dataset = self.inputs  # len(loader.dataset) is used by the reference's prediction code (utils/utils.py:189)
        self.rank, self.world, self._epoch = 0, 1, 0

    def set_shard(self, rank, world):
        """Data-parallel sharding: this rank yields rows rank::world of every global batch.  The permutation of an epoch is
        drawn on the host from a generator seeded identically on all ranks (base seed from the global torch RNG at the time
        of the call, which `main.py`'s seeding makes equal across ranks), so the ranks walk disjoint shards of the SAME
        global batches - the reference's scatter of one batch across GPUs (parallel.py:118)."""
        assert self.batch_size % world == 0, "global batch must be divisible by the number of GPUs"
        self.rank, self.world = int(rank), int(world)
        seed = [int(torch.initial_seed() % (2 ** 31))]
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.broadcast_object_list(seed, src=0)   # one permutation stream for all ranks even if their RNGs were seeded apart
        self._seed = seed[0]

    def _tail(self):
        """Rows of the last, partial global batch that are handed out (0 = there is none).  Sharded: truncated to a multiple of the
        world size, so that every rank gets the same number of rows (the gradient mean over ranks stays the mean over the rows)
        and no rank ever sees an empty batch; the reference's scatter simply uses fewer replicas for a short batch
        (parallel.py:118), which one-process-per-GPU data parallelism cannot do without idling ranks inside a collective."""
        r = 0 if self.drop_last else self.inputs.size(0) % self.batch_size
        return r - r % self.world

    def __len__(self):
        return self.inputs.size(0) // self.batch_size + (1 if self._tail() else 0)

    def __iter__(self):
        n = self.inputs.size(0)
        if self.world > 1:
            g = torch.Generator().manual_seed(self._seed + self._epoch)
            self._epoch += 1
            order = (torch.randperm(n, generator=g) if self.shuffle else torch.arange(n)).to(self.inputs.device)
        else:
            order = torch.randperm(n, device=self.inputs.device) if self.shuffle else torch.arange(n, device=self.inputs.device)
        full = n // self.batch_size
        for i in range(len(self)):
            rows = self.batch_size if i < full else self._tail()
            idx = order[i * self.batch_size:i * self.batch_size + rows][self.rank::self.world]
            x, y = self.inputs[idx], self.targets[idx]
            if self.input_noise_std:
                x = x + self.input_noise_std * torch.randn_like(x)
            if self.target_noise_std:
                y = y + self.target_noise_std * torch.randn_like(y)
            yield x, y, self.third[idx]


class TMGLowDataLoader(object):
    """Reads the .npz files and holds the normalising constants (reference :47-207)."""

    def __init__(self, training_dir='.', testing_dir='.', log=None, device=None):
        self.training_dir, self.testing_dir, self.log = training_dir, testing_dir, log
        self.device = device
        self.input_mean = self.output_mean = self.input_std = self.output_std = None

    def _say(self, msg):
        if self.log is not None:
            self.log.log(msg)

    @staticmethod
    def _read(path, stride):
        with np.load(path) as z:
            d = z['data'][::stride]
        return np.concatenate([d[:, :2], d[:, 3:]], axis=1)  # drop u_z

    def readFluidData(self, input_file_name, target_file_name, fStride=1, cStride=1):
        """-> (low-fidelity [T,3,h,w], high-fidelity [T,3,H,W]) numpy arrays; a missing file gives None for its half."""
        out = []
        for name, stride in ((input_file_name, cStride), (target_file_name, fStride)):
            path = os.path.join(self.training_dir, name)
            out.append(self._read(path, stride) if os.path.isfile(path) else None)
            if out[-1] is None and self.log is not None:
                self.log.error('Data file not found: {}'.format(path))
        return out[0], out[1]

    def calcNormalizingParams(self, inputData, targetData):
        """Mean / (unbiased) std of the first three channels over everything else, [b,t,c,h,w] tensors."""
        def stats(t):
            flat = t[:, :, :3].transpose(0, 2).reshape(3, -1)
            return flat.mean(1), flat.std(1)
        self.input_mean, self.input_std = stats(inputData)
        self.output_mean, self.output_std = stats(targetData)

    def setNormalizingParams(self, model):
        self.input_mean, self.input_std = model.in_mu.cpu(), model.in_std.cpu()
        self.output_mean, self.output_std = model.out_mu.cpu(), model.out_std.cpu()

    def transferNormalizingParams(self, model):
        dev = next(model.parameters()).device
        model.in_mu, model.in_std = self.input_mean.to(dev), self.input_std.to(dev)
        model.out_mu, model.out_std = self.output_mean.to(dev), self.output_std.to(dev)

    @staticmethod
    def _zscore(t, mean, std):
        t[:, :, :3] = (t[:, :, :3] - mean.view(1, 1, 3, 1, 1).to(t)) / std.view(1, 1, 3, 1, 1).to(t)
        return t

    def normalizeInputData(self, inputData):
        return self._zscore(inputData, self.input_mean, self.input_std)

    def normalizeTargetData(self, targetData):
        return self._zscore(targetData, self.output_mean, self.output_std)

    # ---- shared pipeline ----------------------------------------------------------------------------------------
    def _load_cases(self, pattern, cases, inUpscale):
        lo, hi = zip(*[self.readFluidData(pattern[0].format(i), pattern[1].format(i)) for i in cases])
        lo, hi = torch.from_numpy(np.stack(lo)).float(), torch.from_numpy(np.stack(hi)).float()
        b, t = lo.shape[:2]
        lo = F.interpolate(lo.flatten(0, 1), scale_factor=inUpscale, mode='bilinear', align_corners=True)
        return lo.view(b, t, *lo.shape[1:]), hi

    def _normalise(self, lo, hi):
        if self.input_mean is None or self.input_std is None:
            self.calcNormalizingParams(lo, hi)
        return self.normalizeInputData(lo), self.normalizeTargetData(hi)

    @staticmethod
    def _split(t, tSplit):
        n = t.size(1) // tSplit
        return torch.cat([t[:, i * n:(i + 1) * n] for i in range(tSplit)], dim=0)


def _by_inlet(lo, hi, u0):
    """Backward step: fields / (u0, u0, u0^2); u0 appended to the input as a constant 4th channel (reference :252-260)."""
    s = torch.stack([u0, u0, u0 * u0], dim=1).view(-1, 1, 3, 1, 1)
    lo, hi = lo / s, hi / s
    plane = u0.view(-1, 1, 1, 1, 1).expand(lo.size(0), lo.size(1), 1, lo.size(3), lo.size(4))
    return torch.cat([lo, plane], dim=2), hi


class BackwardStepLoader(TMGLowDataLoader):
    FILES = ("backwardStepCoarse{:d}-[U,p].npz", "backwardStepFine{:d}-[U,p].npz")

    def __init__(self, training_dir, testing_dir, shuffle=True, log=None, device=None):
        super().__init__(training_dir, testing_dir, log, device)
        self.shuffle = shuffle

    def createTrainingLoader(self, ntrain, u0, tSplit=1, inUpscale=1, batch_size=32, tar_noise_std=0):
        batch_size = min(batch_size, len(ntrain) * tSplit)
        lo, hi = self._load_cases(self.FILES, ntrain, inUpscale)
        lo, hi = _by_inlet(lo, hi, torch.tensor([float(u0[i]) for i in ntrain]))
        lo, hi = self._normalise(lo, hi)
        lo, hi = self._split(lo, tSplit), self._split(hi, tSplit)
        for _ in range(tSplit):
            np.random.randint(0, lo.size(1))  # the reference draws (and ignores) one number per split (:273): keep its RNG stream
        seeds = torch.LongTensor(lo.size(0)).random_(0, 1000)
        # the reference passes tar_noise_std as TrainingDataset's 4th positional argument, which is the INPUT noise (:287, :24)
        return DeviceLoader(lo, hi, seeds, batch_size, self.shuffle, False, input_noise_std=tar_noise_std, device=self.device)

    def createTestingLoader(self, ntest, u0, inUpscale=1, batch_size=32):
        batch_size = min(batch_size, len(ntest))
        u = torch.tensor([float(u0[i]) for i in ntest])
        lo, hi = self._load_cases(self.FILES, ntest, inUpscale)
        lo, hi = self._normalise(*_by_inlet(lo, hi, u))
        return DeviceLoader(lo, hi, u, batch_size, self.shuffle, False, device=self.device)


class CylinderArrayLoader(TMGLowDataLoader):
    FILES = ("cylinderArrayCoarse{:d}-[U,p].npz", "cylinderArrayFine{:d}-[U,p].npz")

    def __init__(self, training_dir, testing_dir, shuffle=True, log=None, device=None):
        super().__init__(training_dir, testing_dir, log, device)
        self.shuffle = shuffle

    def createTrainingLoader(self, ntrain, tSplit=1, inUpscale=1, batch_size=32, tar_noise_std=0):
        batch_size = min(batch_size, len(ntrain) * tSplit)
        lo, hi = self._normalise(*self._load_cases(self.FILES, ntrain, inUpscale))
        lo, hi = self._split(lo, tSplit), self._split(hi, tSplit)
        for _ in range(tSplit):
            np.random.randint(0, lo.size(1))
        seeds = torch.LongTensor(lo.size(0)).random_(0, 1000)
        return DeviceLoader(lo, hi, seeds, batch_size, self.shuffle, True, input_noise_std=tar_noise_std, device=self.device)

    def createTestingLoader(self, ntest, inUpscale=1, batch_size=32):
        batch_size = min(batch_size, len(ntest))
        lo, hi = self._normalise(*self._load_cases(self.FILES, ntest, inUpscale))
        return DeviceLoader(lo, hi, torch.ones(lo.size(0)), batch_size, self.shuffle, False, device=self.device)


class DataLoaderAuto(object):
    """Loader factory keyed on `args.exp_type` (reference utils/dataLoader.py:476-538): same case selection, inlet
    velocities, splits and up-scaling; returns (loader object, training loader, testing loader) as `main.py:86` expects.
    The batches live on the model's device; under torchrun the TRAINING loader hands every rank its own shard of each
    global batch (rank-strided, same seeded permutation on all ranks), the reference's `scatter` along dim 0."""

    @classmethod
    def init_data_loaders(cls, args, model, log):
        if args.exp_type == 'backward-step':
            return cls.setupBackwardStepLoaders(args, model, log)
        if args.exp_type == 'cylinder-array':
            return cls.setupCylinderLoaders(args, model, log)
        raise AssertionError("Provided experiment name, {:s}, not supported .".format(args.exp_type))

    @staticmethod
    def _core(model):
        return getattr(model, "module", model)

    @classmethod
    def _finish(cls, args, model, loader, make_train, make_test):
        core = cls._core(model)
        if args.epoch_start > 0:   # resumed run: the normalising constants ride in the model's buffers (reference :507-509)
            loader.setNormalizingParams(core)
            training_loader = make_train()
        else:
            training_loader = make_train()
            loader.transferNormalizingParams(core)
        return loader, _shard_for_rank(training_loader), make_test()

    @classmethod
    def setupBackwardStepLoaders(cls, args, model, log):
        log.log('Setting up backward step loaders.')
        cases = np.arange(0, 64, 1)
        np.random.seed(args.seed)
        np.random.shuffle(cases)
        ntest = cases[-args.ntest:]
        ntrain = np.linspace(0, 63, args.ntrain).astype(int)
        u0 = np.linspace(1, 10, 64)
        dev = next(cls._core(model).parameters()).device
        ld = BackwardStepLoader(args.training_data_dir, args.testing_data_dir, log=log, device=dev)
        return cls._finish(
            args, model, ld,
            lambda: ld.createTrainingLoader(ntrain, u0, tSplit=2, inUpscale=(1.34, 1.34), batch_size=args.batch_size,
                                            tar_noise_std=args.noise_std),
            lambda: ld.createTestingLoader(ntest, u0, inUpscale=(1.34, 1.34), batch_size=args.test_batch_size))

    @classmethod
    def setupCylinderLoaders(cls, args, model, log):
        log.log('Setting up cylinder array loaders.')
        ntest = np.arange(96, 96 + args.ntest, 1).astype(int)
        ntrain = np.linspace(0, 95, args.ntrain).astype(int)
        dev = next(cls._core(model).parameters()).device
        ld = CylinderArrayLoader(args.training_data_dir, args.testing_data_dir, log=log, device=dev)
        return cls._finish(args, model, ld,
                           lambda: ld.createTrainingLoader(ntrain, tSplit=2, batch_size=args.batch_size),
                           lambda: ld.createTestingLoader(ntest, batch_size=args.test_batch_size))


def _shard_for_rank(loader):
    """Under torch.distributed every rank takes rows rank::world of each global batch (the batch must divide evenly,
    reference parallel.py:84-86)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        loader.set_shard(dist.get_rank(), dist.get_world_size())
    return loader
