"""`DataParallelINNModel` for the one-process-per-GPU design.

The reference's class of this name (utils/parallel.py:74-241) is a single-process thread-per-GPU `nn.DataParallel`:
it scatters the batch, re-broadcasts the parameters every BPTT window (`scatterModel`), runs one Python thread per
replica and gathers the LSTM states to the source device after every time-step.  Here every rank is its own process
with a persistent replica (launched with torchrun; `tmg_dist.init_from_env`), so the wrapper keeps the INTERFACE
`main.py:76` and the trainer use - `.module`, `sample()`, `forward()`, `scatterModel()`, `gatherLSTMStates()`,
`gather()`, `parameters()` / `state_dict()` - and the data-path methods are rank-local: nothing is scattered,
re-broadcast or gathered.  The only exchange of the path, the gradient all-reduce, lives in `TrainFlow.trainParallel`."""
import torch
import torch.nn as nn


class DataParallelINNModel(nn.Module):
    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids) if device_ids is not None else []
        self.dim = dim
        import tmg_dist
        tmg_dist.broadcast_parameters(module)   # replicas start identical (no-op for one process)

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)

    def sample(self, *inputs, **kwargs):
        return self.module.sample(*inputs, **kwargs)

    def scatterModel(self, n_gpu=None):
        """Replicas are persistent: nothing to broadcast per BPTT window (reference :137-150)."""
        return None

    def scatterRecurrentStates(self, recFeatures):
        return recFeatures

    def gatherLSTMStates(self, *inputs):
        """LSTM states stay on their rank (reference :156-157 gathers them to the source GPU every time-step)."""
        return inputs[0] if len(inputs) == 1 else inputs

    def gather(self, outputs, output_device=None):
        return outputs


class DataParallelCriterion(nn.Module):
    """Loss wrapper with the reference's call shape `criterion(outputs, *targets)` (utils/parallel.py:243-280); rank-local."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super().__init__()
        self.module = module

    def forward(self, inputs, *targets, **kwargs):
        if isinstance(inputs, (tuple, list)) and not torch.is_tensor(inputs):
            return self.module(*inputs, *targets, **kwargs)
        return self.module(inputs, *targets, **kwargs)
