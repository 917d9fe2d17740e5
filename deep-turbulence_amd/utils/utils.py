"""Workspace (checkpoint) files in the reference's on-disk format (reference
utils/utils.py:55-148): `<name><id>.zip` holding `torchModel<id>.pth` = {'epoch', 'state_dict', 'optimizer'} and
`args.json`.  A workspace written by either code base loads in the other (the state_dict schema is the reference's, see
tests/test_boundary_cpu.py)."""
import io
import json
import os
import zipfile

import torch

# run-specific arguments a loaded workspace must not overwrite (reference utils/utils.py:20)
PARAM_BLACKLIST = ['epoch_start', 'epochs', 'run_dir', 'ckpt_dir', 'pred_dir']


def _model_entry(file_id):
    return 'torchModel{:d}.pth'.format(file_id)


def saveWorkspace(args, model, optimizer, file_name="nsWorkspace", file_id=0):
    """Write `<args.ckpt_dir>/<file_name><file_id>.zip`.  Built in memory: nothing but the zip touches the disk."""
    blob = io.BytesIO()
    torch.save({'epoch': file_id, 'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict()}, blob)
    arg_dict = {k: v for k, v in vars(args).items() if k != 'device'}
    path = os.path.join(args.ckpt_dir, '{}{:d}.zip'.format(file_name, file_id))
    with zipfile.ZipFile(path, 'w', compression=zipfile.ZIP_DEFLATED) as z:
        z.writestr(_model_entry(file_id), blob.getvalue())
        z.writestr('args.json', json.dumps(arg_dict, indent=4, default=str))
    return path


def loadWorkspace(args, file_dir, file_name="nsWorkspace", file_id=0):
    """-> (args, model_state_dict, optimizer_state_dict), or None when the zip does not exist (as the reference).
    Arguments stored in the workspace overwrite those of `args` except the PARAM_BLACKLIST ones.  Tensors are loaded on
    the host; `load_state_dict` moves them."""
    path = os.path.join(file_dir, '{}{:d}.zip'.format(file_name, file_id))
    if not os.path.isfile(path):
        print('[LoadWorkspace] Could not find workspace zip file: {}'.format(path))
        return None
    with zipfile.ZipFile(path) as z:
        names = set(z.namelist())
        if 'args.json' in names:
            try:
                for key, val in json.loads(z.read('args.json').decode()).items():
                    if key not in PARAM_BLACKLIST:
                        setattr(args, key, val)
            except ValueError as e:  # a damaged args file does not block the weights (reference :130-132)
                print('[LoadWorkspace] Could not read args.json: {}'.format(e))
        if _model_entry(file_id) not in names:
            raise FileNotFoundError('{} holds no {}'.format(path, _model_entry(file_id)))
        state = torch.load(io.BytesIO(z.read(_model_entry(file_id))), map_location='cpu', weights_only=False)
    return args, state['state_dict'], state['optimizer']


def modelPred(args, model, testing_loader, log, samples=1, stride=1, tmax=1):
    """Roll the model out over the test set for post-processing (reference utils/utils.py:151-235, the plotting scripts'
    entry): `samples` independent roll-outs of `tmax` steps per test case, every `stride`-th step kept, everything
    un-normalised and scaled back by the case's inlet velocity u0 (velocities x u0, pressure x u0^2).

    Returns (ypred [samples, N, tmax // stride, C, H, W], ytarget [N, T, C, H, W], yinput [N, T, 3, h, w]) on the CPU.
    The recurrent states are re-anchored half-way to their seed states every 20 steps, as the reference does."""
    core = getattr(model, "module", model)
    core.eval()
    dev = torch.device(args.device) if getattr(args, "device", None) is not None else next(core.parameters()).device
    shp = (1, -1, 1, 1)
    in_std, in_mu = core.in_std.to(dev).view(shp), core.in_mu.to(dev).view(shp)
    out_std, out_mu = core.out_std.to(dev).view(shp), core.out_mu.to(dev).view(shp)
    nkeep = tmax // stride
    preds, targets, inputs = [], [], []
    with torch.no_grad():
        for mbIdx, (input0, target0, u0) in enumerate(testing_loader):
            log.log('Running mini-batch {:d}/{:d}'.format(mbIdx + 1, len(testing_loader)))
            u = u0.to(dev).view(-1, 1, 1, 1, 1)
            u = torch.cat((u, u, u ** 2), dim=2)                      # [N,1,3,1,1]: (ux, uy, p) scales
            inp = input0.to(dev)
            tgt = u * (out_std * target0.to(dev) + out_mu)
            inputs.append((u * (in_std * inp[:, :, :3] + in_mu)).cpu())
            targets.append(tgt.cpu())
            mb = torch.full((samples, inp.size(0), nkeep) + tuple(tgt.shape[2:]), 10000.0, device=dev, dtype=inp.dtype)
            for i in range(samples):
                log.log('Running sample {:d}.'.format(i))
                seeds = torch.LongTensor(inp.size(0)).random_(0, int(1e8))
                key = core.initLSTMStates(seeds, [tgt.size(-2), tgt.size(-1)], cache=False)
                h0 = [(h.clone(), c.clone()) for h, c in key]
                for tstep in range(tmax):
                    y0, _logp, h0 = core.sample(inp[:, tstep], h0)
                    if tstep % stride == 0 and tstep // stride < nkeep:
                        mb[i, :, tstep // stride] = u[:, 0] * (out_std * y0 + out_mu)
                    if tstep % 20 == 0:
                        h0 = [(0.5 * h + 0.5 * hk, 0.5 * c + 0.5 * ck) for (h, c), (hk, ck) in zip(h0, key)]
            log.log('Number of elements unset: {}'.format(int((mb > 10000).sum())))
            preds.append(mb.cpu())
    return torch.cat(preds, dim=1), torch.cat(targets, dim=0), torch.cat(inputs, dim=0)
