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
