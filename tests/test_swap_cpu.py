"""The module-level drop-in of INTEGRATION.md section 1, executed for real (build container only: needs /root/reference).

A child process puts `deep-turbulence_amd/` AHEAD of the reference's `tmglow/` on sys.path and then executes the reference's
own `main.py` lines - the import block (:13-20), the model construction (:61-72), the wrapper / optimizer / scheduler
(:76-79), the data loaders (:86) and the trainer (:87) - read from the reference file at test time (nothing of it is stored
here).  Everything up to the first kernel launch runs on the CPU (`args.src_device = "cpu"`), so this checks that every name
`main.py` imports resolves, resolves to THIS package where it provides the module, and accepts the arguments main.py
passes."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deep-turbulence_amd")
REF = "/root/reference/tmglow"

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "main.py")), reason="reference checkout not present")

CHILD = r'''
import os, sys
PKG, REF, DATA = sys.argv[1:4]
sys.argv = ["main.py", "--exp-type", "cylinder-array", "--ntrain", "2", "--ntest", "1", "--batch-size", "2", "--test-batch-size", "1",
            "--training_data_dir", DATA, "--testing_data_dir", DATA, "--enc-blocks", "1", "1", "--glow-blocks", "2", "2",
            "--rec-features", "4", "--cond-features", "4", "--init-features", "8", "--exp-dir", os.path.join(DATA, "exp")]
sys.path[:0] = [PKG, REF]                       # INTEGRATION.md section 1: this package ahead of tmglow/
src = open(os.path.join(REF, "main.py")).read().splitlines()
block = lambda a, b: "\n".join(l[4:] if l.startswith("    ") else l for l in src[a - 1:b])   # 1-based, inclusive; de-indent
ns = {"__name__": "swap_test"}
exec(compile("\n".join(src[12:24]), "main.py:13-24", "exec"), ns)       # the import block (+ torch / numpy / os)
import inspect
for name, mod in (("TMGlow", "nn/tmGlow.py"), ("TrainFlow", "nn/trainFlowParallel.py"), ("DataLoaderAuto", "utils/dataLoader.py"),
                  ("saveWorkspace", "utils/utils.py"), ("Log", "utils/log.py"), ("DataParallelINNModel", "utils/parallel.py")):
    f = inspect.getsourcefile(ns[name])
    assert f == os.path.join(PKG, mod), (name, f)
assert inspect.getsourcefile(ns["Parser"]) == os.path.join(REF, "args.py")   # host config stays the reference's
import utils.viz, pc.grad1Filter                  # modules only the reference has keep resolving (pkgutil.extend_path)
assert utils.viz.__file__.startswith(REF) and pc.grad1Filter.__file__.startswith(REF)
exec(block(31, 31), ns)                          # args = Parser().parse()
exec(block(35, 35), ns)                          # log = Log(args, record=True)
args = ns["args"]
args.device, args.device_ids, args.src_device, args.n_gpu = ns["torch"].device("cpu"), [0], "cpu", 1
exec(block(61, 72), ns)                          # model = TMGlow(...).to(args.src_device)
exec(block(76, 79), ns)                          # DataParallelINNModel, Adam(amsgrad), ExponentialLR
exec(block(86, 87), ns)                          # DataLoaderAuto.init_data_loaders, TrainFlow
model, trainer = ns["model"], ns["modelTrainer"]
assert type(model).__name__ == "DataParallelINNModel" and type(model.module).__name__ == "TMGlow"
x, y, seeds = next(iter(ns["training_loader"]))
assert x.shape[0] == 2 and x.shape[2] == 3 and y.shape[-1] == 4 * x.shape[-1] and seeds.shape == (2,)
assert float(model.module.out_std.abs().sum()) > 0          # transferNormalizingParams reached the wrapped model
assert hasattr(trainer, "trainParallel") and hasattr(trainer, "test")
import tempfile
ns["saveWorkspace"](args, model.module, ns["optimizer"], file_id=1)     # main.py:120
assert ns["loadWorkspace"](args, args.ckpt_dir, file_id=1) is not None   # main.py:34
print("SWAP-OK", len(list(model.parameters())))
'''


def test_main_py_import_block_and_construction_resolve_here(tmp_path):
    rs = np.random.RandomState(5)
    for case in (0, 95, 96):
        np.savez(os.path.join(tmp_path, "cylinderArrayCoarse%d-[U,p].npz" % case), data=rs.standard_normal((4, 4, 4, 4)).astype(np.float32))
        np.savez(os.path.join(tmp_path, "cylinderArrayFine%d-[U,p].npz" % case), data=rs.standard_normal((4, 4, 16, 16)).astype(np.float32))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", MPLBACKEND="Agg")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", CHILD, PKG, REF, str(tmp_path)], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "SWAP-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
