set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
python -m pytest tests/test_dist_gpu.py tests/test_hip_ops.py tests/test_model_parity.py -q -x -m gpu -k "dist or adam or trainer_epoch or cfg5_stated_batch_with_fp16 or captured_window" 2>&1 | tail -25 > gpurun_out/r5a/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
python tools/window_run.py --windows 3 > gpurun_out/r5a/window_default.json 2> gpurun_out/r5a/window_default.err
python tools/window_run.py --windows 3 --eager --no-adopt > gpurun_out/r5a/window_eager.json 2> gpurun_out/r5a/window_eager.err
tail -3 gpurun_out/r5a/tests.log; cat gpurun_out/r5a/window_default.json gpurun_out/r5a/window_eager.json; python -c "
import json; d=json.load(open('gpurun_out/r5a/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('second_class'), d.get('cpu_baseline'), d.get('cpu_baseline_cfg1'))"
