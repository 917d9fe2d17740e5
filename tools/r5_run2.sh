set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
python -X faulthandler -m pytest tests/test_model_parity.py -q -x -m gpu -k "trainer_default or cfg5_stated_batch_with_fp16" 2>&1 | tail -40 > gpurun_out/r5b/tests.log
python -X faulthandler tools/window_run.py --windows 3 > gpurun_out/r5b/window_default.json 2> gpurun_out/r5b/window_default.err
echo "rc=$?" >> gpurun_out/r5b/window_default.err
tail -5 gpurun_out/r5b/tests.log; cat gpurun_out/r5b/window_default.json; tail -5 gpurun_out/r5b/window_default.err | cut -c1-400
