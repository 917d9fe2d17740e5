cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
python -X faulthandler -m pytest tests/test_model_parity.py -q -x -m gpu -k "trainer_default or cfg5_stated_batch_with_fp16" > gpurun_out/r5d/t_default.log 2>&1; echo "default rc=$?"
python -X faulthandler tools/window_run.py --windows 3 > gpurun_out/r5d/w_default.json 2> gpurun_out/r5d/w_default.err; echo "default rc=$?"
python -X faulthandler tools/window_run.py --windows 3 --eager --no-adopt > gpurun_out/r5d/w_eager.json 2> gpurun_out/r5d/w_eager.err; echo "eager rc=$?"
python -X faulthandler tools/window_run.py --windows 3 --noc 4 --capture --adam hip > gpurun_out/r5d/w_M_capture.json 2> gpurun_out/r5d/w_M_capture.err; echo "M capture rc=$?"
tail -4 gpurun_out/r5d/t_default.log; cat gpurun_out/r5d/w_*.json | cut -c1-700
