cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
python -X faulthandler -m pytest tests/test_model_parity.py -q -x -m gpu -k "trainer_default" > gpurun_out/r5c/t_default.log 2>&1; echo "default rc=$?"
TMG_NO_HIP_ADAM=1 python -X faulthandler -m pytest tests/test_model_parity.py -q -x -m gpu -k "trainer_default" > gpurun_out/r5c/t_default_noadopt.log 2>&1; echo "default noadopt rc=$?"
python -X faulthandler -m pytest tests/test_model_parity.py -q -x -m gpu -k "trainer_epoch or captured_window" > gpurun_out/r5c/t_explicit.log 2>&1; echo "explicit rc=$?"
python -X faulthandler tools/window_run.py --windows 2 --capture --batch 8 > gpurun_out/r5c/w_capture.json 2> gpurun_out/r5c/w_capture.err; echo "capture b8 rc=$?"
python -X faulthandler tools/window_run.py --windows 3 --batch 8 > gpurun_out/r5c/w_default.json 2> gpurun_out/r5c/w_default.err; echo "default b8 rc=$?"
python -X faulthandler tools/window_run.py --windows 3 --batch 8 --no-adopt > gpurun_out/r5c/w_default_na.json 2> gpurun_out/r5c/w_default_na.err; echo "default b8 noadopt rc=$?"
for f in t_default t_default_noadopt t_explicit; do tail -4 gpurun_out/r5c/$f.log; done
