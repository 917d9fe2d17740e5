cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
timeout 900 python tools/window_run.py --config cfg5 --batch 64 --windows 1 --recompute --adam hip > gpurun_out/r5f/window_cfg5_recompute.json 2> gpurun_out/r5f/window_cfg5_recompute.err; echo "cfg5 recompute rc=$?"
cat gpurun_out/r5f/window_cfg5_recompute.json | cut -c1-600; tail -3 gpurun_out/r5f/window_cfg5_recompute.err | cut -c1-300
timeout 600 python tools/window_run.py --config M --noc 4 --batch 64 --windows 2 --recompute --adam hip > gpurun_out/r5f/window_M_recompute.json 2> gpurun_out/r5f/window_M_recompute.err; echo "M recompute rc=$?"
cat gpurun_out/r5f/window_M_recompute.json | cut -c1-600
timeout 600 python tools/window_run.py --config M --noc 4 --batch 64 --windows 2 --adam hip > gpurun_out/r5f/window_M_stored.json 2> gpurun_out/r5f/window_M_stored.err; echo "M stored rc=$?"
cat gpurun_out/r5f/window_M_stored.json | cut -c1-600
