cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
python -m pytest tests/test_hip_ops.py tests/test_model_parity.py -q -x -m gpu -k "coupling or level_kernels or dense2 or c1 or conv_cases or winograd_conv or fused" 2>&1 | tail -3
for i in 1 2 3; do
  (cd _old && LAYOUT=split python tools/bench_narrow.py 2>/dev/null | sed 's/^/old /') >> gpurun_out/r5g/ab_narrow.txt
  LAYOUT=split python tools/bench_narrow.py 2>/dev/null | sed 's/^/new /' >> gpurun_out/r5g/ab_narrow.txt
done
cat gpurun_out/r5g/ab_narrow.txt
