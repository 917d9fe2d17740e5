cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
python -m pytest tests/test_model_parity.py -q -x -m gpu -k "bf16x3" 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5e/bench.json 2> gpurun_out/r5e/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r5e/bench.json')); print(d['value'], d['ms_per_step']); print(d.get('wino_bf16x3_variant'))"
TMG_WINO_BF3=1 python -m pytest tests/test_model_parity.py -q -x -m gpu -k "stated_batches_match_oracle_with_gradients and M" 2>&1 | tail -4
cp gpurun_out/parity_*M*batch64*.json gpurun_out/r5e/ 2>/dev/null; ls gpurun_out | head -30
