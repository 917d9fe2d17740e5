#!/bin/bash
# round-6 extras on the GPU box: cfg5 lines (fp16 / fp32 mixes), the one-rank RCCL line, the trainer's window (default / eager), the
# metric configuration's window, the adversarial bf16x3 numbers
cd $GRAFT_REPO_ROOT
python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/r6_bench_cfg5_B64_fp16_mix.json 2> gpurun_out/r6_cfg5_f16.err
python bench.py --config cfg5 --mix f32 --no-cpu-baseline > gpurun_out/r6_bench_cfg5_B64_fp32_mix.json 2> gpurun_out/r6_cfg5_f32.err
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --gpus 1 --force-bucket --no-cpu-baseline > gpurun_out/r6_bench_config_M_one_rank_rccl.json 2> gpurun_out/r6_rccl1.err
python tools/window_run.py > gpurun_out/r6_window_trainer_256x256x3_B64_default.json 2> gpurun_out/r6_win_a.err
python tools/window_run.py --eager --no-adopt > gpurun_out/r6_window_trainer_256x256x3_B64_eager_torch_adam.json 2> gpurun_out/r6_win_b.err
python tools/window_run.py --noc 4 --adam hip --capture > gpurun_out/r6_window_config_M_B64_captured.json 2> gpurun_out/r6_win_c.err
python tools/window_run.py --noc 4 --adam hip --eager > gpurun_out/r6_window_config_M_B64_eager.json 2> gpurun_out/r6_win_d.err
python -m pytest tests/test_hip_ops.py -q -m gpu -s -k "adversarial" 2>&1 | grep "bf16x3 adversarial" > gpurun_out/r6_bf16x3_adversarial.txt
for f in gpurun_out/r6_bench_cfg5_B64_fp16_mix.json gpurun_out/r6_bench_cfg5_B64_fp32_mix.json gpurun_out/r6_bench_config_M_one_rank_rccl.json gpurun_out/r6_window_*.json; do echo "== $f"; tail -1 $f | cut -c1-400; done
cat gpurun_out/r6_bf16x3_adversarial.txt
