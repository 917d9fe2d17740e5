#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/bench_wino.py 2>/dev/null | sed "s/bf16x3.*direct/direct/; s/(few-output kernel, fp32)//" | cut -c1-75 | tail -7
