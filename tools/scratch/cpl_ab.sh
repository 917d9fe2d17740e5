#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python tools/bench_coupling.py 2>/dev/null; done
