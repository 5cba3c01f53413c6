#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q -x -k "persistent_tile or budgeted or more_iterations" 2>&1 | tail -3
timeout 300 python tools/tile_budget_sweep.py 0 2>&1 | tail -1
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed"
timeout 300 python tools/bench_configs.py C4 2>&1 | grep -v amdgpu | cut -c1-200
