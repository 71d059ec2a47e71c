#!/bin/bash
# smoke() and the quick bench lines on the tree as it is
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 driver setting', round(d['value'],1), round(d['roofline']['frac'],4))"
timeout 300 python bench.py --config c1 --no-cpu-baseline --no-breakdown 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c1', round(d['value'],1))"
timeout 600 python -m pytest tests/test_gpu_moment_parity.py -q -k "wide_forest or dense_tree_tops or falls_back" 2>&1 | tail -1
