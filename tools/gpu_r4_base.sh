#!/bin/bash
# round-4 baseline: driver-setting C2 line, steady-state line, tick stamps of the persistent closed-block kernel, C4 line
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r4_base.log
: > $O
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 20/5', d['value'], d.get('steady_state'), d['roofline']['frac'])" >> $O; done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$O.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 200/20', d['value'], d.get('steady_state'), d['roofline']['frac'])" >> $O
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --option psd_debug=2 2>&1 | grep "cu debug" | tail -3 >> $O
timeout 300 python bench.py --config c4 --no-cpu-baseline 2>>$O.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', d['value'], d.get('steady_state'), d['roofline']['frac'])" >> $O
cat $O
