#!/bin/bash
# development loop of the C2 kernel: bit-identity tests of the batched launches, then the driver-setting and steady-state lines (+ tick stamps)
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/c2_iter.log
: > $O
if [ "${1:-}" != "notest" ]; then timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_fused.py -x -q 2>&1 | tail -3 >> $O; fi
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 20/5', round(d['value'],1), round(d['steady_state']['value'],1), round(d['roofline']['frac'],4))" >> $O; done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$O.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 200/20', round(d['value'],1), round(d['steady_state']['value'],1), round(d['roofline']['frac'],4))" >> $O
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --option psd_debug=2 2>&1 | grep "cu debug" | tail -2 >> $O
cat $O
