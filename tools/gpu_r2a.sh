set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_psd.py tests/test_gpu_ops.py -x -q 2>&1 | tail -15
timeout 300 python tools/probe_sign.py 32 10000 2>&1 | tail -8
timeout 300 python tools/probe_sign.py 27 3000 2>&1 | tail -8
timeout 300 python tools/probe_sign.py 45 3000 2>&1 | tail -8
timeout 300 python tools/probe_sign.py 64 2000 2>&1 | tail -8
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -3
for w in 2 4; do CUADMM_PSD_W32_WPG=$w timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1; done
