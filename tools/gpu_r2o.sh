cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_psd.py -x -q 2>&1 | tail -2
for occ in 4 3 4 3; do CUADMM_PSD_W32_OCC=$occ timeout 300 python bench.py --no-cpu-baseline --no-breakdown 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('occ $occ', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"; done
CUADMM_PSD_W32_OCC=4 timeout 300 python tools/probe_sign.py 32 10000 2>&1 | tail -5
