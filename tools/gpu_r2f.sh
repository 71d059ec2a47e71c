set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_psd.py -x -q 2>&1 | tail -3
for occ in 4 3; do for wpg in 1 4; do
echo "OCC $occ WPG $wpg"
CUADMM_PSD_W32_OCC=$occ CUADMM_PSD_W32_WPG=$wpg CUADMM_PSD_DEBUG=1 timeout 300 python tools/probe_psd.py 32 10000 2 2>&1 | grep "psd debug" | tail -1
CUADMM_PSD_W32_OCC=$occ CUADMM_PSD_W32_WPG=$wpg timeout 300 python bench.py --no-cpu-baseline --no-breakdown 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
done; done
timeout 300 python tools/probe_sign.py 32 10000 2>&1 | tail -5
timeout 300 python tools/probe_sign.py 21 3000 2>&1 | tail -5
