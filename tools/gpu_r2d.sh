set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_psd.py tests/test_gpu_configs.py -x -q 2>&1 | tail -5
timeout 300 python tools/probe_sign.py 100 300 2>&1 | tail -8
timeout 300 python tools/run_config.py c3 2000 100 2>&1 | grep "RESULT\|psd_project"
CUADMM_PSD_SIGN_SYNC=0 timeout 300 python tools/run_config.py c3 2000 100 2>&1 | grep "RESULT\|psd_project"
CUADMM_PSD_DEBUG=1 timeout 300 python tools/run_config.py c3 2000 5 2>&1 | grep "psd debug" | tail -2
timeout 900 python tools/run_all_real.py 2>&1 | tail -12
