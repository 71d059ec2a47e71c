set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_f4_free_and_rank.py tests/test_gpu_mex.py tests/test_gpu_psd.py -x -q 2>&1 | tail -8
timeout 600 python tools/probe_eig_large.py 256 512 1024 2000 2>&1 | tail -6
