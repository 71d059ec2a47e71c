cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/run_all_real.py 2>&1 | grep "PlanarHand\|pendulum\|PushT\|ros_2000\|1dc" | cut -c1-360
