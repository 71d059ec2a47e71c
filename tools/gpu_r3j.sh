#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
python tools/probe_rampup.py 64
python tools/probe_rampup.py 0
