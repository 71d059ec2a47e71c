"""Two ranks on one GPU for a moment fixture under environment variants, against the committed oracle trajectory.
python tools/dbg/two_rank_bisect.py taha1a 'CUADMM_TAIL_K=0' 'CUADMM_PSD_LG_MERGE=0' ..."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1]
rec = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_traj_moment.json")))[name + "/switch=11000"]
port = 29700
for variant in [""] + sys.argv[2:]:
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for kv in variant.split():
        k, v = kv.split("=")
        env[k] = v
    out = os.path.join(tempfile.mkdtemp(), "res.npz")
    port += 1
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_sharded_worker.py"), out, name], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode:
        print(variant or "default", "FAILED", r.stderr[-800:])
        continue
    d = np.load(out)
    devs = {}
    for nm in ("errRp", "errRd", "pobj", "dobj", "relgap"):
        ref = np.array([float(x) for x in rec[nm]])
        got = d[nm][:ref.size]
        devs[nm] = float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)))
        first = int(np.argmax(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300) > 1e-6)) if devs[nm] > 1e-6 else -1
        devs[nm] = (devs[nm], first)
    print(variant or "default", "shard", d["shard"], "counters", d["counters"], devs)
