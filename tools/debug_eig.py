import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import batch_eig_gpu
n = int(sys.argv[1]); count = int(sys.argv[2])
rng = np.random.default_rng(0)
A = rng.standard_normal((count, n, n)); A = (A + np.swapaxes(A, 1, 2)) / 2
W, V, info = batch_eig_gpu(A)
print("n", n, "info", info[:8], "maxdiff W", np.max(np.abs(W - np.linalg.eigvalsh(A))))
