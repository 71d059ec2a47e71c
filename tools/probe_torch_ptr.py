import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cuadmm_amd
from tests.helpers import Dev
lib = cuadmm_amd.load()     # HIP runtime from /opt/rocm first
d = Dev(np.arange(8, dtype=np.float64))
import torch
print("torch", torch.__version__, torch.cuda.is_available())
class _Ptr:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (count,), "typestr": "<f8", "version": 2}
t = torch.as_tensor(_Ptr(d.ptr.value, 8), device="cuda:0")
print(t)
t += 1
torch.cuda.synchronize()
print(d.get())
import subprocess
print(subprocess.run("cat /proc/%d/maps | grep -E 'amdhip|rccl' | awk '{print $6}' | sort -u" % os.getpid(), shell=True, capture_output=True, text=True).stdout)
