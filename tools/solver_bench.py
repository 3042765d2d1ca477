"""Diagnostic: time of one dependent pose update (solve + exponential + pose product) on one wavefront."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from slambench_amd import native
L = native.lib()
L.kfh_debug_solver_us.argtypes = [C.c_void_p, C.c_void_p]
rng = np.random.default_rng(0)
def system(eigs):
    q, _ = np.linalg.qr(rng.standard_normal((6, 6)))
    Cm = (q * np.asarray(eigs, np.float64)) @ q.T
    b = Cm @ (1e-3 * rng.standard_normal(6))
    s = np.zeros(32, np.float32); s[0] = 1; s[1:7] = b; k = 7
    for i in range(6):
        for j in range(i, 6):
            s[k] = Cm[i, j]; k += 1
    s[28] = 1000
    return s
for name, eigs in (("well conditioned", [2e2, 1e3, 4e4, 5e4, 1e5, 6e5]), ("smallest dropped", [1.5, 40, 300, 5e3, 2e5, 2.2e6])):
    s = system(eigs)
    out = C.c_float()
    native.check(L.kfh_debug_solver_us(s.ctypes.data_as(C.c_void_p), C.byref(out)))
    print("%-18s %.3f us per update" % (name, out.value))
