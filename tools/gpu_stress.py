import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_check import *
ok = True
t0=time.time()
ok &= check_sequence(B=96, N=100, steps=40, seed=11, full=False)
print("t", time.time()-t0)
ok &= check_sequence(B=4, N=100, steps=25, seed=19, full=True)
ok &= check_sequence(B=6, N=500, steps=4, seed=3, full=True, turn=(0.3, 9.0))
ok &= check_sequence(B=3, N=7, steps=12, seed=4, full=True, turn=(0.05, 0.2))
print("t", time.time()-t0)
print("STRESS", "OK" if ok else "FAIL")
sys.exit(0 if ok else 1)
