"""Time of the reduced system's back-substitution on Ladybug-1723 / Venice-1778 (HIP events around it: kernel timing mode 1).
SK_BS_RESIDENT=0: one launch per block step; default: one resident launch.  python tools/time_backsolve.py [workload]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import skeres_amd as sk  # noqa: E402
from skeres_amd import bal  # noqa: E402
from helpers import bal_problem_to_sk  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ladybug-1723-156502"
prob = bal.generate_named(name, seed=1723, perturb=(1e-2, 1e-1, 1e-1))
problem, params, loss = bal_problem_to_sk(prob)
o = sk.Solver.Options()
o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
o.setMaxNumIterations(1000)
o.setFunctionTolerance(0.0); o.setGradientTolerance(0.0); o.setParameterTolerance(0.0)
s = sk.StepSolver(o, problem)
for _ in range(3):
    s.step()
s.setKernelTiming(1)
for _ in range(10):
    s.step()
sec, n = s.kernelSeconds("backsolve")
print("%s SK_BS_RESIDENT=%s: back-substitution of the reduced system %.3f ms per solve (%d solves)" % (name, os.environ.get("SK_BS_RESIDENT", "1"), 1e3 * sec / max(1, n), n))

if os.environ.get("SK_BS_STAMPS"):
    import numpy as np
    st = np.loadtxt(os.environ["SK_BS_STAMPS"])
    kb, wait0, got, stored = st[:, 0].astype(int), st[:, 1] * 0.01, st[:, 2] * 0.01, st[:, 3] * 0.01
    t0 = stored[0]
    print("block column: microseconds from the first stored y: started waiting for y of the next column / had it / own y stored | hop = stored - stored(next column), transfer = had it - stored(next), compute = stored - had it")
    for i in range(1, len(kb)):
        print("%4d  %8.2f %8.2f %8.2f | hop %5.2f  transfer %5.2f  compute %5.2f" % (kb[i], wait0[i] - t0, got[i] - t0, stored[i] - t0, stored[i] - stored[i - 1], got[i] - stored[i - 1], stored[i] - got[i]))
    hop = stored[1:] - stored[:-1]
    print("mean hop %.2f us, transfer %.2f, compute %.2f" % (hop.mean(), (got[1:] - stored[:-1]).mean(), (stored[1:] - got[1:]).mean()))
