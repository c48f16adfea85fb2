#!/usr/bin/env python3
"""scratch: Ladybug three iterations with a given plan; prints the costs.  argv: retained=on|off dissection=auto|off"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import skeres_amd as sk
from skeres_amd import bal
from helpers import bal_problem_to_sk
kw = dict(a.split("=") for a in sys.argv[1:])
prob = bal.generate_named("ladybug-1723-156502", seed=1723, perturb=(1e-2, 1e-1, 1e-1))
problem, params, loss = bal_problem_to_sk(prob)
o = sk.Solver.Options()
o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
o.setMaxNumIterations(3)
if kw.get("retained") == "on":
    o.setRetainedPoints("on", 12)
else:
    o.setRetainedPoints("off")
o.setCholeskyDissection(kw.get("dissection", "auto"))
solver = sk.StepSolver(o, problem)
d = solver.stat("dissected")
if kw.get("timing"):
    solver.setKernelTiming(int(kw["timing"]))  # 1: every kernel timed -> the same plan launch by launch
while not solver.step():
    pass
s = sk.Solver.Summary(); solver.finish(s)
print(" ".join(sys.argv[1:]), "| withhold", os.environ.get("SK_CHAIN_TEST_WITHHOLD_MARKER"), "bs_resident", os.environ.get("SK_BS_RESIDENT"), "| dissected", d,
      "| costs", ["%.10e" % it["cost"] for it in s.iterations()], flush=True)
