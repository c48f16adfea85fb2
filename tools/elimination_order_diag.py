#!/usr/bin/env python3
"""Developer tool: the first three LM steps of Ladybug-1723 (or of `C,P,N,seed`) under elimination orders that are equal in exact
arithmetic — every point eliminated / twelve retained, one front / three — and how far their costs are apart (profiles/r04_elimination_order_rounding.txt)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import skeres_amd as sk
from skeres_amd import bal
from helpers import bal_problem_to_sk

which = sys.argv[1] if len(sys.argv) > 1 else "ladybug"
if which == "ladybug":
    prob = bal.generate_named("ladybug-1723-156502", seed=1723, perturb=(1e-2, 1e-1, 1e-1))
else:
    C, P, N, seed = [int(v) for v in which.split(",")]
    prob = bal.generate(C, P, N, seed=seed)

def run(**kw):
    problem, params, loss = bal_problem_to_sk(prob)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    o.setMaxNumIterations(3)
    for k, v in kw.items():
        getattr(o, k)(*v) if isinstance(v, tuple) else getattr(o, k)(v)
    solver = sk.StepSolver(o, problem)
    st = {k: solver.stat(k) for k in ("retained_points", "dissected", "envelope_fill", "cholesky_columns_resident", "dissection_head_cameras", "dissection_separator_cameras")}
    while not solver.step():
        pass
    s = sk.Solver.Summary(); solver.finish(s)
    return [it["cost"] for it in s.iterations()], st

ref, st = run(setRetainedPoints="off", setCholeskyDissection="off")
print("reference (all eliminated, undissected)", ["%.12e" % c for c in ref], st, flush=True)
variants = {
    "retained, undissected": dict(setRetainedPoints=("on", 12), setCholeskyDissection="off"),
    "retained, dissected (lock-step)": dict(setRetainedPoints=("on", 12)),
    "retained, dissected, no resident kernels": dict(setRetainedPoints=("on", 12), setResidentKernels=False),
    "all eliminated, dissected": dict(setRetainedPoints="off"),
}
for name, kw in variants.items():
    c, st = run(**kw)
    print("%-45s" % name, ["%.3e" % abs(a / b - 1.0) for a, b in zip(c, ref)], st, flush=True)
