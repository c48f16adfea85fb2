import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import skeres_amd as sk
from skeres_amd import bal
from helpers import bal_problem_to_sk
import torch
for (C, P, N, seed) in ((400, 30000, 140000, 77), (800, 60000, 300000, 5), (1200, 100000, 450000, 9)):
    prob = bal.generate(C, P, N, seed=seed)
    problem, params, loss = bal_problem_to_sk(prob)
    o = sk.Solver.Options(); o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    o.setMaxNumIterations(1000); o.setFunctionTolerance(0.0); o.setGradientTolerance(0.0); o.setParameterTolerance(0.0)
    s = sk.StepSolver(o, problem)
    for _ in range(5): s.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): s.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print("C=%d: dissected %d, %.3f ms per iteration (model %.0f -> %.0f us)" % (C, s.stat("dissected"), 1e3 * dt, s.stat("dissection_model_us_plain"), s.stat("dissection_model_us")))
