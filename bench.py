#!/usr/bin/env python3
"""LM iterations/sec on a synthetic BAL Ladybug-1723-shaped problem (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N>1 with torch.distributed.run).  A
"step" is ONE Levenberg-Marquardt iteration of the DENSE_SCHUR solve: LM
diagonal, Schur-complement assembly, dense fp64-MFMA Cholesky of the 9C x 9C
reduced camera system, back-substitution, candidate cost, accept/reject and
(on acceptance) the Jet-autodiff Jacobian evaluation at the new point.
Termination tolerances are set to zero so exactly K iterations run in the timed
region of a real solve; how many of them were successful steps (an unsuccessful one skips the Jacobian evaluation) is
reported in config.

Multi-GPU (total work fixed => "strong"; the solver chooses, config.parallelism says what it chose): SEGMENTED — the
camera sequence is cut into as many segments as pay (at most one per rank), rank r's device eliminates segment r and its
points, the separators' block-tridiagonal system is all-reduced (a few MB) and factored by every rank; or SHARDED — points
sharded, the reduced system (its blocks inside the envelope) all-reduced, Cholesky replicated; or REPLICATED when neither
pays.  The same line carries the other configurations of BASELINE.json: one GPU — `c2` (BAL-49), `c4` (Venice-1778), `c5`
(dense rows); several ranks — `venice` and `c5` (rows sharded), each with the distribution used, the bytes all-reduced per
iteration and the milliseconds that takes; and on one GPU `predicted_multi_gpu`, the chain model's figures for 2, 4, 8.

stdout carries ONE line: the compact headline (at most 4 KB — metric, value, config, `roofline`, `cpu_baseline`), printed by
rank 0 as soon as it is complete.  Everything else this run measures (the headline's detailed record and the other
configurations) goes to stderr, one compact JSON line per record, and to a side file (SK_BENCH_DETAILS, default
gpurun_out/bench_details.json): an extra configuration can neither cost the headline its line nor make it unparseable.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (datasheet; MI355X_MICROARCH.md has no fp64 row)
PERTURB = (1e-2, 1e-1, 1e-1)  # angle-axis, translation, point perturbation of the initial guess
SEED = 1723


def available_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host, e.g. 256, on a box whose share is 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
        except (OSError, ValueError, IndexError):
            pass
    cap = os.environ.get("SKERES_CPU_THREADS")  # optional cap; by default every CPU of the process's share is used
    return max(1, min(n, int(cap))) if cap else max(1, n)


def build_problem(sk, prob):
    params = sk.RichDoubleArray.fromArray(prob.parameters)
    problem = sk.Problem()
    loss = sk.PredefinedLossFunctions.trivialLoss()
    offs = np.stack([9 * prob.camera_index.astype(np.int64),
                     9 * prob.num_cameras + 3 * prob.point_index.astype(np.int64)], axis=1)
    problem.addResidualBlocks(sk.SnavelyReprojectionError.FUNCTOR_ID, prob.observations, loss, params, offs)
    return problem, params, loss


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*pmc_traffic.json, made by
    tools/pmc_summary.py from two separate rocprofv3 --pmc runs of this script with --no-resident-kernels), LIKE FOR LIKE: the
    counter bytes per SYRK launch of the plan those passes ran, beside the algorithmic C-tile bytes per launch of THAT plan
    (from the passes' own bench line).  PMC counters cannot be read from inside the timed run: None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        lfl = d.get("syrk_like_for_like")
        if lfl:
            return {"bytes_per_launch": lfl["counter_bytes_per_launch"], "algorithmic_bytes_per_launch": lfl["algorithmic_c_tile_bytes_per_launch"],
                    "ratio": lfl["ratio"], "plan": lfl.get("plan"), "launches_per_iteration": lfl.get("launches_per_iteration"),
                    "per_iteration": d.get("hbm_bytes_per_iteration"), "algorithmic_per_iteration": d.get("algorithmic_bytes_per_iteration"),
                    "file": os.path.basename(files[-1])}
        return None  # (a summary of an earlier round: its per-launch figure belongs to another plan than the one it was printed beside)
    except (KeyError, ValueError, OSError):
        return None


def cpu_baseline(prob, iters, envelope=True, full_iters=1):
    """The oracle (CPU restatement, kind "port") on the same problem and — like for like — the same plan: the
    Cholesky of the reduced camera system skips the structural zeros outside its envelope exactly as the GPU
    factorisation does (oracle/chol.cpp: bit-identical to factoring everything).  `iters` LM iterations on all host
    cores, value = 1 / median iteration time; the full factorisation (no envelope) is timed beside it on `full_iters`
    iterations (it costs ~7x the work)."""
    import oracle
    nthreads = available_cpus()

    def run(n_it, env):
        o = oracle.default_options(linear_solver_type=oracle.DENSE_SCHUR, num_threads=nthreads, max_num_iterations=n_it,
                                   function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0,
                                   cholesky_envelope=1 if env else 0)
        t0 = time.time()
        _, s = oracle.solve_bal(prob.num_cameras, prob.num_points, prob.camera_index, prob.point_index,
                                prob.observations, prob.parameters, o)
        return s, time.time() - t0
    s, wall = run(iters, envelope)
    times = sorted(s.iteration_seconds())
    med = times[len(times) // 2] if times else float("nan")
    out = {"value": 1.0 / med, "unit": "LM iterations/s", "cores": int(s.num_threads_used), "kind": "port",
           "plan": "block envelope of the reduced camera system (as the GPU run)" if envelope else "full factorisation (as the GPU run)",
           "sample": "median of %d LM iterations of the same bundle-adjustment-shaped problem (CPU restatement, not Ceres; %.1f s wall; "
                     "cholesky %.1f s, schur assembly %.1f s in total)" % (len(times), wall, s.t_linear_cholesky_s, s.t_linear_assemble_s)}
    if envelope and full_iters > 0:
        sf, wall_f = run(full_iters, False)
        tf = sorted(sf.iteration_seconds())
        out["value_full_factorisation"] = 1.0 / tf[len(tf) // 2]
        out["sample_full_factorisation"] = "%d iteration(s), %.1f s wall (cholesky %.1f s)" % (len(tf), wall_f, sf.t_linear_cholesky_s)
    return out


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E, ~8 TB/s


def phase_rooflines(phases_ms, n_obs, envelope_bytes):
    """HBM rooflines of the phases around the factorisation: SURVEY.md section 8(d)'s ALGORITHMIC bytes per observation
    (A: 328 B — 120 read, 208 written; B: 208 B read + the blocks of the reduced system written once, here the blocks inside
    the envelope; D: 136 B — back-substitution and cost-only evaluation) over the measured phase time."""
    def rec(nbytes, ms, what):
        gbs = nbytes / (ms * 1e-3) * 1e-9 if ms and ms > 0 else None
        return {"bound": "hbm", "bytes": nbytes, "ms": ms, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS if gbs else None, "what": what}
    return {"A_jacobian_eval": rec(328.0 * n_obs, phases_ms.get("jacobian_eval"), "328 B/observation (per accepted step)"),
            "B_schur_assemble": rec(208.0 * n_obs + envelope_bytes, phases_ms.get("schur_assemble"), "208 B/observation read + the reduced system's blocks inside the envelope written once"),
            "D_backsub_cost": rec(136.0 * n_obs, (phases_ms.get("back_substitute") or 0.0) + (phases_ms.get("cost_eval") or 0.0), "136 B/observation")}


PHASE_NAMES = ["jacobian_eval", "schur_assemble", "cholesky", "back_substitute", "cost_eval", "allreduce"]


def phase_snapshot(solver):
    """Seconds the solver has accumulated per phase so far (HIP events on its stream; sk_solver_stat "phase_seconds_<i>")."""
    return [solver.stat("phase_seconds_%d" % i) for i in range(len(PHASE_NAMES))]


def phases_per_step(before, after, steps):
    """Milliseconds per step of each phase over the timed region alone.  Under hipGraph replay (BAL-49) the solver has no events
    inside the replayed graph and books the whole linear solve + candidate evaluation under "cholesky": the sum of the phases
    still is what the stream spent per step, and can no longer exceed ms_per_step (round-4 verdict: it included the captures)."""
    return {k: 1e3 * (after[i] - before[i]) / max(1, steps) for i, k in enumerate(PHASE_NAMES)}


HEADLINE_MAX_BYTES = 4096


def compact(x, digits=5):
    """Floats to `digits` significant digits, recursively (the headline is for a parser and a reader, not for arithmetic)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: compact(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [compact(v, digits) for v in x]
    return x


def headline_line(d):
    """The one stdout line: the contract's fields, `roofline` and `cpu_baseline`, nothing else — from the detailed record `d` that
    main() assembles (its keys are the same with or without a GPU: tests/test_bench_line_cpu.py builds one from canned figures).
    `roofline` describes what bounds the iteration: the Cholesky PHASE (every kernel of the factorisation and the triangular solves —
    flops of the plan over the phase's duration by HIP events on the solver's stream) against the fp64 matrix peak, the serial chain
    inside it (steps x microseconds per step; the resident potrf server is one launch per factorisation and has no roofline of its own:
    it is latency-bound), and the trailing SYRK's launches as a sub-record."""
    cfg = d["config"]
    chol = d.get("roofline_cholesky_phase") or {}
    syrk = d.get("roofline_syrk") or {}
    full = d.get("roofline_full") or {}
    steps = cfg.get("chain_steps") or 0
    roof = {"bound": "mfma", "kernel": "Cholesky phase of the reduced camera system (all its kernels: potrf server, column launches, trailing SYRKs, triangular solves)",
            "achieved": chol.get("achieved"), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": chol.get("frac"),
            "flops_per_iteration": chol.get("flops"), "ms_per_iteration": chol.get("ms"),
            # HBM bytes by the counters cannot be read inside a timed run; the committed passes are in syrk.traffic, with the plan they measured
            "traffic": None,
            "chain": {"steps": steps, "us_per_step": (1e3 * chol["ms"] / steps) if steps and chol.get("ms") else None,
                      "model_us": (d.get("chain_model") or {}).get("model_us"), "measured_over_model": (d.get("chain_model") or {}).get("measured_over_model"),
                      "what": "block columns in sequence (two leaf fronts in lock-step count once), each potrf128 + two tile round trips + hand-overs; latency-bound"},
            "syrk": {"kernel": "sk::syrk_trailing(_thin)_f64_kernel", "achieved": syrk.get("achieved"), "frac": syrk.get("frac"),
                     "launches_per_iteration": syrk.get("launches_per_iteration"), "avg_launch_us": 1e3 * syrk["avg_launch_ms"] if syrk.get("avg_launch_ms") else None,
                     "flops_per_launch": (syrk.get("flops_per_solve") or 0.0) / max(1, syrk.get("launches_per_iteration") or 1),
                     "frac_alone": syrk.get("frac_alone"), "traffic": syrk.get("traffic"),
                     "traffic_algorithmic": (syrk.get("traffic_like_for_like") or {}).get("algorithmic_bytes_per_launch"),
                     "traffic_from": (syrk.get("traffic_like_for_like") or {}).get("file")},
            "full_factorisation": ({"achieved": full.get("achieved"), "frac": full.get("frac"), "ms_per_step": full.get("ms_per_step")} if full else None)}
    cpu = d.get("cpu_baseline")
    if cpu:
        cpu = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "sample", "value_full_factorisation") if cpu.get(k) is not None}
        if len(cpu.get("sample", "")) > 260:
            cpu["sample"] = cpu["sample"][:257] + "..."
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": cfg["workload"], "linear_solver": cfg.get("linear_solver"), "plan": cfg.get("plan"), "chain_steps": steps,
                      "retained_points": (cfg.get("retained_points") or {}).get("points"), "envelope_fill": cfg.get("envelope_fill"),
                      "successful_steps_in_timed_region": cfg.get("successful_steps_in_timed_region"), "parallelism": cfg.get("parallelism_short") or cfg.get("parallelism")}
    line["roofline"] = roof
    line["phases_ms_per_step"] = d.get("phases_ms_per_step")
    if d.get("distribution"):
        line["distribution"] = d["distribution"]
    if d.get("independent_solves"):
        line["independent_solves"] = {k: d["independent_solves"][k] for k in ("value", "unit", "scaling")}
    line["cpu_baseline"] = cpu
    line["details"] = d.get("details_file")
    line = compact(line)
    text = json.dumps(line, separators=(",", ":"))
    # (never over the cap: drop the optional sub-records first, then shorten the strings)
    for drop in (("roofline", "full_factorisation"), ("roofline", "syrk"), ("phases_ms_per_step",), ("independent_solves",)):
        if len(text) <= HEADLINE_MAX_BYTES:
            break
        tgt = line
        for k in drop[:-1]:
            tgt = tgt.get(k) or {}
        tgt.pop(drop[-1], None)
        text = json.dumps(line, separators=(",", ":"))
    assert "\n" not in text and len(text) <= HEADLINE_MAX_BYTES, len(text)
    return text


REVISITS = [(200, 900, 40, 150), (450, 1300, 40, 150), (700, 1600, 40, 150)]  # three places seen twice, 40 cameras and 150 tracks each


def all_ranks_ok(ok, world, dist_mod, torch, what):
    """Failure is collective: every rank learns whether every rank got this far, so that all of them skip a record together
    instead of one of them leaving the others inside a collective."""
    if world > 1:
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cuda")
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MIN)
        if float(t.item()) < 1.0 and ok:
            raise RuntimeError("another rank failed at: %s" % what)


def chain_model_record(stat, cholesky_ms):
    """The chain model's figure for the factorisation plan THIS solver runs (microseconds per block column: the larger of its panel
    chain and its trailing update — choose_dissection / choose_border, bal_solver.hip) beside the measured Cholesky phase, which
    also holds the back-substitution (~0.4 ms at Ladybug size) the model does not count.  The model decides the dissection, the
    border and the number of segments of a world of ranks; the ratio says how far it is off on this problem (round-3 verdict,
    item 4d: 0.85-0.9 on the Ladybug shape, i.e. it over-estimates; it was never fitted to the mid-size bands)."""
    def get(k):
        try:
            return stat(k)
        except Exception:  # noqa: BLE001
            return 0.0
    which, model = "undissected", get("model_us_segments_1")
    kept = " with %d retained points" % int(get("retained_points")) if get("retained_points") > 0 else ""
    if get("dissected") == 1.0 and get("dissection_model_us") > 0:
        which, model = "lock-step dissection" + kept, get("dissection_model_us")
    elif get("retained_points") > 0 and get("retained_model_us") > 0:
        which, model = "bordered" + kept, get("retained_model_us")
    elif get("border_cameras") > 0 and get("border_model_us") > 0:
        which, model = "bordered", get("border_model_us")
    if not model:
        return None
    return {"plan": which, "model_us": model, "measured_cholesky_phase_us": 1e3 * cholesky_ms,
            "measured_over_model": 1e3 * cholesky_ms / model if model > 0 else None}


def bal_record(sk, bal, name, seed, steps, warmup, local_rank, stream, rank, world, dist_mod=None, torch=None, long_range_fraction=0.0,
               revisits=(), border=None, retained=None):
    """One more bundle-adjustment workload of BASELINE.json in the same run (configs[1] BAL-49, configs[3] Venice-1778): `steps`
    LM iterations of the same solve as the headline, timed the same way (barrier + synchronise on both sides, max over
    ranks); with several ranks, the distribution the solver chose, what travels per iteration and how long it takes."""
    prob = bal.generate_named(name, seed=seed, perturb=PERTURB, long_range_fraction=long_range_fraction, revisits=revisits)
    problem, params, loss = build_problem(sk, prob)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    o.setMaxNumIterations(warmup + steps + 1000)
    o.setFunctionTolerance(0.0)
    o.setGradientTolerance(0.0)
    o.setParameterTolerance(0.0)
    o.setDevice(local_rank)
    o.setStream(stream.cuda_stream)
    if border is not None:
        o.setCholeskyBorder(border)
    if retained is not None:
        o.setRetainedPoints(retained)
    hook = None
    solver, err = None, None
    try:
        if world > 1:
            from skeres_amd import dist as sk_dist
            hook = sk_dist.attach(o, problem, rank, world)
        solver = sk.StepSolver(o, problem)
    except Exception as e:  # noqa: BLE001
        err = e
    all_ranks_ok(err is None, world, dist_mod, torch, "set-up of %s" % name)
    if err is not None:
        raise err
    mode = solver.distribution()[0] if world > 1 else "single"
    stats = {k: solver.stat(k) for k in ("envelope_fill", "allreduce_bytes", "segments", "cholesky_flops_plan", "border_cameras", "border_model_us",
                                         "border_model_us_plain", "cholesky_columns_resident", "dissected", "model_us_segments_1", "dissection_model_us",
                                         "retained_points", "retained_model_us", "retained_model_us_without")}
    for _ in range(warmup):
        solver.step()
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    phases0 = phase_snapshot(solver)
    t0 = time.perf_counter()
    done = 0
    for _ in range(steps):
        done += 1
        if solver.step():  # (terminated: e.g. the trust region collapsed on a converged small problem)
            break
    steps = done
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
        elapsed = float(t.item())
    phases = phases_per_step(phases0, phase_snapshot(solver), steps)  # the timed steps only (not set-up, warm-up or a graph's capture)
    summ = sk.Solver.Summary()
    solver.finish(summ)
    its = summ.iterations()
    chol_s = phases["cholesky"] * 1e-3
    rec = {"workload": "BAL %s (synthetic, shape-exact: C=%d P=%d N=%d, seed %d%s%s), DENSE_SCHUR" % (
               name, prob.num_cameras, prob.num_points, prob.num_observations, seed, ", long_range_fraction %g" % long_range_fraction if long_range_fraction else "",
               ", revisits (first camera a, first camera b, cameras, tracks) %s" % (list(revisits),) if revisits else ""),
           "n_gpus": world, "steps": steps, "ms_per_step": 1e3 * elapsed / steps, "iterations_per_second": steps / elapsed,
           "phases_ms_per_step": phases, "envelope_fill": stats["envelope_fill"],
           "roofline_cholesky_phase": {"bound": "mfma", "flops": stats["cholesky_flops_plan"], "achieved": stats["cholesky_flops_plan"] / chol_s * 1e-12 if chol_s > 0 else None,
                                       "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": stats["cholesky_flops_plan"] / chol_s * 1e-12 / FP64_MFMA_PEAK_TFLOPS if chol_s > 0 else None},
           "costs": [its[0]["cost"], its[-1]["cost"]]}
    if revisits or long_range_fraction:
        rec["border"] = {"mode": border or "auto", "cameras": int(stats["border_cameras"]), "chain_model_us": stats["border_model_us"],
                         "chain_model_us_plain_order": stats["border_model_us_plain"], "block_columns_resident": int(stats["cholesky_columns_resident"]),
                         "dissected": int(stats["dissected"])}
    # retained points (sk_options_set_retained_points): the widest tracks stay in the reduced system instead of being eliminated
    rec["retained_points"] = {"mode": retained or "auto", "points": int(stats["retained_points"]), "chain_model_us": stats["retained_model_us"],
                              "chain_model_us_all_eliminated": stats["retained_model_us_without"], "dissected": int(stats["dissected"]),
                              "block_columns_resident": int(stats["cholesky_columns_resident"])}
    if world == 1:
        rec["chain_model"] = chain_model_record(lambda k: stats.get(k, 0.0), phases["cholesky"])
    if world > 1:
        rec["distribution"] = {"mode": mode, "segments": int(stats["segments"]), "allreduce_bytes_per_iteration": stats["allreduce_bytes"],
                               "allreduce_ms_per_step": phases["allreduce"]}
    else:
        rec["roofline_phases_hbm"] = phase_rooflines(phases, prob.num_observations, stats["allreduce_bytes"])
    del solver
    return rec


def c5_record(sk, m=1000000, n=10000, iters=2, seed=5, rank=0, world=1, stream=None, torch=None, dist_mod=None):
    """BASELINE.json config 5 (dense rows: m residuals x n parameters, DENSE_NORMAL_CHOLESKY) in the same run: the J^T J
    formation (syrk_gram_f64_kernel, fp64 MFMA) timed by HIP events around its launches, priced with the ALGORITHMIC
    flop count m n (n + 1) of SURVEY.md section 8(d) (not the padded tiles the launch computes)."""
    rng = np.random.default_rng(seed)
    x_star = rng.normal(size=n)
    y = sk.api.synth_dense_targets(seed, m, n, x_star) + rng.normal(0, 1e-3, m)
    consts = np.stack([np.full(m, float(seed)), np.arange(m, dtype=np.float64), y], axis=1)
    x = sk.DoubleArray(n)
    problem = sk.Problem()
    problem.addDenseRows(10, consts, None, x, n)
    o = sk.Solver.Options()
    o.setLinearSolverType(sk.LinearSolverType.DENSE_NORMAL_CHOLESKY)
    o.setMaxNumIterations(1000)
    o.setFunctionTolerance(0.0)
    o.setGradientTolerance(0.0)
    o.setParameterTolerance(0.0)
    hook = None
    if world > 1:  # the rows are sharded over the ranks, J^T J all-reduced, the Cholesky replicated
        from skeres_amd import dist as sk_dist
        o.setStream(stream.cuda_stream)
        hook = sk_dist.TorchAllReduce()
        o.setDistributed(rank, world, hook)
    s, err = None, None
    try:
        s = sk.StepSolver(o, problem)
    except Exception as e:  # noqa: BLE001 (e.g. not enough free HBM for this rank's rows of the Jacobian on a shared device)
        err = e
    all_ranks_ok(err is None, world, dist_mod, torch, "set-up of the dense-rows problem")
    if err is not None:
        raise err
    s.setKernelTiming(1)
    if world > 1:
        torch.cuda.synchronize()
        dist_mod.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        s.step()
    if world > 1:
        torch.cuda.synchronize()
        dist_mod.barrier()
    dt = (time.perf_counter() - t0) / iters
    sec, launches = s.kernelSeconds("syrk_gram")
    flops_alg = float(m) * float(n) * (float(n) + 1.0)  # the whole problem's, all ranks together
    flops_tiles = s.syrkFlopsPerSolve()
    allreduce_bytes = s.stat("allreduce_bytes") if world > 1 else 0.0
    summ = sk.Solver.Summary()
    s.finish(summ)
    if world > 1:  # the slowest rank's J^T J time and iteration time
        t = torch.tensor([sec, dt], dtype=torch.float64, device="cuda")
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
        sec, dt = float(t[0].item()), float(t[1].item())
    rate = flops_alg * launches / sec * 1e-12 if sec > 0 else None
    rec = {"workload": "synthetic dense NLLS, %d residuals x %d parameters, DENSE_NORMAL_CHOLESKY (BASELINE.json configs[4])" % (m, n),
           "n_gpus": world, "iterations_per_second": 1.0 / dt, "seconds_per_iteration": dt,
           "jtj": {"bound": "mfma", "kernel": "sk::syrk_gram_f64_kernel", "achieved": rate, "peak": FP64_MFMA_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                   "frac": rate / (FP64_MFMA_PEAK_TFLOPS * world) if rate else None, "flops_per_launch": flops_alg / world,
                   "flops_per_launch_padded_tiles": flops_tiles, "launches": launches, "avg_launch_ms": 1e3 * sec / max(1, launches),
                   "note": "aggregate over the ranks: the whole problem's m n (n + 1) flops over the slowest rank's launch time; peak = %d x %.1f" % (world, FP64_MFMA_PEAK_TFLOPS)},
           "costs": [it["cost"] for it in summ.iterations()]}
    if world > 1:
        n_it = max(1, len(summ.iterations()) - 1)
        rec["distribution"] = {"mode": "rows sharded x%d, lower block triangle of J^T J all-reduced, Cholesky replicated" % world,
                               "allreduce_bytes_per_iteration": allreduce_bytes, "allreduce_ms_per_step": 1e3 * summ.phaseSeconds(5) / n_it}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ladybug-1723-156502")
    ap.add_argument("--cpu-iters", type=int, default=5, help="LM iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-full-iters", type=int, default=1, help="... and of the same baseline factoring every block (0 = skip)")
    ap.add_argument("--long-range", type=float, default=0.0, help="fraction of the tracks seen from two distant windows (loop closures)")
    ap.add_argument("--no-c5", action="store_true", help="skip the BASELINE.json config-5 (dense rows) record")
    ap.add_argument("--group", type=int, default=0, help="(tuning) SYRK depth in 128-column blocks")
    ap.add_argument("--no-lookahead", action="store_true", help="(tuning) single-stream Cholesky")
    ap.add_argument("--no-alone", action="store_true", help="skip the untimed side measurements (profiling runs)")
    ap.add_argument("--full-factorisation", action="store_true", help="factor every 128-block of the reduced system (no block envelope)")
    ap.add_argument("--no-resident-kernels", action="store_true", help="sk_options_set_resident_kernels(o, 0): the same plans launch by launch (counter-collection passes)")
    ap.add_argument("--dissection", default="auto", choices=["auto", "on", "off"], help="(tuning) two-way dissection of the camera sequence on ONE device")
    ap.add_argument("--retained", default="auto", help="(tuning) retained points: auto, off, or a count (sk_options_set_retained_points(o, ON, count))")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))

    import torch
    import torch.distributed as dist
    # developer knobs to rehearse the multi-rank path on a one-GPU box: all ranks on one device, exchange over gloo
    backend = os.environ.get("SK_BENCH_DIST_BACKEND", "nccl")
    if "SK_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SK_BENCH_DEVICE"])
    def fail(stage, e):
        """One JSON line the driver can parse instead of a traceback, and a non-zero exit code (a plain exit: never a re-exec)."""
        if rank == 0:
            print(json.dumps({"metric": "LM iterations/sec", "value": None, "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "error": "%s: %s: %s" % (stage, type(e).__name__, e)}), flush=True)
        sys.stderr.write("[bench] rank %d failed at %s: %r\n" % (rank, stage, e))
        sys.stderr.flush()
        os._exit(3)

    try:
        torch.cuda.set_device(local_rank)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            # the first collective is where a communicator really comes up (RCCL connects lazily): do it here, where a failure is reported as one
            probe = torch.ones(1, dtype=torch.float64, device="cuda")
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("the first all-reduce over %d ranks returned %r" % (world, probe.item()))
    except Exception as e:  # noqa: BLE001
        fail("process group (%s, %d ranks)" % (backend, world), e)

    import skeres_amd as sk
    from skeres_amd import bal
    sk.lib()  # fail loudly if the HIP library is missing

    prob = bal.generate_named(args.workload, seed=SEED, perturb=PERTURB, long_range_fraction=args.long_range)
    problem, params, loss = build_problem(sk, prob)
    options = sk.Solver.Options()
    options.setLinearSolverType(sk.LinearSolverType.DENSE_SCHUR)
    options.setMaxNumIterations(args.warmup + args.steps + 1000)
    options.setFunctionTolerance(0.0)
    options.setGradientTolerance(0.0)
    options.setParameterTolerance(0.0)
    options.setDevice(local_rank)
    options.setCholeskyTuning(args.group, not args.no_lookahead)
    options.setCholeskyEnvelope(not args.full_factorisation)
    options.setCholeskyDissection(args.dissection)
    if args.retained != "auto":
        options.setRetainedPoints("off") if args.retained == "off" else options.setRetainedPoints("on", int(args.retained))
    if args.no_resident_kernels:
        options.setResidentKernels(False)
    # a stream of our own, not torch's default (null) stream: the null stream synchronises implicitly with every
    # blocking stream, which would serialise the factorisation's CU-masked SYRK stream against it
    stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(stream)  # torch's current stream too: the all-reduce hook's collectives are ordered with the solver's kernels
    options.setStream(stream.cuda_stream)
    hook = None
    if world > 1:
        from skeres_amd import dist as sk_dist
        hook = sk_dist.attach(options, problem, rank, world)  # reduce buffer + all-reduce hook over torch.distributed (RCCL)

    t_create = time.time()
    try:
        solver = sk.StepSolver(options, problem)  # uploads the shard, builds the pair lists, runs iteration 0 (several ranks: the first calls of the all-reduce hook)
    except Exception as e:  # noqa: BLE001
        fail("solver set-up", e)
    t_create = time.time() - t_create  # (not part of `value`: the plan of the reduced system, the lists, the uploads, the device's one-time queue trial, iteration 0)
    dist_mode, t_allreduce, t_saved = solver.distribution() if world > 1 else ("single", 0.0, 0.0)
    allreduce_mb = solver.stat("allreduce_bytes") / 1e6 if world > 1 else 0.0
    for _ in range(args.warmup):
        solver.step()
    solver.setKernelTiming(2)  # HIP events around the trailing SYRK's launches only
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    phases0 = phase_snapshot(solver)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if solver.step():
            raise SystemExit("solver terminated inside the timed region")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    phases_timed = phases_per_step(phases0, phase_snapshot(solver), args.steps)
    syrk_s, syrk_n = solver.kernelSeconds("gemm_syrk")
    syrk_flops = solver.syrkFlopsPerSolve()
    syrk_c_bytes = solver.syrkCBytesPerSolve()
    plan = {k: solver.stat(k) for k in ("envelope_fill", "camera_order", "cholesky_flops_full", "cholesky_flops_plan", "cholesky_columns_resident", "segments",
                                        "allreduce_bytes", "dissected")}
    model_us = {}
    for k in range(1, 9):
        try:
            model_us[k] = solver.stat("model_us_segments_%d" % k)
        except sk.SkeresError:
            model_us[k] = 0.0
    # (what the model says of THIS run's factorisation: the lock-step dissection of one device where that was taken)
    model_one = model_us.get(1, 0.0)
    try:
        if solver.stat("dissected") == 1.0 and solver.stat("dissection_model_us") > 0.0:
            model_one = solver.stat("dissection_model_us")
    except sk.SkeresError:
        pass
    headline_stats = {}
    for k in ("model_us_segments_1", "border_cameras", "border_model_us", "dissected", "dissection_model_us", "retained_points", "retained_model_us",
              "retained_model_us_without", "dissection_head_cameras", "dissection_tail_cameras", "dissection_separator_cameras", "chain_steps"):
        try:
            headline_stats[k] = solver.stat(k)
        except sk.SkeresError:
            headline_stats[k] = 0.0
    try:
        solver_two_segments_us = solver.stat("model_us_two_segments_with_members")
    except sk.SkeresError:
        solver_two_segments_us = 0.0
    summary = sk.Solver.Summary()
    solver.finish(summary)
    # Untimed side measurement: the same kernel with the look-ahead off, i.e. alone on the chip.  In the
    # timed region above it shares the CUs with the panel chain of the next block-column group (and its
    # stream is masked off a few CUs per XCD), so its per-launch time there is longer by design.
    alone = None
    full_ms = None
    full_chol_s = None
    if world == 1 and not args.no_lookahead and not args.no_alone:
        del solver

        def side_run(lookahead, envelope, nsteps=3):
            prob2 = bal.generate_named(args.workload, seed=SEED, perturb=PERTURB, long_range_fraction=args.long_range)
            problem2, params2, loss2 = build_problem(sk, prob2)
            options.setCholeskyTuning(args.group, lookahead)
            options.setCholeskyEnvelope(envelope)
            solver2 = sk.StepSolver(options, problem2)
            solver2.step()
            solver2.setKernelTiming(2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nsteps):
                solver2.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / nsteps
            s2, n2 = solver2.kernelSeconds("gemm_syrk")
            rate = (solver2.syrkFlopsPerSolve() * nsteps) / s2 * 1e-12 if s2 > 0 else None
            flops2 = solver2.syrkFlopsPerSolve()
            summ2 = sk.Solver.Summary()
            solver2.finish(summ2)
            chol_s = summ2.phaseSeconds(2) / max(1, len(summ2.iterations()) - 1)  # Cholesky phase, seconds per iteration
            del solver2
            return rate, dt, flops2, chol_s
        alone, _, _, _ = side_run(False, not args.full_factorisation)
        # ... and the factorisation of EVERY block (no envelope): the same arithmetic (other SYRK grouping), more work in
        if not args.full_factorisation:
            _, full_dt, full_flops, full_chol_s = side_run(True, False, args.steps)  # over as many steps as the headline
            full_ms = 1e3 * full_dt
    its = summary.iterations()
    timed = its[1 + args.warmup: 1 + args.warmup + args.steps]
    n_success = int(sum(it["step_is_successful"] for it in timed))
    line = None
    traffic = pmc_traffic() if (args.workload == "ladybug-1723-156502" and args.group <= 0 and not args.full_factorisation) else None
    if rank == 0:
        achieved = (syrk_flops * args.steps) / syrk_s * 1e-12 if syrk_s > 0 else 0.0
        line = {
            "metric": "LM iterations/sec", "value": args.steps / elapsed, "unit": "LM iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BAL %s (synthetic, shape-exact: C=%d P=%d N=%d, seed %d), DENSE_SCHUR" % (
                args.workload, prob.num_cameras, prob.num_points, prob.num_observations, SEED),
                "linear_solver": "DENSE_SCHUR", "reduced_system_n": 9 * prob.num_cameras, "observations": prob.num_observations,
                "envelope_bytes": plan["allreduce_bytes"],  # the lower-triangular 128-blocks inside the envelope, in bytes (what the Schur assembly writes once)
                "cholesky": ("full: every 128-block of the reduced system" if args.full_factorisation else
                             "block envelope: cameras ordered for a banded reduced system, the structurally zero 128-blocks outside the envelope "
                             "skipped — the same arithmetic as the full factorisation (bit-identical at equal SYRK depth: "
                             "tests/test_gpu_parity.py::test_envelope_*); SYRK depth 2 and launch-by-launch look-ahead where the trailing SYRK "
                             "is the long pole, a resident panel chain (potrf server workgroup + per-column launches) where the serial chain is"),
                "ms_per_step_full_factorisation": full_ms,
                "envelope_fill": plan["envelope_fill"],  # fraction of the lower-triangular 128-blocks of the reduced system that is factored
                "camera_order": {0: "first appearance", 1: "memory order of the camera blocks", 2: "reverse Cuthill-McKee"}.get(int(plan["camera_order"])),
                "cholesky_block_columns_resident": int(plan["cholesky_columns_resident"]),
                "long_range_fraction": args.long_range,
                # the points the Schur complement does NOT eliminate (sk_options_set_retained_points, AUTO): the widest tracks stay in the
                # reduced system as border rows; "dissection": the lock-step two-way dissection of the camera band (head, tail, separator cameras)
                "retained_points": {"points": int(headline_stats.get("retained_points", 0)), "chain_model_us": headline_stats.get("retained_model_us", 0.0),
                                    "chain_model_us_all_eliminated": headline_stats.get("retained_model_us_without", 0.0)},
                "dissection": {"dissected": int(headline_stats.get("dissected", 0)), "head_cameras": int(headline_stats.get("dissection_head_cameras", 0)),
                               "tail_cameras": int(headline_stats.get("dissection_tail_cameras", 0)),
                               "separator_cameras_and_pseudo_cameras": int(headline_stats.get("dissection_separator_cameras", 0))},
                "solver_create_seconds": t_create,
                "successful_steps_in_timed_region": n_success, "parallelism": ("one GPU" if world == 1 else
                                "camera sequence cut into %d segments over the %d ranks (SK_DISTRIBUTION_SEGMENTED): rank r's device eliminates segment r "
                                "and its points (ranks beyond the segments replicate and add zeros); per iteration the separators' block-tridiagonal "
                                "system is all-reduced (%.1f MB) and factored by every rank" % (int(plan["segments"]), world, allreduce_mb) if dist_mode == "segmented" else
                                "points sharded x%d, reduced system all-reduced (%.0f MB: the blocks inside the envelope), Cholesky replicated" % (world, allreduce_mb) if dist_mode == "sharded" else
                                "replicated x%d: the solver measured %.1f ms for the all-reduce of the reduced system against %.1f ms of "
                                "per-iteration work sharding would remove, and did not shard" % (world, 1e3 * t_allreduce, 1e3 * t_saved)),
                "cost_first_timed": timed[0]["cost"] if timed else None, "cost_last_timed": timed[-1]["cost"] if timed else None},
            "roofline_syrk": {"bound": "mfma", "kernel": "sk::syrk_trailing_f64_kernel / sk::syrk_trailing_thin_f64_kernel (Cholesky trailing SYRK, one GEMM "
                                                     "body in 128x128 and 32x128 tiles, v_mfma_f64_16x16x4_f64)",
                         "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "plan": ("%s, %s" % ("camera sequence dissected (head + tail in one sequence of launches, root)" if plan["dissected"] else "undissected",
                                              "no resident kernels (launch by launch)" if args.no_resident_kernels else "resident panel chain")),
                         "launches_per_iteration": syrk_n // max(1, args.steps),
                         # each 128x128 tile of the trailing matrix read and written once per launch (K = 256 per launch where the SYRK is the long pole, 128 under the chain)
                         "algorithmic_c_tile_bytes_per_launch": syrk_c_bytes / max(1, syrk_n // max(1, args.steps)),
                         # HBM bytes by the counters: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH doubled, gfx950; Infinity-Cache hits
                         # included), run with --no-resident-kernels on the UNDISSECTED plan (counter collection serialises kernels) — and compared with the
                         # algorithmic bytes of THAT plan: traffic.ratio is like for like; the timed region above runs another plan
                         "traffic": (traffic or {}).get("bytes_per_launch"),
                         "traffic_like_for_like": traffic,
                         "launches": syrk_n, "avg_launch_ms": 1e3 * syrk_s / max(1, syrk_n),
                         "flops_per_solve": syrk_flops,
                         "achieved_alone": alone, "frac_alone": (alone / FP64_MFMA_PEAK_TFLOPS) if alone else None,
                         "note": "achieved/frac: live, inside the timed region, where the SYRK shares the chip with the look-ahead "
                                 "panel chain (masked off 16-32 CUs) and most launches are a few dozen tile rows (a sparse envelope); "
                                 "*_alone: same kernel, look-ahead off, 3 untimed steps; "
                                 "flops_per_solve counts the blocks inside the envelope only.  With retained points (config.retained_points) "
                                 "the factorisation has a fifth of the flops and every block column is bound by the serial panel chain: the "
                                 "SYRK launches that are left are ~60 per iteration of ~20 us (a few hundred 32 x 128 tiles, K = 128) and no "
                                 "longer where the time goes — by time the dominant kernel is the resident potrf server, which is latency-bound; "
                                 "the record all_points_eliminated holds the SYRK-bound plan of rounds 1-3 (its launches at 0.44 of the peak: "
                                 "profiles/r04_c_bench.json), roofline_full the dense case (0.57)"},
            "phases_ms_per_step": phases_timed,  # the timed region alone
        }
        hs = headline_stats
        line["config"]["chain_steps"] = int(hs.get("chain_steps", 0))
        line["config"]["plan"] = ("full factorisation" if args.full_factorisation else "block envelope") + (
            "; %d retained points" % int(hs.get("retained_points", 0)) if hs.get("retained_points", 0) > 0 else "; every point eliminated") + (
            "; lock-step dissection head %d | tail %d | separator %d cameras and pseudo-cameras" % (int(hs.get("dissection_head_cameras", 0)), int(hs.get("dissection_tail_cameras", 0)),
                                                                                                  int(hs.get("dissection_separator_cameras", 0))) if hs.get("dissected", 0) else "; undissected") + (
            "; launch by launch" if args.no_resident_kernels else "; resident panel chain")
        line["config"]["parallelism_short"] = "one GPU" if world == 1 else "%s x%d (%.1f MB all-reduced per iteration)" % (dist_mode, world, allreduce_mb)
        line["roofline_phases_hbm"] = phase_rooflines(line["phases_ms_per_step"], prob.num_observations if world == 1 else prob.num_observations / world,
                                                      plan["allreduce_bytes"] if world == 1 else 0.0)
        if world > 1:
            line["distribution"] = {"mode": dist_mode, "segments": int(plan["segments"]), "allreduce_bytes_per_iteration": plan["allreduce_bytes"],
                                    "allreduce_ms_per_step": line["phases_ms_per_step"]["allreduce"]}
        # What the chain model (microseconds per block column — 42 under a resident chain, 70 launch by
        # launch, the trailing update at 32 TFLOP/s where that is longer) predicts for this problem cut into 2, 4 and 8
        # segments, calibrated on this run: model(k) / model(1) x the measured Cholesky phase + the measured phases that
        # shard with the points / k (k segments).  UNMEASURED on more than one device until a SCALE run exists; printed so that one can be
        # read against it.
        if world == 1 and model_us.get(1, 0.0) > 0.0:
            ph = line["phases_ms_per_step"]
            shard = ph["jacobian_eval"] + ph["schur_assemble"] + ph["back_substitute"] + ph["cost_eval"]
            pred = {}
            for n_dev in (2, 4, 8):
                # the plan a world of n_dev ranks takes by itself (sk_problem_segment_plan, not forced: the rule of a real run — a
                # third or later segment has to beat the plan so far by 5 % in the model)
                k = int(problem.segmentPlan(n_dev, forced=False)[0])
                k = k if model_us.get(k, 0.0) > 0.0 else 1
                # (ranks beyond the segments are replicas of one: the point work shards by segment, not by rank)
                pred[str(n_dev)] = {"ms_per_step": ph["cholesky"] * model_us[k] / model_one + shard / k, "segments": k}
            note = "chain model, calibrated on this run's Cholesky phase; no scaling curve has been measured on hardware"
            if headline_stats.get("retained_points", 0) > 0:
                # with retained points a world of ranks takes the camera sequence as TWO segments — head and tail on two devices, the retained
                # points' pseudo-cameras members of the one separator (round 5) — when the chain model puts that 10 % under this device's
                # lock-step plan; ranks beyond two replicate.  Otherwise it shards the points or replicates.
                try:
                    two_us = solver_two_segments_us
                except NameError:
                    two_us = 0.0
                host_ms = line["ms_per_step"] - sum(ph.values())
                if two_us > 0.0 and two_us < 0.9 * model_one:
                    pred = {}
                    for n in (2, 4, 8):
                        # (more than two segments — the retained points' pseudo-cameras a border of the root and of every segment's front —
                        # where the model puts that 5 % under two: the solver's own rule)
                        k, us = 2, two_us
                        for kk in range(3, n + 1):
                            if model_us.get(kk, 0.0) > 0.0 and model_us[kk] < 0.95 * us:
                                k, us = kk, model_us[kk]
                        ms = ph["cholesky"] * us / model_one + shard / k + host_ms
                        pred[str(n)] = {"ms_per_step": ms, "speed_up": line["ms_per_step"] / ms, "segments": k,
                                        "plan": "%d segments, retained points in the %s%s" % (k, "separator" if k == 2 else "root's border", "" if n == k else "; %d replicas" % (n - k))}
                    note = ("chain model (a device per segment: %.0f us in two against %.0f us for this run's lock-step plan), calibrated on this run's Cholesky phase; the phases that "
                            "shard with the points divided by the segments; UNMEASURED on more than one device" % (two_us, model_one))
                else:
                    envelope_mb = plan["allreduce_bytes"] / 1e6
                    pred = {str(n): {"ms_per_step": line["ms_per_step"], "plan": "replicated"} for n in (2, 4, 8)}
                    note = ("with retained points a world of ranks would take two segments (%.0f us in the chain model against %.0f us on one device: no gain), shard the points or "
                            "replicate: sharding saves %.2f ms x (1 - 1 / N) of this iteration and adds an all-reduce of %.0f MB, so AUTO is expected to replicate; no scaling curve has "
                            "been measured on hardware" % (two_us, model_one, shard, envelope_mb))
            line["predicted_multi_gpu"] = {"model_us_per_segments": {str(k): v for k, v in model_us.items() if v > 0.0}, "model_us_this_run": model_one, "per_n_gpus": pred,
                                           "note": note}
        # The whole Cholesky phase (factorisation + triangular solves, every kernel of it) against the MFMA peak.  The
        # timed region factors the blocks inside the envelope (cholesky_flops_plan); the structure-independent figure
        # is the FULL factorisation — SURVEY.md section 8(d)'s n^3 / 3 over the measured Cholesky phase of the side run
        # that factors every block.
        chol_live_s = 1e-3 * phases_timed["cholesky"]
        if world == 1:
            line["chain_model"] = chain_model_record(lambda k: headline_stats.get(k, 0.0), 1e3 * chol_live_s)
        line["roofline_cholesky_phase"] = {
            "bound": "mfma", "flops": plan["cholesky_flops_plan"], "ms": 1e3 * chol_live_s,
            "achieved": plan["cholesky_flops_plan"] / chol_live_s * 1e-12 if chol_live_s > 0 else None, "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": plan["cholesky_flops_plan"] / chol_live_s * 1e-12 / FP64_MFMA_PEAK_TFLOPS if chol_live_s > 0 else None,
            "note": "all kernels of phase C (potrf, TRSM, updates, back-substitution) of the run above; flops of the blocks inside the envelope"}
        if full_chol_s:
            f_full = plan["cholesky_flops_full"]
            line["roofline_full"] = {
                "bound": "mfma", "flops": f_full, "cholesky_ms": 1e3 * full_chol_s, "ms_per_step": full_ms,
                "iterations_per_second": 1e3 / full_ms if full_ms else None,
                "achieved": f_full / full_chol_s * 1e-12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": f_full / full_chol_s * 1e-12 / FP64_MFMA_PEAK_TFLOPS,
                "note": "every 128-block factored (sk_options_set_cholesky_envelope(o, 0)): (9C)^3 / 3 flops over the measured Cholesky "
                        "phase, 3 untimed steps — what the iteration costs when the camera graph has no band to exploit"}
        if world > 1 and dist_mode == "replicated":
            # The solver measured that sharding this problem costs more than it saves, and every rank solved the WHOLE problem with no
            # collective in the timed region: `value` above counts those K iterations once (strong scaling, speed-up 1).  What the same
            # timed region also is: N complete, independent solves, one per GPU — the throughput a caller with N problems gets (weak
            # scaling without a collective).  Reported beside the headline, never in its place.
            line["independent_solves"] = {"value": world * args.steps / elapsed, "unit": "LM iterations/s over %d independent solves" % world, "scaling": "weak",
                                          "note": "the timed region of this run: each of the %d ranks completed its own solve of the whole problem, no collective" % world}
        if args.cpu_iters > 0 and world == 1:
            line["cpu_baseline"] = cpu_baseline(prob, args.cpu_iters, envelope=not args.full_factorisation, full_iters=args.cpu_full_iters)
        elif world == 1:
            line["cpu_baseline"] = None
    # The headline is complete here: rank 0 prints it NOW — the one stdout line of this run.  The other configurations below run on
    # every rank (with several ranks they use collectives) and report on stderr and in the side file; should one of them hang, a
    # watchdog ends the run after SK_BENCH_EXTRAS_TIMEOUT seconds with exit code 0 (a plain exit, never a re-exec) — whatever the
    # number of ranks.
    details_path = os.environ.get("SK_BENCH_DETAILS", os.path.join(ROOT, "gpurun_out", "bench_details.json"))
    records = {}

    def write_details():
        if rank != 0:
            return
        try:
            os.makedirs(os.path.dirname(details_path), exist_ok=True)
            with open(details_path + ".tmp", "w") as f:
                json.dump(records, f, indent=1)
            os.replace(details_path + ".tmp", details_path)
        except OSError as e:  # (a read-only tree: stderr still has every record)
            sys.stderr.write("[bench] details not written: %s\n" % e)

    def report(key, rec):
        """One record of the run: a compact line on stderr, and the side file rewritten (complete after every record)."""
        if rank != 0:
            return
        records[key] = rec
        sys.stderr.write(json.dumps({"record": key, **(rec if isinstance(rec, dict) else {"value": rec})}, separators=(",", ":")) + "\n")
        sys.stderr.flush()
        write_details()

    if rank == 0:
        line["details_file"] = os.path.relpath(details_path, ROOT) + " (and stderr: one JSON line per record)"
        report("headline", line)
        print(headline_line(line), flush=True)
    import threading
    timeout_s = float(os.environ.get("SK_BENCH_EXTRAS_TIMEOUT", "900"))

    def watchdog():
        report("extras_error", {"error": "the extra configurations did not finish within %s s; the headline is complete" % timeout_s})
        os._exit(0)
    timer = threading.Timer(timeout_s, watchdog)
    timer.daemon = True
    timer.start()
    # ---- the other configurations of BASELINE.json in the same run (every rank takes part; rank 0 reports).  None of them may
    # cost the headline its line: whatever goes wrong in one is recorded in its place. ----
    extra = {}
    if not args.no_alone and args.workload == "ladybug-1723-156502":
        try:
            del solver
        except NameError:
            pass
        dist_mod = dist if world > 1 else None

        def record(key, fn):
            try:
                extra[key] = fn()
            except Exception as e:  # noqa: BLE001 (e.g. not enough free HBM for the 80 GB Jacobian on a shared device)
                extra[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            report(key, extra[key])
        if world == 1:
            record("c2", lambda: bal_record(sk, bal, "problem-49-7776", 49, 20, 2, local_rank, stream, rank, world, dist_mod, torch))
        record("c4" if world == 1 else "venice",
               lambda: bal_record(sk, bal, "venice-1778-993923", 1778, 10, 2, local_rank, stream, rank, world, dist_mod, torch))
        if world == 1:
            # what the headline rests on: the same problem with 0.5 % of the tracks seen from two distant windows (loop closures:
            # the envelope of the reduced system fills up and the block-envelope factorisation has no zeros left to skip)
            record("loop_closures", lambda: bal_record(sk, bal, "ladybug-1723-156502", SEED, 6, 2, local_rank, stream, rank, world, dist_mod, torch,
                                                       long_range_fraction=0.005))
            # ... and with LOCALISED loop closures — three places seen twice, as a real sequence revisits streets: the revisiting
            # cameras are ordered into a trailing border of the reduced system (sk_options_set_cholesky_border, AUTO: the chain
            # model's choice), and beside it the same problem in the band's own order (border off)
            record("revisits", lambda: bal_record(sk, bal, "ladybug-1723-156502", SEED, 10, 2, local_rank, stream, rank, world, dist_mod, torch, revisits=REVISITS))
            # ... and the headline's problem with EVERY point eliminated (sk_options_set_retained_points(o, OFF): the plan of rounds 1-3)
            record("all_points_eliminated", lambda: bal_record(sk, bal, "ladybug-1723-156502", SEED, 10, 2, local_rank, stream, rank, world, dist_mod, torch, retained="off"))
            # (border off, retained points AUTO: a border of POINTS — the revisits' tracks among them — instead of cameras)
            record("revisits_retained_points_only", lambda: bal_record(sk, bal, "ladybug-1723-156502", SEED, 6, 2, local_rank, stream, rank, world, dist_mod, torch,
                                                                       revisits=REVISITS, border="off"))
            record("revisits_plain_order", lambda: bal_record(sk, bal, "ladybug-1723-156502", SEED, 6, 2, local_rank, stream, rank, world, dist_mod, torch,
                                                              revisits=REVISITS, border="off", retained="off"))
        if not args.no_c5:
            record("c5", lambda: c5_record(sk, rank=rank, world=world, stream=stream, torch=torch, dist_mod=dist_mod))
    timer.cancel()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
