python -m pytest tests -x -q -m gpu --durations=12 -k "bal or dense_schur or schedule or traced or host or envelope or default_plan or lm_step or properties or border or sharded or world" > gpurun_out/r4d_tests.log 2>&1; tail -22 gpurun_out/r4d_tests.log; python tools/phase_times.py ladybug 10 2>&1 | tail -n 1; python tools/phase_times.py venice 6 2>&1 | tail -n 1; python bench.py --steps 20 --warmup 5 --no-c5 --cpu-iters 0 > gpurun_out/r4d_bench.json 2>gpurun_out/r4d_bench.err; python - <<EOP
import json
d=json.loads(open("gpurun_out/r4d_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["phases_ms_per_step"])
for k in ("c2","c4","loop_closures","revisits"):
    r=d.get(k,{})
    print(k, r.get("error") or (r["iterations_per_second"], r["ms_per_step"], r["phases_ms_per_step"]))
EOP
