"""world_size-2 (and 3) gloo runs of the multi-GPU decomposition on the CPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_reduced_system_sums_to_the_whole(built, world):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    port = 29500 + world + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DIST_OK world=%d" % world in out.stdout


@pytest.mark.parametrize("world", [2, 4, 8])
def test_segmented_plan_and_substructuring_on_the_cpu(built, world):
    """The SEGMENTED distribution over 2, 4 and 8 ranks: the plan from the product's host
    logic (sk_problem_segment_plan), the substructuring arithmetic it implies in numpy on the oracle's reduced systems, one
    gloo all-reduce of the separators' system — against the solution of the whole reduced system (tests/dist_segments_worker.py)."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    port = 29300 + world + (os.getpid() % 150)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_segments_worker.py"), str(world)]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DIST_SEGMENTS_OK world=%d segments=%d" % (world, world) in out.stdout
