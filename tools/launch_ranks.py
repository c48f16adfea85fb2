#!/usr/bin/env python3
"""Start N ranks of a worker script with the torch.distributed environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT) WITHOUT importing torch: `python -m torch.distributed.run` itself opens the GPU, and the GPU boxes allow six
processes on a card — with this launcher a rehearsal of six ranks sharing one GPU fits (tools/rehearse_worlds.sh).

  python tools/launch_ranks.py N script.py [args...]

Exit code: the first non-zero exit code of a rank (the others are terminated), else 0."""
import os
import subprocess
import sys
import time


def main():
    n = int(sys.argv[1])
    cmd = sys.argv[2:]
    port = str(29400 + os.getpid() % 500)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable] + cmd, env=env))
    rc = 0
    live = set(range(n))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in live:
                    procs[q].terminate()
        time.sleep(0.2)
    sys.exit(rc)


if __name__ == "__main__":
    main()
