"""CPU, world_size 2 (gloo): the N>1 path's host logic and algorithm -- brick
partition, interface plans, canonical-order halo sums, owner-masked reductions,
owner-held penalties -- reproduce the serial solution."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 4])
def test_multidomain_algorithm_matches_serial(world, oracle):
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    port = 29500 + world + (os.getpid() % 500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "_dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.count("ok=True") == world
