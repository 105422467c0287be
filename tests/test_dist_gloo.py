"""N > 1 path on CPU: world_size 2 and 3 over gloo, emulator kernels, oracle as checker."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,order", [(2, "lat"), (3, "lat"), (4, "lat"), (2, "m")])
def test_wset_sharding_with_alltoallv(world, order):
    """order: row order inside the exchanged Fourier blocks -- latitude-major (default) or wavenumber-major
    (EMI_FB_ORDER=m, kept for A/B measurements)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    port = 29510 + world
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1024", EMI_TEST_NSMAX="9", EMI_FB_ORDER=order)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("DIST OK rank %d" % rank) in out, out
