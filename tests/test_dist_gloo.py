"""N > 1 path on CPU: world_size 2 and 3 over gloo, emulator kernels, oracle as checker."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])  # 4 tasks and more: GPU tier (tests/test_gpu_shims.py)
def test_wset_sharding_with_alltoallv(world):
    """W-set sharding (zonal wavenumbers zig-zag, latitude bands) with the all-to-all-v of whole Fourier blocks over gloo."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    port = 29510 + world
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1024", EMI_TEST_NSMAX="9",
                   EMI_GATH_CHUNK="1000" if world == 3 else "")  # world 3: GATH_* in chunks of one or two fields (the bounded-memory path)
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


def run_vsets(world, nprtrv, port, device="cpu", extra=None, timeout=1200):
    """tests/vsets_worker.py on `world` = NPRTRW x `nprtrv` tasks; every task checks its pieces against the oracle"""
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1024", EMI_TEST_NPRTRV=str(nprtrv), EMI_TEST_DEVICE=device)
        env.update(extra or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "vsets_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("VSETS OK rank %d" % rank) in out, out


@pytest.mark.parametrize("world,nprtrv", [(2, 2), (4, 2)])  # 2 x 3: GPU tier
def test_vset_sharding(world, nprtrv):
    """NPRTRV > 1 (sump_trans0_mod.F90:49, inv_trans.F90:212-300): NPRTRW x NPRTRV tasks -- 1 x 2 and 2 x 2 (2 x 3, 4 x 2: GPU tier).
    Spectral arrays hold the fields of the task's V-set (KVSETUV / KVSETSC / KVSETSC2 / KVSETSC3A), grid arrays ALL fields on
    the task's latitudes; TRLTOG / TRGTOL between the V-sets of a band is a second all-to-all-v.  Emulator kernels, gloo."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    run_vsets(world, nprtrv, 29530 + world + nprtrv)


def test_vset_that_owns_no_level():
    """NPRTRV = 2 with ONE level: the task of V-set 2 passes no PSPSC3A, so it cannot name the variable count of PGP3A from its
    spectral array -- the reference takes it from UBOUND(PSPSC3A,3) of a zero-level array (inv_trans.F90:272-277); here it also
    travels from the grid array (emi_vsets_t.nvar3a_g), or the tasks would list different global fields and the TRLTOG / TRGTOL
    counts would disagree."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    run_vsets(2, 2, 29549, extra={"EMI_TEST_NLEV": "1"})
