"""1 task vs N tasks: the reference's checksum test (tests/compare_checksums.py:11-60, fixture wiring
tests/CMakeLists.txt:197-241) on the CPU tier -- emulator kernels, gloo exchange.  The GPU tier runs the same workers
with the HIP kernels (tests/test_gpu_shims.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_decompositions(outdir, worlds, device, nsmax, port0, extra_env=None, nthreads="1024"):
    """runs tests/invariance_worker.py for every task count; returns {world: (checksum file, fields file)}"""
    res = {}
    for world in worlds:
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port0 + world),
                       EMI_TEST_OUT=str(outdir), EMI_TEST_DEVICE=device, EMI_TEST_NSMAX=str(nsmax), OMP_NUM_THREADS=nthreads)
            env.update(extra_env or {})
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "invariance_worker.py")], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        for rank, p in enumerate(procs):
            try:
                out, _ = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            assert p.returncode == 0 and ("INVARIANCE OK rank %d" % rank) in out, out
        res[world] = (os.path.join(str(outdir), "benchmark_mpi%d" % world), os.path.join(str(outdir), "fields_mpi%d.npz" % world))
    return res


def check_invariance(res, tol=1e-13):
    """every decomposition against the one-task run: gathered fields to `tol` of the field maximum, and the checksum
    dumps byte for byte (the reference's criterion)"""
    from ectrans_amd.checksums import compare_checksums
    ref_ck, ref_f = res[1]
    ref = np.load(ref_f)
    kinds = ("_inv_trans.checksums", "_dir_trans.checksums")
    text = [open(ref_ck + k).read() for k in kinds]
    assert all(t.count("iteration") == 2 for t in text) and "zgpuv (1, 1) = " in text[0] and "zgp2 (1) = " in text[0]
    assert "zspsc3a (2, 2) = " in text[1] and "zspsc2 (1) = " in text[1]
    for world, (ck, f) in res.items():
        if world == 1:
            continue
        got = np.load(f)
        for k in ("grid", "spec"):
            err = np.abs(got[k] - ref[k]).max() / np.abs(ref[k]).max()
            assert err < tol, (world, k, err)
        for k in kinds:
            assert compare_checksums(ref_ck + k, ck + k), "checksum dumps %s of 1 and %d tasks differ" % (k, world)


def test_one_task_and_n_tasks_give_the_same_fields_and_checksums(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    res = run_decompositions(tmp_path, (1, 2, 4), "cpu", 9, 29560)
    check_invariance(res)
