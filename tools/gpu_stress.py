"""One-off stress run on the GPU box (not part of the test suite): a wider version of
tests/test_gpu_parity.py::test_random_reduced_grids_match_oracle -- more latitudes, longer rows (through the
global-scratch FFT kernels too), more fields (several column tiles, field batches), both precisions, random options.
    python tools/gpu_stress.py [ncases] [first_seed]        prints one line per failure and a summary."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before the library: one HIP runtime)

torch.cuda.init()
import ectrans_amd as et  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from tests.common import adjoint_case, run_case  # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    et.setup_trans0(kmax_resol=4, device=0)
    dev = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0"), lambda t: t.cpu().numpy())
    bad, worst, worst_adj = 0, {8: 0.0, 4: 0.0}, {8: 0.0, 4: 0.0}
    t0 = time.time()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    cur = open(os.path.join(ROOT, "gpurun_out", "stress_current_seed.txt"), "w")
    for i in range(ncases):
        seed = seed0 + i
        rng = np.random.default_rng(seed)
        nh = int(rng.integers(4, 33))
        top = int(rng.choice([300, 1500, 6000, 12000], p=[0.4, 0.3, 0.25, 0.05]))
        half = np.sort(rng.integers(8, top, nh))
        shape = rng.random()
        if shape < 0.08:  # full grid: every latitude the same (possibly odd) number of points
            half[:] = int(rng.integers(8, top))
        elif shape < 0.16:  # one to three latitudes per hemisphere
            nh = int(rng.integers(1, 4))
            half = half[-nh:]
        nloen = np.concatenate([half, half[::-1]]).astype(np.int32)
        nsmax = int(rng.integers(2, max(2 * nh, 4)))
        if rng.random() < 0.15:  # long Legendre side (k_leg_dir's one-parity tiles start at 65 (n - m) pairs), often finer than the grid
            nsmax = int(rng.integers(65, 400))
        big = rng.random() < 0.25
        nuv = int(rng.integers(0, 40 if big else 3))
        nsc = int(rng.integers(0, 90 if big else 4))
        if nuv + nsc == 0:
            nsc = 1
        flags = dict(scders=bool(rng.integers(2)) and nsc > 0, uvder=bool(rng.integers(2)) and nuv > 0,
                     vorgp=bool(rng.integers(2)) and nuv > 0, divgp=bool(rng.integers(2)) and nuv > 0)
        nproma = [None, 17, 100, 1000, 4096][int(rng.integers(5))]
        prec = 8 if rng.random() < 0.7 else 4
        mb = int(rng.choice([0, 64, 128]))
        tol = 1e-11 if prec == 8 else 3e-5
        adj = rng.random() < 0.1
        cur.seek(0)
        cur.write("%d\n" % seed)  # a memory fault kills the process: the seed of the case in flight survives in this file
        cur.flush()
        try:
            et.set_max_batch(mb)
            if adj:  # INV_TRANSAD / DIR_TRANSAD against INV_TRANS / DIR_TRANS (dot-product identity, the reference's tolerance of 2000 epsilons)
                e_inv, e_dir = adjoint_case(et, dev, nsmax, nloen, nuv, nsc, nproma, seed=seed, precision=prec)
                tol = 2000 * float(np.finfo(np.float32 if prec == 4 else np.float64).eps)
                worst_adj[prec] = max(worst_adj[prec], e_inv, e_dir)
            else:
                e_inv, e_dir = run_case(et, Oracle, dev, nsmax, nloen, nuv, nsc, flags, nproma, seed=seed, precision=prec)
                worst[prec] = max(worst[prec], e_inv, e_dir)
            ok = e_inv < tol and e_dir < tol
            msg = "e_inv %.2e e_dir %.2e" % (e_inv, e_dir)
        except Exception as exc:  # noqa: BLE001
            ok, msg = False, "%s: %s" % (type(exc).__name__, exc)
        if not ok:
            bad += 1
            print("FAIL seed", seed, dict(nloen=nloen.tolist(), nsmax=nsmax, nuv=nuv, nsc=nsc, flags=flags, nproma=nproma,
                                          precision=prec, max_batch=mb, adjoint=adj), msg, flush=True)
    et.set_max_batch(0)
    print("stress: %d cases, %d failures, worst fp64 %.2e, worst fp32 %.2e (adjoint identities: %.2e, %.2e), %.0f s"
          % (ncases, bad, worst[8], worst[4], worst_adj[8], worst_adj[4], time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
