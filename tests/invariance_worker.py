"""One task of the decomposition-invariance test (tests/test_decomposition_invariance.py, tests/test_gpu_shims.py).

The reference's harness dumps a CRC-64 of every gathered field after every transform
(src/programs/ectrans-benchmark.F90:1455-1600) and its test requires the dump of the serial run to be byte-identical
to the dump of every MPI/OpenMP decomposition (tests/compare_checksums.py:11-60).  This worker is that harness loop for
one task of an N-task W-set: dense call-mode-2 fields (the same global fields whatever N), 2 x (INV_TRANS, dump,
DIR_TRANS, dump); task 1 also saves the gathered arrays of the last iteration for a tolerance comparison."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ectrans_amd as et  # noqa: E402
from ectrans_amd import checksums  # noqa: E402
from tests.common import octahedral  # noqa: E402


def main():
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    out = os.environ["EMI_TEST_OUT"]
    on_gpu = os.environ.get("EMI_TEST_DEVICE", "cpu") == "cuda"
    N = int(os.environ.get("EMI_TEST_NSMAX", "10"))
    nlev, nvar = int(os.environ.get("EMI_TEST_NLEV", "2")), 2
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if on_gpu:
        import torch
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
        et.setup_trans0(kmax_resol=2, kprtrw=world, myproc=rank + 1, device=0)
    else:
        to = lambda a: np.ascontiguousarray(a)
        et._use_library_for_tests(os.path.join(ROOT, "tests", "emu", "libectrans_mi_emu.so"))
        et.setup_trans0(kmax_resol=2, kprtrw=world, myproc=rank + 1, device=None)
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    ns2g, ngg = et.trans_inq(r, "nspec2g"), et.trans_inq(r, "ngptotg")
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    # the same GLOBAL dense fields on every decomposition (global order: m = 0..N, n = m..N, re/im)
    rng = np.random.default_rng(20251114)
    n_of = np.concatenate([np.repeat(np.arange(m, N + 1), 2) for m in range(N + 1)])
    dense = lambda nf: rng.uniform(-0.5, 0.5, (ns2g, nf)) / (n_of[:, None] + 1.0)
    gvor, gdiv, gsc3, gsc2 = dense(nlev), dense(nlev), [dense(nlev) for _ in range(nvar)], dense(1)
    for a in (gvor, gdiv, gsc2, *gsc3):
        a[1:2 * (N + 1):2] = 0.0  # imaginary parts of m = 0
    gvor[0] = gdiv[0] = 0.0
    idx = et._global_spec_index(r)
    spvor, spdiv, spsc2 = to(gvor[idx]), to(gdiv[idx]), to(gsc2[idx])
    spsc3a = to(np.stack([g[idx] for g in gsc3]))
    npr = int(os.environ.get("EMI_TEST_NPROMA", "0")) or ng
    nb = (ng - 1) // npr + 1
    z = lambda *s: to(np.zeros(s))
    gpuv, gp3a, gp2 = z(nb, 2, nlev, npr), z(nb, nvar, nlev, npr), z(nb, 1, npr)
    fn = os.path.join(out, "benchmark_mpi%d" % world)  # + _inv_trans.checksums / _dir_trans.checksums, as the reference
    for it in (1, 2):
        et.inv_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2, kproma=npr)
        checksums.dump_checksums(fn + "_inv_trans.checksums", it, r, kproma=npr, zgpuv=gpuv, zgp3a=gp3a, zgp2=gp2)
        et.dir_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2, kproma=npr)
        checksums.dump_checksums(fn + "_dir_trans.checksums", it, r, zspvor=spvor, zspdiv=spdiv, zspsc3a=spsc3a, zspsc2=spsc2)
    # gathered arrays of the last iteration, for the tolerance comparison
    npy = et._np
    flat = lambda a: npy(a).reshape(npy(a).shape[0], -1, npy(a).shape[-1])
    ggrid = et.gath_grid(r, np.concatenate([flat(gpuv), flat(gp3a), flat(gp2)], axis=1), 2 * nlev + nvar * nlev + 1, kto=1)
    gspec = et.gath_spec(r, np.concatenate([npy(spvor), npy(spdiv), npy(spsc2)] + [npy(spsc3a)[v] for v in range(nvar)], axis=1),
                         2 * nlev + 1 + nvar * nlev, kto=1)
    if rank == 0:
        assert ggrid.shape == (2 * nlev + nvar * nlev + 1, ngg) and gspec.shape[0] == ns2g
        np.savez(os.path.join(out, "fields_mpi%d.npz" % world), grid=ggrid, spec=gspec)
    et.trans_release(r)
    et.trans_end()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    print("INVARIANCE OK rank %d" % rank, flush=True)


if __name__ == "__main__":
    main()
