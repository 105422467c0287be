"""One task of the sharded TCo1279 parity test (tests/test_gpu_fullsize.py::test_tco1279_sharded_over_8_tasks_on_one_gpu).

BASELINE configs[3] is TCo1279, 137 levels x 10 fields on 8 GPUs.  A one-GPU box cannot time it, but it can RUN it: the 8
tasks of the W-set share cuda:0 (288 GB), the all-to-all-v of TRMTOL / TRLTOM is staged through gloo (RCCL refuses two
ranks on one device), and everything else is the production path -- per-task tile maps, `fftrow` exchange-order tables,
the 4-batch 3-stream pipeline (EMI_PIPELINE_DIST) -- at the real resolution with >= 274 Fourier fields.  As the
reference does (ectrans-benchmark.F90:743-756, 847-871: norms are checked at the decomposition that is timed), every task
checks its share of every field against the oracle: the call-mode-2 arrays hold c_f x (one of 3 scalar base fields | the
vor/div base pair), the parent test transformed the 5 base fields once with the lazy-panel oracle.  Task 1 also gathers
three fields (GATH_GRID / GATH_SPEC) and stores them with their CRC-64 for the comparison with the one-task run
(/root/reference/tests/compare_checksums.py:11-60)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ectrans_amd as et  # noqa: E402
from ectrans_amd.checksums import crc64  # noqa: E402
from tests.common import octahedral  # noqa: E402


def main():
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    out = os.environ["EMI_TEST_OUT"]
    N, nlev, nfld = int(os.environ["EMI_TEST_NSMAX"]), int(os.environ["EMI_TEST_NLEV"]), int(os.environ["EMI_TEST_NFLD"])
    dev = torch.device("cuda:0")
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    et.setup_trans0(kmax_resol=2, kprtrw=world, myproc=rank + 1, device=0)
    nloen = octahedral(N)
    r = et.setup_trans(N, len(nloen), nloen)
    base = np.load(os.path.join(out, "base.npz"))  # V, D (ns2g), S (ns2g, 3); gref (5, ngg): u, v, s0, s1, s2; vr, dr (ns2g), sr (ns2g, 3)
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    idx = et._global_spec_index(r)
    lat0, lat1 = et.trans_inq(r, "nfrstlat") - 1, et.trans_inq(r, "nlstlat")
    gp0 = int(nloen[:lat0].sum())
    assert ng == int(nloen[lat0:lat1].sum()) and ns2 == idx.size
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    nb = 3
    cuv = torch.tensor([(1.0 + 0.37 * l / nlev) * (-1) ** l for l in range(nlev)], dtype=torch.float64, device=dev)
    c3 = torch.tensor([[(0.5 + (v * nlev + l + 1) / (nfld * nlev)) * (-1) ** (v + l) for l in range(nlev)] for v in range(nfld)],
                      dtype=torch.float64, device=dev)
    c2 = -1.75
    base3 = torch.tensor([[(v * nlev + l) % nb for l in range(nlev)] for v in range(nfld)], device=dev)
    tV, tD, tS = t(base["V"][idx]), t(base["D"][idx]), t(base["S"][idx].T)  # local spectral shares; tS (3, ns2)
    spvor, spdiv = tV[:, None] * cuv[None, :], tD[:, None] * cuv[None, :]
    spsc3a = torch.empty((nfld, ns2, nlev), dtype=torch.float64, device=dev)
    for v in range(nfld):
        spsc3a[v] = tS[base3[v]].T * c3[v][None, :]
    spsc2 = (c2 * tS[0])[:, None].contiguous()
    z = lambda *s: torch.zeros(s, dtype=torch.float64, device=dev)
    gpuv, gp3a, gp2 = z(1, 2, nlev, ng), z(1, nfld, nlev, ng), z(1, 1, ng)
    tg = t(base["gref"][:, gp0:gp0 + ng])  # this task's latitude band of the oracle's grid fields

    def worst(got, coef, refs, ix):
        """max over the fields of max |got_f - c_f ref_ix(f)| / max |c_f ref_ix(f)|  (ref maxima over the task's share)"""
        rmax = refs.abs().amax(dim=1)
        w = 0.0
        for i in range(0, got.shape[0], 16):
            c, j = coef[i:i + 16], ix[i:i + 16]
            d = refs[j]
            d.mul_(c[:, None]).sub_(got[i:i + 16]).abs_()
            w = max(w, float((d.amax(dim=1) / (c.abs() * rmax[j])).max()))
        return w

    zero = lambda n: torch.zeros(n, dtype=torch.long, device=dev)
    one = lambda x: torch.tensor([x], dtype=torch.float64, device=dev)
    et.inv_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    torch.cuda.synchronize()
    e_inv = max(worst(gpuv[0, 0], cuv, tg[0:1], zero(nlev)), worst(gpuv[0, 1], cuv, tg[1:2], zero(nlev)), worst(gp2[0], one(c2), tg[2:3], zero(1)))
    for v in range(nfld):
        e_inv = max(e_inv, worst(gp3a[0, v], c3[v], tg[2:], base3[v]))
    # three gathered grid fields for the comparison with the one-task run: u of level 1, the last 3-D scalar field, the surface field
    sel_g = np.stack([gpuv[0, 0, 0].cpu().numpy(), gp3a[0, nfld - 1, nlev - 1].cpu().numpy(), gp2[0, 0].cpu().numpy()])[None]  # (1, 3, ng)
    ggrid = et.gath_grid(r, sel_g, 3, kto=1)
    # ---- direct transform of the exact images of the oracle's grid fields
    for l0 in range(0, nlev, 16):
        sl = slice(l0, min(nlev, l0 + 16))
        gpuv[0, 0, sl] = cuv[sl, None] * tg[0][None, :]
        gpuv[0, 1, sl] = cuv[sl, None] * tg[1][None, :]
        for v in range(nfld):
            gp3a[0, v, sl] = c3[v, sl, None] * tg[2 + base3[v, sl]]
    gp2[0, 0] = c2 * tg[2]
    for a in (spvor, spdiv, spsc3a, spsc2):
        a.zero_()
    et.dir_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    torch.cuda.synchronize()
    tvr, tdr, tsr = t(base["vr"][idx]), t(base["dr"][idx]), t(base["sr"][idx].T)
    # reference maxima over the GLOBAL field (a task's wavenumbers may all be small)
    gmax = lambda a: float(np.abs(a).max())

    def worst_sp(got, coef, refs, ix, rmaxg):
        w = 0.0
        for i in range(0, got.shape[0], 16):
            c, j = coef[i:i + 16], ix[i:i + 16]
            d = refs[j]
            d.mul_(c[:, None]).sub_(got[i:i + 16]).abs_()
            w = max(w, float((d.amax(dim=1) / (c.abs() * rmaxg[j])).max()))
        return w

    mv, md = torch.tensor([gmax(base["vr"])], device=dev), torch.tensor([gmax(base["dr"])], device=dev)
    ms = torch.tensor([gmax(base["sr"][:, k]) for k in range(nb)], device=dev)
    e_dir = max(worst_sp(spvor.T, cuv, tvr[None], zero(nlev), mv), worst_sp(spdiv.T, cuv, tdr[None], zero(nlev), md),
                worst_sp(spsc2.T, one(c2), tsr[0:1], zero(1), ms[0:1]))
    for v in range(nfld):
        e_dir = max(e_dir, worst_sp(spsc3a[v].T, c3[v], tsr, base3[v], ms))
    sel_s = np.stack([spvor[:, 0].cpu().numpy(), spsc3a[nfld - 1, :, nlev - 1].cpu().numpy(), spsc2[:, 0].cpu().numpy()], axis=1)  # (ns2, 3)
    gspec = et.gath_spec(r, sel_s, 3, kto=1)
    print("rank %d/%d: nspec2 %d ngptot %d e_inv %.2e e_dir %.2e" % (rank, world, ns2, ng, e_inv, e_dir), flush=True)
    assert e_inv < 1e-11 and e_dir < 1e-11, (e_inv, e_dir)
    if rank == 0:
        crc = [crc64(np.ascontiguousarray(a)) for a in (ggrid[0], ggrid[1], ggrid[2], gspec[:, 0].copy(), gspec[:, 1].copy(), gspec[:, 2].copy())]
        np.savez(os.path.join(out, "gathered_mpi%d.npz" % world), grid=ggrid, spec=gspec, crc=np.array(crc, dtype=np.uint64))
    et.trans_release(r)
    et.trans_end()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    print("FULLSIZE DIST OK rank %d" % rank, flush=True)


if __name__ == "__main__":
    main()
