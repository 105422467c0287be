#!/usr/bin/env python3
"""bench.py -- dir+inv transform-pairs/s of the MI355X-native spectral transform.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by
torch.distributed.run with one rank per GPU.  A *step* is one INV_TRANS + DIR_TRANS pair
(exactly the timed loop of the reference harness, src/programs/ectrans-benchmark.F90:619-769)
over device-resident synthetic fields: nlev levels of (vor,div), nfld x nlev 3-D scalars and one
2-D scalar, every coefficient zero except Re(m=4,n=19)=1 (ectrans-benchmark.F90:1381-1419).

Default workload = BASELINE.json's metric config: TCo1279 (O1280), 137 levels x 10 fields
(KF = 2*137 + 10*137 + 1 = 1645 Fourier-space fields), fp64, inputs resident in HBM.

N > 1 (one rank per GPU, launched by torch.distributed.run): the SAME global workload is sharded the
way the reference shards it over its W-sets -- zonal wavenumbers zig-zag over ranks, latitudes in
contiguous bands -- with one RCCL all-to-all-v of the device-resident Fourier buffer per direction
(DESIGN.md section 7).  Total work is fixed, so "scaling" is "strong" and `value` is the pairs/s
of the whole job.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# fp64 matrix-core peak of MI355X: 256 CU x 4 SIMD x 32 FLOP/clk/SIMD (v_mfma_f64_16x16x4_f64 =
# 2048 FLOP / 64 cycles) x 2.4 GHz = 78.6 TFLOP/s (AMD MI355X datasheet "FP64 matrix 78.6 TF";
# /opt/skills/guides/MI355X_MICROARCH.md lists no fp64 row; tools/mfma_f64_peak.hip measures 78.0 on the box,
# profiles/r1e_mfma_peak.txt).
PEAK_F64_MFMA_TFLOPS = 78.6
# fp32 matrix-core peak (v_mfma_f32_16x16x4_f32: 2048 FLOP / 32 cycles): 157.3 TFLOP/s
# (/opt/skills/guides/MI355X_MICROARCH.md, "FP32 matrix")
PEAK_F32_MFMA_TFLOPS = 157.3


def octahedral(nsmax):
    h = nsmax + 1
    return np.array([20 + 4 * i for i in range(h)] + [20 + 4 * i for i in reversed(range(h))], dtype=np.int32)


def cpu_baseline(nsmax, kf_full, budget_s=20.0):
    """Oracle (C restatement of the reference CPU path, OpenMP) on this host's cores, on a
    bounded sample of the same workload: same grid/truncation, fewer fields; pairs/s scaled
    linearly in the field count (flops and bytes are exactly linear in KF)."""
    from oracle.oracle import Oracle
    t0 = time.time()
    o = Oracle(nsmax, octahedral(nsmax))
    t_setup = time.time() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))

    def pair(nf):
        sp = np.zeros((o.nspec2, nf))
        sp[o.nasm0[4] - 1 + 2 * (19 - 4)] = 1.0
        t = time.time()
        g = o.inv_trans(spvor=sp, spdiv=sp, spsc=sp)
        o.dir_trans(g, nuv=nf, nsc=nf)
        return time.time() - t

    t1 = pair(1)  # KF = 3 (u, v, one scalar)
    nf = int(max(1, min(64, budget_s / max(t1, 1e-3))))
    t = pair(nf) if nf > 1 else t1
    kf = 3 * nf
    return {"value": (1.0 / t) * kf / kf_full, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "same grid+truncation, %d of %d Fourier fields (vor/div/scalar x %d), scaled linearly in KF; "
                      "oracle setup %.1fs not included" % (kf, kf_full, nf, t_setup)}


def recorded_traffic(N, nlev, nfld, esz, world):
    """HBM bytes per Legendre launch from the committed PMC passes (profiles/*_pmc_traffic.json, collected
    with tools/collect_profiles.sh on this exact workload; rocprofv3 cannot run inside the timed region).
    Only reported for the workload the counters were taken on."""
    if (N, nlev, nfld, esz, world) != (1279, 137, 10, 8, 1):
        return None, None
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1]))["kernels"]
        vals = [k[n]["hbm_bytes_per_launch"] for n in ("emi_f64::k_leg_inv", "emi_f64::k_leg_dir")]
        return sum(vals) / len(vals), os.path.basename(files[-1])
    except (KeyError, ValueError, OSError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nsmax", type=int, default=1279)
    ap.add_argument("--nlev", type=int, default=137)
    ap.add_argument("--nfld", type=int, default=10)
    ap.add_argument("--precision", type=int, default=8, choices=(4, 8),
                    help="8: fp64 library (headline metric); 4: fp32 library (BASELINE configs[4]'s arithmetic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-batch", type=int, default=0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: start the N ranks as a child job (one process per
        # GPU, as the driver does with torch.distributed.run) and pass its exit code on
        import subprocess
        port = 29500 + os.getpid() % 1000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("EMI_BENCH_ONE_GPU"):
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # EMI_BENCH_BACKEND=gloo + EMI_BENCH_ONE_GPU=1: test configuration (several ranks sharing one GPU,
    # exchange staged through the host); the driver's runs use RCCL, one GPU per rank.
    backend = os.environ.get("EMI_BENCH_BACKEND", "nccl")
    rdev = dev if backend == "nccl" else torch.device("cpu")  # where small reductions live
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import ectrans_amd as et
    N, nlev, nfld = args.nsmax, args.nlev, args.nfld
    kf = 2 * nlev + nfld * nlev + 1
    et.setup_trans0(kmax_resol=2, device=local, kprtrw=world, myproc=rank + 1)
    if args.max_batch:
        et.set_max_batch(args.max_batch)
    t0 = time.time()
    r = et.setup_trans(N, 2 * (N + 1), octahedral(N), precision=args.precision)
    esz = args.precision
    peak = PEAK_F64_MFMA_TFLOPS if esz == 8 else PEAK_F32_MFMA_TFLOPS
    t_setup = time.time() - t0
    nspec2, ngptot = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")  # this rank's share
    a4 = int(et.trans_inq(r, "nasm0")[4])
    i419 = a4 - 1 + 2 * (19 - 4) if a4 > 0 else None  # only the rank owning m=4 holds the harmonic

    def z(*shape):
        return torch.zeros(shape, dtype=torch.float64 if esz == 8 else torch.float32, device=dev)

    # call mode 2 arrays of the reference harness (ectrans-benchmark.F90:450-479)
    spvor, spdiv, spsc3a, spsc2 = z(nspec2, nlev), z(nspec2, nlev), z(nfld, nspec2, nlev), z(nspec2, 1)
    if i419 is not None:
        for a in (spvor, spdiv, spsc2):
            a[i419] = 1.0
        spsc3a[:, i419] = 1.0
    gpuv, gp3a, gp2 = z(1, 2, nlev, ngptot), z(1, nfld, nlev, ngptot), z(1, 1, ngptot)
    n0 = et.specnorm(r, spsc2)[0]

    def step():
        et.inv_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
        et.dir_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # ---- timed region: exactly K steps; HIP-event phase timers run inside and are only resolved after the
    # region (accumulating mode): no host synchronisation between the calls
    et.set_profile(2)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        et.inv_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
        et.dir_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2)
    barrier()
    dt = time.perf_counter() - t0
    pack_ms, leg_ms, fft_ms = et.last_phase_ms()
    leg_launches = et.last_phase_launches()[1]
    wm = et.work_model(r, kf)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # whole-job Legendre rate: flops of all ranks / slowest rank's kernel time
        red = torch.tensor([wm["legendre_flops"], wm["fourier_bytes"], float(ngptot)], dtype=torch.float64, device=rdev)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        mx = torch.tensor([leg_ms, fft_ms, pack_ms], dtype=torch.float64, device=rdev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        wm["legendre_flops"], wm["fourier_bytes"], ngptot = float(red[0]), float(red[1]), int(red[2].item())
        leg_ms, fft_ms, pack_ms = (float(x) for x in mx)
    n1 = et.specnorm(r, spsc2)[0]

    if rank == 0:
        # per launch: algorithmic flops of the launch (all ranks) / average launch duration (slowest rank)
        flops_per_launch = wm["legendre_flops"] * 2 * args.steps / max(leg_launches, 1)
        ms_per_launch = leg_ms / max(leg_launches, 1)
        ach = flops_per_launch / (ms_per_launch * 1e-3) / 1e12 if ms_per_launch > 0 else 0.0
        traffic, traffic_src = recorded_traffic(N, nlev, nfld, esz, world)
        out = {
            "metric": "dir+inv transform-pairs/sec, TCo%d %dL x %d fields; spectral-norm rel-error" % (N, nlev, nfld),
            "value": args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64" if esz == 8 else "f32", "data": "synthetic",
            "config": {"workload": "TCo%d/O%d, %d levels x %d 3-D fields + vor/div + 1 surface field, KF=%d, "
                                   "device-resident call-mode-2 arrays" % (N, N + 1, nlev, nfld, kf),
                       "parallelism": "1 GPU" if world == 1 else
                       "%d GPUs: zonal wavenumbers zig-zag + latitude bands, RCCL all-to-all-v per direction" % world,
                       "setup_s": round(t_setup, 2)},
            "spectral_norm_rel_error": abs(n0 / n1 - 1.0),
            "roofline": {"bound": "mfma", "kernel": "k_leg_inv + k_leg_dir (fp%d MFMA Legendre transforms)" % (8 * esz),
                         "achieved": ach, "peak": peak * world, "unit": "TFLOP/s",
                         "frac": ach / (peak * world), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": 90.5e9 if traffic else None,
                         "launches": leg_launches, "avg_launch_ms": ms_per_launch,
                         "algorithmic_flops_per_launch": flops_per_launch},
            "phase_ms_per_step": {"spectral_pack_unpack": pack_ms / args.steps, "legendre_mfma": leg_ms / args.steps,
                                  "fft": fft_ms / args.steps},
            "fft_hbm": {"algorithmic_GB_per_step": 2 * (wm["fourier_bytes"] + kf * ngptot * float(esz)) / 1e9,
                        "achieved_GBps": 2 * (wm["fourier_bytes"] + kf * ngptot * float(esz)) / 1e9 / max(fft_ms / args.steps * 1e-3, 1e-9),
                        "peak_GBps": 8000.0 * world},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, kf)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
