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


def random_spectrum(rng, nasm0, nsmax, nspec2, nf, zero00):
    """Dense case of SURVEY 8d: U(-0.5, 0.5) / (n + 1), imag(m = 0) = 0, (0, 0) = 0 for vor/div."""
    n_of = np.zeros(nspec2)
    for m in range(nsmax + 1):
        i0 = nasm0[m] - 1
        n_of[i0:i0 + 2 * (nsmax - m + 1)] = np.repeat(np.arange(m, nsmax + 1), 2)
    sp = rng.uniform(-0.5, 0.5, (nspec2, nf)) / (n_of[:, None] + 1.0)
    sp[1:2 * (nsmax + 1):2] = 0.0
    if zero00:
        sp[0] = 0.0
    return sp


def cpu_baseline(nsmax, kf_full, budget_s=20.0, gpu=None):
    """Oracle (C restatement of the reference CPU path, OpenMP) on this host's cores, on a bounded sample of the same
    workload: same grid/truncation, fewer fields -- the DENSE seed-20251114 spectrum (the loops do not depend on the
    data); pairs/s scaled linearly in the field count (flops and bytes are exactly linear in KF).
    gpu = (et, kresol, device): the sampled fields also go through the HIP path, and the oracle -- here the checker --
    gives the `dense` sub-record of the bench line: errors of both directions, spectral norms, and the
    benchmark's own drift criterion on a dense field."""
    from oracle.oracle import Oracle
    t0 = time.time()
    o = Oracle(nsmax, octahedral(nsmax))
    t_setup = time.time() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    rng = np.random.default_rng(20251114)
    keep = {}

    def pair(nf):
        v, d = (random_spectrum(rng, o.nasm0, nsmax, o.nspec2, nf, True) for _ in range(2))
        sc = random_spectrum(rng, o.nasm0, nsmax, o.nspec2, nf, False)
        t = time.time()
        g = o.inv_trans(spvor=v, spdiv=d, spsc=sc)
        out = o.dir_trans(g, nuv=nf, nsc=nf)
        dt = time.time() - t
        keep.update(nf=nf, v=v, d=d, sc=sc, g=g, out=out)
        return dt

    t1 = pair(1)  # KF = 3 (u, v, one scalar)
    nf = int(max(1, min(64, budget_s / max(t1, 1e-3))))
    t = pair(nf) if nf > 1 else t1
    kf = 3 * nf
    blas = None
    if os.environ.get("EMI_BENCH_BLAS", "1") != "0":
        try:
            nf_ = keep["nf"]
            blas = cpu_baseline_blas(o, nsmax, kf_full, keep["sc"], keep["g"][2 * nf_:3 * nf_], keep["out"][2], cores)
        except Exception as e:  # the port stays the reported baseline if the library leg cannot run on this host
            blas = {"error": repr(e)}
    base = {"value": (1.0 / t) * kf / kf_full, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "same grid+truncation, dense spectrum, %d of %d Fourier fields (vor/div/scalar x %d), scaled linearly in KF; "
                      "oracle setup %.1fs not included" % (kf, kf_full, nf, t_setup)}
    dense = None
    if gpu is not None:
        import torch
        et, r, dev = gpu
        nf = keep["nf"]
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        rel = lambda a, b: float((np.abs(a - b).max(axis=-1) / np.abs(b).max(axis=-1)).max())
        gp = torch.zeros((1, 3 * nf, o.ngptot), dtype=torch.float64, device=dev)
        et.inv_trans(r, pspvor=to(keep["v"]), pspdiv=to(keep["d"]), pspscalar=to(keep["sc"]), pgp=gp)
        e_inv = rel(gp[0].cpu().numpy(), keep["g"])
        v2, d2, s2 = (torch.zeros((o.nspec2, nf), dtype=torch.float64, device=dev) for _ in range(3))
        et.dir_trans(r, pspvor=v2, pspdiv=d2, pspscalar=s2, pgp=to(keep["g"][None]))
        e_dir = max(rel(a.cpu().numpy().T, b.T) for a, b in zip((v2, d2, s2), keep["out"]))
        e_norm = max(float(np.abs(et.specnorm(r, a) / o.specnorm(b) - 1.0).max()) for a, b in zip((v2, d2, s2), keep["out"]))
        # ectrans-benchmark.F90:743-756 on the dense scalars: |norm(x0) / norm(dir(inv(x0))) - 1| (octahedral grids: the
        # round trip carries the grid's own truncation error)
        drift = float(np.abs(et.specnorm(r, to(keep["sc"])) / et.specnorm(r, s2) - 1.0).max())
        dense = {"fields": kf, "checker": "oracle (CPU restatement), same inputs", "inv_max_rel_err": e_inv, "dir_max_rel_err": e_dir,
                 "spectral_norm_rel_err_vs_oracle": e_norm, "spectral_norm_rel_error_round_trip": drift}
    return base, dense, blas


def cpu_baseline_blas(o, nsmax, kf_full, sc_ref, g_ref, s_ref, cores, nf_total=256):
    """A second CPU baseline on LIBRARY BLAS + FFT -- the closest stand-in for the reference's FFTW + BLAS path that may travel
    (north_star; ledir_mod.F90:130,204 / leinv_mod.F90:133,166 call DGEMM, tpm_fftw.F90:294-316 FFTW) -- parallelised the way the
    reference parallelises it: a pool of threads over the zonal wavenumbers with a SEQUENTIAL BLAS call per thread
    (ltdir_ctl_mod.F90:90-98, ltinv_ctl_mod.F90:118-138: !$OMP PARALLEL DO SCHEDULE(DYNAMIC,1) over m) and over the latitudes for the
    Fourier transforms (ftdir_ctl_mod.F90:182-190), on ALL host cores, with >= 512 columns per panel product (256 fields x re / im).
    Legendre transforms = torch.matmul (MKL, one thread per call) on the ORACLE's panels, Fourier transforms = scipy.fft (pocketfft)
    per latitude; scalar fields only (the wind stencils are not where the time goes).  The first columns are the oracle's own sample
    and must reproduce its results; pairs/s scaled linearly in the field count, as cpu_baseline."""
    import scipy.fft
    import torch
    from concurrent.futures import ThreadPoolExecutor
    torch.set_num_threads(1)  # every worker thread runs its products on one BLAS thread, as the reference's OpenMP threads do
    nloen = octahedral(nsmax)
    ndgl, H = len(nloen), len(nloen) // 2
    nasm0, nmen, ndglu, rw = o.nasm0, o.nmen, o.ndglu, o.rw
    off = np.concatenate([[0], np.cumsum(nloen)])
    nref = sc_ref.shape[1]
    rng = np.random.default_rng(7)
    nf = max(nf_total, nref)
    sc = np.concatenate([sc_ref, random_spectrum(rng, nasm0, nsmax, o.nspec2, nf - nref, False)], axis=1) if nf > nref else sc_ref
    pool = ThreadPoolExecutor(max_workers=cores)
    # the panels as the BLAS operands (set-up, not timed: the reference keeps RPNMA / RPNMS in memory too)
    def panels(m):
        ps, pa = o.rpnm(m, True), o.rpnm(m, False)  # [column: n descending][latitude]
        # [k][lat], n = m + 2 k (+ 1); the panels also hold the row n = N + 1 the wind stencils use: dropped here (scalars only)
        return torch.from_numpy(ps[::-1][:(nsmax - m) // 2 + 1].copy()), torch.from_numpy(pa[::-1][:(nsmax - m + 1) // 2].copy())

    PP = list(pool.map(panels, range(nsmax + 1)))
    Ps, Pa = [p[0] for p in PP], [p[1] for p in PP]

    def coeffs(spec, m):
        i0 = nasm0[m] - 1
        z = spec[i0:i0 + 2 * (nsmax - m + 1)].reshape(nsmax - m + 1, 2, -1)  # [n - m][re | im][field]
        return torch.from_numpy(np.ascontiguousarray(z[0::2].reshape(-1, 2 * z.shape[2]))), torch.from_numpy(np.ascontiguousarray(z[1::2].reshape(-1, 2 * z.shape[2])))

    tm = {}

    # work arrays that persist between calls, as the reference's with LDALLOPERM = .TRUE. (what its benchmark sets,
    # ectrans-benchmark.F90:378-381): allocated and touched once, outside the timed pair
    buf = {}

    def work(name, shape):
        a = buf.get(name)
        if a is None or a.shape != shape:
            a = buf[name] = np.zeros(shape)
        return a

    def inverse(spec):
        nfl = spec.shape[1]
        FN, FS = work("FN", (H, nsmax + 1, 2 * nfl)), work("FS", (H, nsmax + 1, 2 * nfl))

        def wave(m):
            K = int(min(H, ndglu[m]))
            if K <= 0:
                return
            xs, xa = coeffs(spec, m)
            S = (Ps[m].T @ xs).numpy()  # LEINV: (K x ILS) (ILS x 2 KF)
            A = (Pa[m].T @ xa).numpy() if xa.shape[0] else 0.0
            FN[H - K:, m], FS[H - K:, m] = S + A, S - A  # ASRE1B

        t0 = time.time()
        list(pool.map(wave, range(nsmax + 1)))
        tm["inverse_legendre"] = time.time() - t0
        # grid-point fields with the FIELD index fastest, as the reference's ZGTF(KF, NLENGTF) (ftinv_ctl_mod.F90:145-156): a latitude row
        # is one contiguous block and the transform runs along its first axis -- no strided Python-side copies around the library call
        grid = work("grid", (int(off[-1]), nfl))

        def row(j):
            n, jn = int(nloen[j]), (j if j < H else ndgl - 1 - j)
            M = int(min(nmen[j], n // 2))
            F = (FN if j < H else FS)[jn, :M + 1].reshape(M + 1, 2, nfl)  # [m][re | im][field]
            X = np.zeros((n // 2 + 1, nfl), dtype=np.complex128)
            X[:M + 1].real, X[:M + 1].imag = F[:, 0], F[:, 1]
            grid[off[j]:off[j + 1]] = scipy.fft.irfft(X, n, axis=0) * n  # FTINV: c2r, unscaled

        t0 = time.time()
        list(pool.map(row, range(ndgl)))
        tm["inverse_fourier"] = time.time() - t0
        return grid

    def direct(grid):
        nfl = grid.shape[1]
        FN, FS = work("FN", (H, nsmax + 1, 2 * nfl)), work("FS", (H, nsmax + 1, 2 * nfl))

        def row(j):
            n, jn = int(nloen[j]), (j if j < H else ndgl - 1 - j)
            M = int(min(nmen[j], n // 2))
            X = scipy.fft.rfft(grid[off[j]:off[j + 1]], axis=0)[:M + 1] * (rw[j] / n)  # FTDIR scaled 1 / NLOEN, times the Gaussian weight
            Fd = (FN if j < H else FS)[jn, :M + 1].reshape(M + 1, 2, nfl)
            Fd[:, 0], Fd[:, 1] = X.real, X.imag

        t0 = time.time()
        list(pool.map(row, range(ndgl)))
        tm["direct_fourier"] = time.time() - t0
        spec = work("spec", (o.nspec2, nfl))

        def wave(m):
            K = int(min(H, ndglu[m]))
            if K <= 0:
                return
            fn, fs = torch.from_numpy(FN[H - K:, m]), torch.from_numpy(FS[H - K:, m])
            xs = (Ps[m] @ (fn + fs)).numpy().reshape(-1, 2, nfl)  # LEDIR: (ILS x K) (K x 2 KF)
            xa = (Pa[m] @ (fn - fs)).numpy().reshape(-1, 2, nfl)
            i0 = nasm0[m] - 1
            z = spec[i0:i0 + 2 * (nsmax - m + 1)].reshape(nsmax - m + 1, 2, nfl)
            z[0::2], z[1::2] = xs, xa
            if m == 0:
                z[:, 1] = 0.0

        t0 = time.time()
        list(pool.map(wave, range(nsmax + 1)))
        tm["direct_legendre"] = time.time() - t0
        return spec

    # guard: the leg must stay a bounded sample whatever the host's BLAS threading does -- a probe of `cores` panel products first
    t = time.time()
    list(pool.map(lambda m: Ps[m].T @ coeffs(sc, m)[0], range(0, min(cores, nsmax + 1))))
    t_probe = time.time() - t
    if t_probe * (4.0 * (nsmax + 1) / max(1, min(cores, nsmax + 1))) > 240.0:
        pool.shutdown()
        return {"error": "skipped: %d concurrent panel products took %.1f s on this host" % (min(cores, nsmax + 1), t_probe)}
    # one untimed pair first, as the reference harness's warm-up iterations (ectrans-benchmark.F90:47-50): the FFT plans of every
    # row length, the BLAS threads' buffers and the first touch of the work arrays are set-up, not transform time
    direct(inverse(sc))
    # two timed pairs, the faster one quoted: the legs of one pair vary by a factor of three between hosts of the pool (thread placement of a
    # 256-thread pool on a shared host), and a baseline is owed its best run
    dt, t_inv, tm_best = None, None, None
    for _ in range(2):
        t = time.time()
        g = inverse(sc)
        ti = time.time() - t
        s2 = direct(g)
        d = time.time() - t
        if dt is None or d < dt:
            dt, t_inv, tm_best = d, ti, dict(tm)
    tm = tm_best
    rel = lambda a, b: float((np.abs(a - b).max(axis=-1) / np.abs(b).max(axis=-1)).max())
    e_inv, e_dir = rel(g[:, :nref].T, g_ref), rel(s2[:, :nref].T, s_ref.T)
    pool.shutdown()
    return {"value": (1.0 / dt) * nf / kf_full, "unit": "pairs/s", "cores": cores, "kind": "library BLAS + FFT on the oracle's panels",
            "libraries": "torch.matmul (MKL, sequential) for LEINV / LEDIR on a pool of %d threads over the zonal wavenumbers (the reference: OpenMP over m, "
                         "sequential DGEMM per thread), scipy.fft (pocketfft) per latitude on the same pool" % cores,
            "sample": "same grid+truncation, dense spectrum, %d scalar fields (%d columns per panel product) of %d Fourier fields, scaled linearly in KF; "
                      "panels and thread pool set up outside the timed pairs; one warm-up pair, the faster of two timed pairs" % (nf, 2 * nf, kf_full),
            "seconds_at_sample": dict(tm, inverse=t_inv, direct=dt - t_inv),
            "inv_max_rel_err_vs_oracle": e_inv, "dir_max_rel_err_vs_oracle": e_dir}


def api_level(et, r, N, kf_full, esz, nf=128, pairs=2):
    """The API-level (PCIe-inclusive) rate SURVEY 8d asks for beside the device-resident one: numpy arrays in host memory
    through EMI_MEM_HOST (what a Fortran / C caller of the reference passes), `nf` scalar fields, scaled linearly to the
    full field count.  PINNED memory (the benchmark pins its fields too, ectrans-benchmark.F90:207-209) is the quoted
    figure; the same call on PAGEABLE memory (an ordinary ALLOCATE; the runtime pins it in place) is reported beside
    it.  Never `value`."""
    import numpy as np
    import torch
    ns2, ng = et.trans_inq(r, "nspec2"), et.trans_inq(r, "ngptot")
    dt = torch.float64 if esz == 8 else torch.float32
    i419 = int(et.trans_inq(r, "nasm0")[4]) - 1 + 2 * (19 - 4)

    def run(sp, gp):
        sp[i419] = 1.0

        def pair():
            et.inv_trans(r, pspscalar=sp, pgp=gp)
            et.dir_trans(r, pspscalar=sp, pgp=gp)

        pair()
        t = time.perf_counter()
        for _ in range(pairs):
            pair()
        return (time.perf_counter() - t) / pairs, abs(float(sp[i419, 0]) - 1.0)

    sp = torch.zeros((ns2, nf), dtype=dt).pin_memory().numpy()
    gp = torch.zeros((1, nf, ng), dtype=dt).pin_memory().numpy()
    moved = 2.0 * (sp.nbytes + gp.nbytes)
    t, chk = run(sp, gp)
    del sp, gp
    npdt = np.float64 if esz == 8 else np.float32
    tp, chkp = run(np.zeros((ns2, nf), dtype=npdt), np.zeros((1, nf, ng), dtype=npdt))
    return {"kf": nf, "pairs_per_s_at_kf": 1.0 / t, "pairs_per_s_scaled_to_full_kf": (1.0 / t) * nf / kf_full, "ms_per_pair_at_kf": t * 1e3,
            "host_device_GB_per_pair": moved / 1e9, "effective_GBps": moved / 1e9 / t, "memory": "pinned host (torch pin_memory), EMI_MEM_HOST",
            "harmonic_check": chk,
            "pageable": {"ms_per_pair_at_kf": tp * 1e3, "effective_GBps": moved / 1e9 / tp, "harmonic_check": chkp,
                         "memory": "pageable host (numpy), EMI_MEM_HOST"}}


def fortran_device_resident(N, nlev, nfld, steps, warmup, ms_step_python):
    """The SAME pair driven from Fortran: ectrans_amd/fortran/emi_bench_host.F90 -- the reference harness's timed loop
    (ectrans-benchmark.F90:619-769) over the drop-in shim's INV_TRANS / DIR_TRANS (libectrans_mi_f.so, the reference's keyword
    interfaces) on call-mode-2 arrays that live in device memory (hipMalloc + C_F_POINTER; the shim passes EMI_MEM_AUTO and the
    library uses them in place).  A child process: it initialises the library and allocates its own fields.  Wall time per pair
    measured in Fortran with SYSTEM_CLOCK around calls that return when the fields are there."""
    import subprocess
    exe = os.path.join(ROOT, "ectrans_amd", "fortran", "emi_bench_host")
    if not os.path.exists(exe):
        return {"error": "ectrans_amd/fortran/emi_bench_host is not built (__graft_entry__.build())"}
    try:
        p = subprocess.run([exe, str(N), str(nlev), str(nfld), str(steps), str(warmup)], capture_output=True, text=True, timeout=900)
    except subprocess.TimeoutExpired:
        return {"error": "emi_bench_host timed out"}
    line = [l for l in p.stdout.splitlines() if l.startswith("EMI_BENCH_HOST")]
    if p.returncode != 0 or not line:
        return {"error": "emi_bench_host failed (rc %d): %s" % (p.returncode, (p.stdout + p.stderr)[-400:])}
    import re
    kv = dict(re.findall(r"(\w+)=\s*([-+0-9.eE]+)", line[0]))
    avg = float(kv["ms_per_pair_avg"])
    return {"host": "Fortran (flang) -> libectrans_mi_f.so (shim, reference dummy-argument lists) -> libectrans_mi.so; arrays: hipMalloc + C_F_POINTER, EMI_MEM_AUTO",
            "pairs_per_s": 1e3 / avg, "ms_per_pair": avg, "ms_per_pair_median": float(kv["ms_per_pair_median"]), "ms_per_pair_min": float(kv["ms_min"]),
            "ms_per_pair_max": float(kv["ms_max"]), "steps": steps, "warmup": warmup, "kf": int(kv["kf"]), "spectral_norm_rel_error": float(kv["norm_err"]),
            "setup_s": float(kv["setup_s"]), "vs_python_host_ms_per_step": avg / ms_step_python}


def recorded_fft_bound(N, nlev, nfld, esz, world, source_hash):
    """The FFT phase against the bound it is actually on: SIMD issue time -- vector-ALU + LDS + other
    instruction issue of the FFT launches as a share of their duration, from the SQ counters of tools/pmc_fft.sh, quoted only
    when the counter file was taken on this very build of the library (same source hash)."""
    import glob
    import json
    if (N, nlev, nfld, world) != (1279, 137, 10, 1):
        return None
    here = os.path.dirname(os.path.abspath(__file__))
    pat = "*_pmc_fft.json" if esz == 8 else "*_pmc_fft_fp32.json"  # tools/pmc_fft.sh TAG 4 writes the fp32 library's table beside the fp64 one
    for f in sorted(glob.glob(os.path.join(here, "profiles", pat)), reverse=True):
        try:
            js = json.load(open(f))
        except Exception:
            continue
        if js.get("source_hash") == source_hash and int(js.get("precision", 8)) == esz:
            return {"bound": "SIMD instruction issue (fp%d vector ALU + LDS + other), chip clock lowered under this load" % (8 * esz),
                    "simd_issue_share": js["simd_issue_share"], "wave_life_share": js["wave_life_share"],
                    "nonfp_valu_share": js.get("nonfp_valu_share"), "nonfp_valu_share_six_heaviest": js.get("nonfp_valu_share_six_heaviest"), "fft_ms_per_pair_under_profiler": js["fft_ms_per_pair"],
                    "source": os.path.basename(f)}
    return {"bound": None, "source": "no profiles/%s for this build (source hash %s): re-run tools/pmc_fft.sh" % (pat, source_hash)}


def recorded_traffic(N, nlev, nfld, esz, world, source_hash):
    """HBM bytes per Legendre launch from the committed PMC passes (profiles/*_pmc_traffic.json, collected with
    tools/collect_profiles.sh on this exact workload; rocprofv3 cannot run inside the timed region).  Reported only for
    the workload AND the library build the counters were taken on (the file is stamped with ectrans_amd.source_hash());
    a stale file gives null and says so."""
    if (N, nlev, nfld, world) != (1279, 137, 10, 1):
        return None, "counters are collected on the default workload (TCo1279 137L x 10, one GPU) only"
    import glob
    pat = "*_pmc_traffic.json" if esz == 8 else "*_pmc_traffic_fp32.json"  # EMI_COLLECT_PRECISION=4 bash tools/collect_profiles.sh TAG
    ns = "emi_f64::" if esz == 8 else "emi_f32::"
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pat)))
    for f in reversed(files):
        try:
            j = json.load(open(f))
            if j.get("source_hash") != source_hash:
                continue
            k = j["kernels"]
            vals = [k[ns + n]["hbm_bytes_per_launch"] for n in ("k_leg_inv", "k_leg_dir")]
            note = os.path.basename(f)
            if esz == 4:  # the guide calibrates FETCH_SIZE x 2 for 16-byte-per-lane loads only; the fp32 k_leg_dir loads 8 bytes per lane
                note += " (fp32 k_leg_dir loads 8 B per lane: its read bytes = 2 x FETCH_SIZE are an upper bound, FETCH_SIZE itself the lower one)"
            return sum(vals) / len(vals), note
        except (KeyError, ValueError, OSError):
            continue
    return None, "no profiles/%s for this build (source hash %s): re-run tools/collect_profiles.sh" % (pat, source_hash)


def exchange_report(rs, world, nprtrw, nprtrv, steps):
    """The multi-GPU part of the bench line: what each rank spent where, and what the exchange cost.
    rs[rank] = [ms per step (host wall clock of the rank), pack, Legendre, FFT, exchange (ms per step, HIP events on the stream each phase
    runs on), bytes sent per step, exchanges per step].  The exchange of one field batch runs beside the kernels of its neighbours
    (three streams), so the phases of a rank add up to MORE than its step: overlap_frac = the share of the exchange time that is hidden,
    (sum of phases - step) / exchange, clipped to 0..1, on the slowest rank.  Bytes per link: xGMI is point to point, a task's blocks to
    its NPRTRW - 1 W-set peers leave over that many links side by side (TRMTOL / TRLTOM: trmtol_mod.F90:101-119, trltom_mod.F90:96-114)."""
    import numpy as np
    rs = np.asarray(rs, dtype=float)
    slow = int(np.argmax(rs[:, 0]))
    step, pack, leg, fft, xch, xb, xn = rs[slow]
    hidden = pack + leg + fft + xch - step
    peers = max(nprtrw - 1, 1)
    names = ("ms_per_step", "spectral_pack_unpack", "legendre_mfma", "fft", "exchange")
    return {
        "exchange_ms_per_step": float(rs[:, 4].max()),
        "overlap_frac": float(min(1.0, max(0.0, hidden / xch))) if xch > 0 else None,
        "rank_phase_ms_per_step": {n: {"min": float(rs[:, i].min()), "max": float(rs[:, i].max()), "mean": float(rs[:, i].mean())} for i, n in enumerate(names)},
        "exchange": {"exchanges_per_step": float(rs[:, 6].max()), "bytes_sent_per_rank_and_step": float(rs[:, 5].max()),
                     "bytes_per_link_and_direction_per_exchange": float(rs[:, 5].max() / max(rs[:, 6].max(), 1.0) / peers),
                     "bytes_per_link_per_step": float(rs[:, 5].max() / peers),
                     "achieved_GBps_per_link": float(rs[:, 5].max() / peers / 1e9 / max(rs[:, 4].max() * 1e-3, 1e-12)),
                     "xgmi_link_peak_GBps_per_direction": 76.8, "links_used_per_rank": peers,
                     "slowest_rank": slow, "compute_ms_per_step_slowest_rank": float(pack + leg + fft),
                     "note": "exchange time = HIP events around the all-to-all-v hook on the exchange stream (includes waiting for the peers); "
                             "with V-sets the grid-space exchange (TRLTOG / TRGTOL) is counted too"},
    }


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
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs: cpu_baseline and the dense sub-record")
    ap.add_argument("--no-api-level", action="store_true", help="skip the host-array (PCIe-inclusive) measurement")
    ap.add_argument("--no-dense-timing", action="store_true", help="skip the second timed loop on the dense spectrum")
    ap.add_argument("--no-fortran", action="store_true", help="skip the Fortran host leg (fortran_device_resident)")
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--nprtrv", type=int, default=int(os.environ.get("EMI_BENCH_NPRTRV", "1")),
                    help="V-sets (N > 1 only): NPRTRW = N / NPRTRV; levels dealt to the V-sets as ectrans-benchmark.F90:329-344, 440-447")
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
    # N > 1: the exchange runs on the native RCCL transport (ectrans_amd/rccl: one group of ncclSend / ncclRecv per field batch
    # on the library's exchange stream -- the path a Fortran host uses); EMI_BENCH_TRANSPORT=torch keeps the Python
    # all_to_all_single callback, which is also the fallback when the native attach fails (and the gloo test configuration)
    transport = "torch"
    if world > 1 and backend == "nccl" and os.environ.get("EMI_BENCH_TRANSPORT", "rccl") == "rccl":
        transport = "rccl"
    transport_note = None
    ok = 1
    nprtrv = max(1, args.nprtrv)
    if world % nprtrv:
        raise SystemExit("bench.py: --nprtrv %d does not divide %d tasks" % (nprtrv, world))
    nprtrw = world // nprtrv
    mysetv = rank % nprtrv + 1
    try:
        et.setup_trans0(kmax_resol=2, device=local, kprtrw=nprtrw, kprtrv=nprtrv, myproc=rank + 1, transport=transport)
    except OSError as e:
        if transport != "rccl":
            raise
        ok, transport_note = 0, str(e)
    preflight = None
    if transport == "rccl" and world > 1:
        # agree on the attach BEFORE the pre-flight: its grouped ncclSend / ncclRecv would wait for ever on a rank whose attach
        # failed and which therefore never posts its side
        flag0 = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag0, op=dist.ReduceOp.MIN)
        if not int(flag0.item()) and ok:
            ok, transport_note = 0, "native attach failed on another rank"
    if (transport == "rccl" or os.environ.get("EMI_BENCH_PREFLIGHT")) and ok and world > 1:
        # pre-flight of the native exchange (its grouped ncclSend / ncclRecv have only ever run inside one-GPU tests): a T63 pair
        # of the benchmark harmonic on two fields; an exception or a wrong norm sends every rank to the torch.distributed callback
        try:
            n_ = 63
            rr = et.setup_trans(n_, 2 * (n_ + 1), octahedral(n_), precision=args.precision)
            ns_, ng_ = et.trans_inq(rr, "nspec2"), et.trans_inq(rr, "ngptot")
            a4_ = int(et.trans_inq(rr, "nasm0")[4])
            dt_ = torch.float64 if args.precision == 8 else torch.float32
            sp_ = torch.zeros((ns_, 2), dtype=dt_, device=dev)
            if a4_ > 0:
                sp_[a4_ - 1 + 2 * (19 - 4)] = 1.0
            gp_ = torch.zeros((1, 2, ng_), dtype=dt_, device=dev)
            kw_ = {"kvsetsc": [1, 1]} if nprtrv > 1 else {}
            if nprtrv > 1 and mysetv != 1:
                sp_ = sp_[:, :0]
            na_ = et.specnorm(rr, sp_, **({"kvset": [1, 1]} if nprtrv > 1 else {}))[0]
            et.inv_trans(rr, pspscalar=sp_ if sp_.shape[1] else None, pgp=gp_, **kw_)
            et.dir_trans(rr, pspscalar=sp_ if sp_.shape[1] else None, pgp=gp_, **kw_)
            torch.cuda.synchronize()
            nb_ = et.specnorm(rr, sp_, **({"kvset": [1, 1]} if nprtrv > 1 else {}))[0]
            et.trans_release(rr)
            preflight = "ok"
            if not (na_ > 0 and abs(nb_ / na_ - 1.0) < (1e-10 if args.precision == 8 else 1e-4)):
                ok, transport_note = 0, "exchange pre-flight: norm %r -> %r" % (na_, nb_)
                preflight = transport_note
        except Exception as e:  # noqa: BLE001 -- whatever it is, the other transport is the answer
            ok, transport_note = 0, "exchange pre-flight: %r" % (e,)
            preflight = transport_note
    if transport == "rccl":
        if world > 1:  # every rank must take the same path: fall back together if the native attach failed anywhere
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            all_ok = int(flag.item())
        else:
            all_ok = ok
        if not all_ok:
            if et.lib().emi_inq_tasks(None, None) == 0:  # the library is attached on this rank: detach before the other transport
                et.trans_end()
                from ectrans_amd import dist as _ed
                _ed.rccl_native_lib().emi_rccl_detach()
            transport_note = transport_note or "native attach failed on another rank"
            transport = "torch"
            et.setup_trans0(kmax_resol=2, device=local, kprtrw=nprtrw, kprtrv=nprtrv, myproc=rank + 1)
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

    # call mode 2 arrays of the reference harness (ectrans-benchmark.F90:450-479); with V-sets the levels go to them in
    # contiguous runs and the surface field to V-set min(nlev + 1, NPRTRV) (:329-344, 440-447)
    kv = {}
    nlevl, nsc2l = nlev, 1
    if nprtrv > 1:
        numll = [nlev // nprtrv + (1 if v < nlev % nprtrv else 0) for v in range(nprtrv)]
        ivset = [v + 1 for v in range(nprtrv) for _ in range(numll[v])]
        ivsetsc2 = [min(nlev + 1, nprtrv)]
        kv = {"kvsetuv": ivset, "kvsetsc3a": ivset, "kvsetsc2": ivsetsc2}
        nlevl, nsc2l = numll[mysetv - 1], int(ivsetsc2[0] == mysetv)
    spvor, spdiv, spsc3a, spsc2 = z(nspec2, nlevl), z(nspec2, nlevl), z(nfld, nspec2, nlevl), z(nspec2, nsc2l)
    if i419 is not None:
        for a in (spvor, spdiv, spsc2):
            a[i419] = 1.0
        spsc3a[:, i419] = 1.0
    gpuv, gp3a, gp2 = z(1, 2, nlev, ngptot), z(1, nfld, nlev, ngptot), z(1, 1, ngptot)
    n0 = et.specnorm(r, spsc2, **({"kvset": kv["kvsetsc2"]} if kv else {}))[0]

    def step():
        et.inv_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2, **kv)
        et.dir_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2, **kv)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(steps):
        """exactly `steps` pairs, bracketed by barrier + synchronize; HIP-event phase timers run inside and are only resolved after
        the region (accumulating mode): no host synchronisation between the calls.  One event per step boundary on the stream the
        calls are queued on (the null stream = torch's default stream) gives the per-step times for the median the reference
        reports (ectrans-benchmark.F90:906-917)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        et.set_profile(2)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            ev[i].record()
            et.inv_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2, **kv)
            et.dir_trans(r, pspvor=spvor, pspdiv=spdiv, pspsc3a=spsc3a, pspsc2=spsc2, pgpuv=gpuv, pgp3a=gp3a, pgp2=gp2, **kv)
        ev[steps].record()
        barrier()
        dt = time.perf_counter() - t0
        sm = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
        med = sm[len(sm) // 2] if len(sm) % 2 else 0.5 * (sm[len(sm) // 2 - 1] + sm[len(sm) // 2])
        phases = et.last_phase_ms()
        launches = et.last_phase_launches()[1]
        xch = et.last_exchange()          # (ms, calls, bytes sent by this rank) of the all-to-all-v exchanges of the loop
        fftk = et.last_fft_launches()     # kernel launches of the FFT phases of the loop
        et.set_profile(0)
        return dt, sm, med, phases, launches, xch, fftk

    for _ in range(args.warmup):
        step()
    # ---- timed region: the reference harness's input (one harmonic), exactly K steps
    dt, step_ms, med_ms, (pack_ms, leg_ms, fft_ms), leg_launches, (xch_ms, xch_calls, xch_bytes), fft_kernels = timed_loop(args.steps)
    n1 = et.specnorm(r, spsc2, **({"kvset": kv["kvsetsc2"]} if kv else {}))[0]

    # ---- the same loop on a DENSE spectrum: every coefficient of every field ~ U(-0.5, 0.5) / (n + 1) from
    # a seed-20251114 generator, every field distinct, imag(m = 0) = 0, (0, 0) of vor / div = 0 -- filled on the device into the same
    # arrays.  99.9 % of the harmonic loop's Legendre products multiply zeros; this one shows whether the kernels (and the chip's
    # clock) care.  `value` stays the harmonic loop, the reference's metric (ectrans-benchmark.F90:1390-1415).
    dense_t = None
    if not args.no_dense_timing:
        nasm0_l = et.trans_inq(r, "nasm0")
        inv_np1 = np.zeros(nspec2)
        i00 = None
        for m in range(N + 1):
            if nasm0_l[m] > 0:
                i0 = int(nasm0_l[m]) - 1
                inv_np1[i0:i0 + 2 * (N - m + 1)] = 1.0 / (np.repeat(np.arange(m, N + 1), 2) + 1.0)
                if m == 0:
                    inv_np1[i0 + 1:i0 + 2 * (N + 1):2] = 0.0  # imag(m = 0)
                    i00 = i0
        prof = torch.from_numpy(inv_np1).to(device=dev, dtype=spvor.dtype)
        gen = torch.Generator(device=dev)
        gen.manual_seed(20251114 + rank)

        def fill(a, spec_axis, zero00):
            shape = [1] * a.dim()
            shape[spec_axis] = nspec2
            for lo in range(0, a.shape[-1], 8):  # in slices of the last axis: no second full-size temporary
                v = a[..., lo:lo + 8]
                v.copy_((torch.rand(v.shape, generator=gen, device=dev, dtype=a.dtype) - 0.5) * prof.view(shape))
            if zero00 and i00 is not None:
                a.select(spec_axis, i00).zero_()

        fill(spvor, 0, True), fill(spdiv, 0, True), fill(spsc3a, 1, False), fill(spsc2, 0, False)
        nd0 = et.specnorm(r, spsc2, **({"kvset": kv["kvsetsc2"]} if kv else {}))[0]
        step()
        d_dt, d_sm, d_med, (d_pack, d_leg, d_fft), d_launch, _, _ = timed_loop(args.steps)
        nd1 = et.specnorm(r, spsc2, **({"kvset": kv["kvsetsc2"]} if kv else {}))[0]
        dense_t = {"dt": d_dt, "med": d_med, "pack": d_pack, "leg": d_leg, "fft": d_fft, "launches": d_launch, "drift": abs(nd0 / nd1 - 1.0)}
    kf_l = 2 * nlevl + nfld * nlevl + nsc2l  # Legendre / Fourier-space fields of this rank (= kf without V-sets)
    wm = et.work_model(r, kf_l)
    # algorithmic HBM bytes of this rank (each array touched once per phase it belongs to, both directions)
    nmen_sum = float(wm["fourier_bytes"]) / (2.0 * esz * max(kf_l, 1))  # Fourier rows (lat, m <= NMEN) of this rank
    ndglu = et.trans_inq(r, "ndglu")
    myms = et.trans_inq(r, "myms")
    pan_bytes = float(sum(int(min(N + 1, ndglu[m])) * (N - m + 2) for m in myms)) * esz  # Legendre matrices, read once per direction
    spec_bytes = float(nspec2) * kf_l * esz
    grid_bytes = float(ngptot) * kf * esz
    # per Legendre launch: packed spectral + panels + Fourier rows (read or written once)
    alg_leg_bytes = spec_bytes + pan_bytes + wm["fourier_bytes"]
    alg_pair_bytes = 2.0 * (grid_bytes + 2.0 * wm["fourier_bytes"] + 2.0 * spec_bytes + pan_bytes)
    rank_stats = None
    if world > 1:
        # what makes the first multi-GPU run self-explaining: every rank's own numbers, gathered BEFORE the maxima below replace them
        mine = torch.tensor([dt / args.steps * 1e3, pack_ms / args.steps, leg_ms / args.steps, fft_ms / args.steps, xch_ms / args.steps,
                             xch_bytes / max(args.steps, 1), float(xch_calls) / max(args.steps, 1)], dtype=torch.float64, device=rdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_stats = torch.stack(allr).cpu().numpy()  # [rank][ms_per_step, pack, legendre, fft, exchange, bytes sent per step, exchanges per step]
        tt = torch.tensor([dt, med_ms], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, med_ms = float(tt[0]), float(tt[1])
        # whole-job Legendre rate: flops of all ranks / slowest rank's kernel time
        red = torch.tensor([wm["legendre_flops"], wm["fourier_bytes"], float(ngptot), alg_leg_bytes, alg_pair_bytes], dtype=torch.float64, device=rdev)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        mx = torch.tensor([leg_ms, fft_ms, pack_ms], dtype=torch.float64, device=rdev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        wm["legendre_flops"], wm["fourier_bytes"], ngptot = float(red[0]), float(red[1]), int(red[2].item())
        alg_leg_bytes, alg_pair_bytes = float(red[3]), float(red[4])
        leg_ms, fft_ms, pack_ms = (float(x) for x in mx)
        if dense_t is not None:
            mxd = torch.tensor([dense_t[k] for k in ("dt", "med", "pack", "leg", "fft")], dtype=torch.float64, device=rdev)
            dist.all_reduce(mxd, op=dist.ReduceOp.MAX)
            dense_t.update(zip(("dt", "med", "pack", "leg", "fft"), (float(x) for x in mxd)))

    if rank == 0:
        # per launch: algorithmic flops of the launch (all ranks) / average launch duration (slowest rank)
        flops_per_launch = wm["legendre_flops"] * 2 * args.steps / max(leg_launches, 1)
        ms_per_launch = leg_ms / max(leg_launches, 1)
        ach = flops_per_launch / (ms_per_launch * 1e-3) / 1e12 if ms_per_launch > 0 else 0.0
        traffic, traffic_src = recorded_traffic(N, nlev, nfld, esz, world, et.source_hash())
        ms_step = dt / args.steps * 1e3
        # the pair against max(sum flops / MFMA peak, sum bytes / HBM bandwidth)
        t_flops = 2.0 * wm["legendre_flops"] / (peak * 1e12 * world)
        t_bytes = alg_pair_bytes / (8.0e12 * world)
        out = {
            "metric": "dir+inv transform-pairs/sec, TCo%d %dL x %d fields; spectral-norm rel-error" % (N, nlev, nfld),
            "value": args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64" if esz == 8 else "f32", "data": "synthetic",
            "config": {"workload": "TCo%d/O%d, %d levels x %d 3-D fields + vor/div + 1 surface field, KF=%d, "
                                   "device-resident call-mode-2 arrays" % (N, N + 1, nlev, nfld, kf),
                       "parallelism": "1 GPU" if world == 1 else
                       "%d GPUs (w%dv%d): zonal wavenumbers zig-zag + latitude bands%s, all-to-all-v per direction over %s" % (
                           world, nprtrw, nprtrv, " x %d V-sets of levels (second all-to-all-v in grid space)" % nprtrv if nprtrv > 1 else "", ("RCCL (native grouped ncclSend / ncclRecv, ectrans_amd/rccl)" if transport == "rccl" else "RCCL (torch.distributed nccl)")
                           if backend == "nccl" else backend + " (test configuration, host staged)"),
                       "nprtrw": nprtrw, "nprtrv": nprtrv,
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "backend": None if world == 1 else ("rccl-native" if transport == "rccl" else backend),
                       "transport_note": transport_note, "exchange_preflight": preflight, "pipeline_batches": int(os.environ.get("EMI_PIPELINE_DIST", "4")) if world > 1 else 1,
                       "setup_s": round(t_setup, 2)},
            # the reference reports 1 / median step time (ectrans-benchmark.F90:906-943); `value` is steps / wall time
            "pairs_per_s_median": 1e3 / med_ms, "ms_per_step_median": med_ms, "ms_per_step_min": step_ms[0], "ms_per_step_max": step_ms[-1],
            "spectral_norm_rel_error": abs(n0 / n1 - 1.0),
            "roofline": {"bound": "mfma", "kernel": "k_leg_inv + k_leg_dir (fp%d MFMA Legendre transforms)" % (8 * esz),
                         "achieved": ach, "peak": peak * world, "unit": "TFLOP/s",
                         "frac": ach / (peak * world), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_leg_bytes,
                         "launches": leg_launches, "avg_launch_ms": ms_per_launch,
                         "algorithmic_flops_per_launch": flops_per_launch},
            "pair_roofline": {"bound_ms": 1e3 * max(t_flops, t_bytes), "mfma_bound_ms": 1e3 * t_flops, "hbm_bound_ms": 1e3 * t_bytes,
                              "algorithmic_flops_per_pair": 2.0 * wm["legendre_flops"], "algorithmic_bytes_per_pair": alg_pair_bytes,
                              "frac": 1e3 * max(t_flops, t_bytes) / ms_step},
            "phase_ms_per_step": {"spectral_pack_unpack": pack_ms / args.steps, "legendre_mfma": leg_ms / args.steps,
                                  "fft": fft_ms / args.steps},
            "fft_hbm": {"algorithmic_GB_per_step": 2 * (wm["fourier_bytes"] + kf * ngptot * float(esz)) / 1e9,
                        "achieved_GBps": 2 * (wm["fourier_bytes"] + kf * ngptot * float(esz)) / 1e9 / max(fft_ms / args.steps * 1e-3, 1e-9),
                        "peak_GBps": 8000.0 * world},
            "fft_bound": recorded_fft_bound(N, nlev, nfld, esz, world, et.source_hash()),
            # FFT kernel launches per direction and field batch (one per row-length class: ftdir_ctl_mod.F90:182-190 loops over latitudes instead)
            "fft_launches_per_direction": fft_kernels / max(2.0 * args.steps, 1.0),
        }
        if world > 1:
            out.update(exchange_report(rank_stats, world, nprtrw, nprtrv, args.steps))
        if dense_t is not None:
            d_ms = dense_t["dt"] / args.steps * 1e3
            d_ach = wm["legendre_flops"] * 2 * args.steps / max(dense_t["leg"] * 1e-3, 1e-12) / 1e12
            out["dense_timing"] = {
                "input": "every coefficient of every field ~ U(-0.5, 0.5) / (n + 1), seed 20251114 (device generator), imag(m=0) = 0, "
                         "vor/div (0,0) = 0; same arrays, same steps as the harmonic loop above",
                "ms_per_step": d_ms, "ms_per_step_median": dense_t["med"], "pairs_per_s": args.steps / dense_t["dt"],
                "phase_ms_per_step": {"spectral_pack_unpack": dense_t["pack"] / args.steps, "legendre_mfma": dense_t["leg"] / args.steps,
                                      "fft": dense_t["fft"] / args.steps},
                "roofline_frac": d_ach / (peak * world), "roofline_achieved_TFLOPs": d_ach,
                "vs_harmonic_ms_per_step": d_ms / ms_step,
                "spectral_norm_rel_error_round_trip": dense_t["drift"]}
        if world == 1 and not args.no_api_level:
            # free the device-resident benchmark arrays first: the staging buffers of the host calls need the room
            del spvor, spdiv, spsc3a, spsc2, gpuv, gp3a, gp2
            torch.cuda.empty_cache()
            out["api_level"] = api_level(et, r, N, kf, esz)
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.empty_cache()
            out["cpu_baseline"], out["dense"], out["cpu_baseline_blas"] = cpu_baseline(N, kf, gpu=(et, r, dev) if esz == 8 else None)
        if world == 1 and esz == 8 and not args.no_fortran:
            # last leg: this process gives the GPU back (the Fortran host allocates the same 200 GB itself)
            try:
                del spvor, spdiv, spsc3a, spsc2, gpuv, gp3a, gp2
            except NameError:
                pass
            et.trans_end()
            torch.cuda.empty_cache()
            try:
                out["fortran_device_resident"] = fortran_device_resident(N, nlev, nfld, args.steps, args.warmup, ms_step)
            except Exception as e:  # noqa: BLE001 -- a side record must never cost the bench line
                out["fortran_device_resident"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
