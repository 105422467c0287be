"""ectrans_amd -- Python host mirror of the ecTrans API over libectrans_mi.so (C-ABI, HIP).

The names and argument meaning follow the reference Fortran interface
(/root/reference/src/trans/include/ectrans/{setup_trans0,setup_trans,inv_trans,dir_trans,
trans_inq,specnorm}.h): ``setup_trans0``, ``setup_trans`` (returns KRESOL), ``inv_trans``,
``dir_trans``, ``trans_inq``, ``specnorm``, ``trans_release``, ``trans_end``.  Errors that the
reference reports through ABORT_TRANS are raised as :class:`TransError` with the same text.

Arrays are passed with the *Fortran* index order reversed (C-contiguous):
``PSPEC(nfld, nspec2)`` is a ``(nspec2, nfld)`` array, ``PGP(nproma, nfld, ngpblks)`` is
``(ngpblks, nfld, nproma)``, ``PSPSC3A(nlev, nspec2, nvar)`` is ``(nvar, nspec2, nlev)``,
``PGPUV(nproma, nlev, nvar, ngpblks)`` is ``(ngpblks, nvar, nlev, nproma)``.
numpy arrays are staged through PCIe (EMI_MEM_HOST); torch CUDA tensors are used in place
(EMI_MEM_DEVICE).  There is no CPU implementation in this package: without the HIP library
and a GPU every call fails loudly.
"""
import ctypes as C
import os

import numpy as np

__all__ = ["TransError", "setup_trans0", "setup_trans", "inv_trans", "dir_trans", "trans_inq", "specnorm",
           "trans_release", "trans_end", "lib", "build"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "libectrans_mi.so")
_L = None

EMI_MEM_HOST, EMI_MEM_DEVICE, EMI_MEM_AUTO = 0, 1, 2


class TransError(RuntimeError):
    """What the reference would have reported through ABORT_TRANS."""


class _Init(C.Structure):
    _fields_ = [("kmax_resol", C.c_int), ("kprintlev", C.c_int), ("prad", C.c_double), ("nproc", C.c_int),
                ("myproc", C.c_int), ("device", C.c_int), ("nprtrv", C.c_int)]


class _VSets(C.Structure):  # emi_vsets_t
    _fields_ = [("kvsetuv", C.POINTER(C.c_int)), ("nuv_g", C.c_int), ("kvsetsc", C.POINTER(C.c_int)), ("nsc_g", C.c_int),
                ("kvsetsc2", C.POINTER(C.c_int)), ("nsc2_g", C.c_int), ("kvsetsc3a", C.POINTER(C.c_int)), ("nsc3a_g", C.c_int),
                ("kvsetsc3b", C.POINTER(C.c_int)), ("nsc3b_g", C.c_int), ("nvar3a_g", C.c_int), ("nvar3b_g", C.c_int)]


class _LegpolIO(C.Structure):
    _fields_ = [("io", C.c_char_p), ("fname", C.c_char_p), ("ptr", C.c_void_p), ("len", C.c_size_t)]


class _Setup(C.Structure):
    _fields_ = [("ksmax", C.c_int), ("kdgl", C.c_int), ("kloen", C.POINTER(C.c_int)), ("kdlon", C.c_int),
                ("precision", C.c_int), ("lduseflt", C.c_int), ("ldll", C.c_int), ("ldstretch", C.c_int),
                ("lduserpnm", C.c_int)]


class _Ext(C.Structure):  # emi_extents_t
    _fields_ = [("sp_dim2", C.c_int), ("gp", C.c_int * 3), ("gpuv", C.c_int * 4), ("gp3a", C.c_int * 4),
                ("gp3b", C.c_int * 4), ("gp2", C.c_int * 3)]


class _Inv(C.Structure):
    _fields_ = [("mem_space", C.c_int), ("spvor", C.c_void_p), ("spdiv", C.c_void_p), ("nf_uv", C.c_int),
                ("spscalar", C.c_void_p), ("nf_scalar", C.c_int), ("spsc3a", C.c_void_p), ("sc3a_nlev", C.c_int),
                ("sc3a_nvar", C.c_int), ("spsc3b", C.c_void_p), ("sc3b_nlev", C.c_int), ("sc3b_nvar", C.c_int),
                ("spsc2", C.c_void_p), ("nf_sc2", C.c_int), ("ldscders", C.c_int), ("ldvorgp", C.c_int),
                ("lddivgp", C.c_int), ("lduvder", C.c_int), ("kproma", C.c_int), ("gp", C.c_void_p),
                ("gp_nfld", C.c_int), ("gpuv", C.c_void_p), ("gp3a", C.c_void_p), ("gp3b", C.c_void_p),
                ("gp2", C.c_void_p), ("stream", C.c_void_p), ("ext", C.POINTER(_Ext)), ("vsets", C.POINTER(_VSets))]


class _Dir(C.Structure):
    _fields_ = [("mem_space", C.c_int), ("spvor", C.c_void_p), ("spdiv", C.c_void_p), ("nf_uv", C.c_int),
                ("spscalar", C.c_void_p), ("nf_scalar", C.c_int), ("spsc3a", C.c_void_p), ("sc3a_nlev", C.c_int),
                ("sc3a_nvar", C.c_int), ("spsc3b", C.c_void_p), ("sc3b_nlev", C.c_int), ("sc3b_nvar", C.c_int),
                ("spsc2", C.c_void_p), ("nf_sc2", C.c_int), ("kproma", C.c_int), ("gp", C.c_void_p),
                ("gp_nfld", C.c_int), ("gpuv", C.c_void_p), ("gp3a", C.c_void_p), ("gp3b", C.c_void_p),
                ("gp2", C.c_void_p), ("stream", C.c_void_p), ("ext", C.POINTER(_Ext)), ("vsets", C.POINTER(_VSets))]


def build(force=False):
    """Compile libectrans_mi.so for gfx950 with hipcc (in-tree)."""
    import subprocess
    src = os.path.join(_HERE, "csrc", "ectrans_mi.hip")
    deps = [src] + [os.path.join(_HERE, "csrc", f) for f in ("emi_kernels.h", "emi_kernels_body.h", "emi_mr_body.h", "emi_mr_tables.h", "emi_types.h", "emi_rt.h", "emi_setup.h", "emi_stage.h")]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "ectrans_mi.h"))
    if not force and os.path.exists(_LIBPATH) and all(os.path.getmtime(_LIBPATH) >= os.path.getmtime(d) for d in deps):
        return _LIBPATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", _LIBPATH, src]
    subprocess.check_call(cmd)
    return _LIBPATH


def source_hash():
    """First 16 hex digits of the SHA-256 over the library sources (csrc/* and the C-ABI header): profiles/*_pmc_traffic.json
    is stamped with it, and bench.py quotes a counter file only for the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "ectrans_mi.h"))
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def _bind(L):
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.emi_init.argtypes = [C.POINTER(_Init)]
    L.emi_setup.argtypes = [C.POINTER(_Setup), ip]
    L.emi_setup_legpol.argtypes = [C.POINTER(_Setup), C.POINTER(_LegpolIO), ip]
    L.emi_inq_int.argtypes = [C.c_int, C.c_char_p, ip]
    L.emi_inq_int_array.argtypes = [C.c_int, C.c_char_p, ip, C.c_int]
    L.emi_inq_real_array.argtypes = [C.c_int, C.c_char_p, dp, C.c_int]
    L.emi_inq_legendre.argtypes = [C.c_int, C.c_int, C.c_int, dp, ip, ip]
    L.emi_inv_trans.argtypes = [C.c_int, C.POINTER(_Inv)]
    L.emi_dir_trans.argtypes = [C.c_int, C.POINTER(_Dir)]
    L.emi_specnorm.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, dp]
    L.emi_inv_transad.argtypes = [C.c_int, C.POINTER(_Inv)]
    L.emi_dir_transad.argtypes = [C.c_int, C.POINTER(_Dir)]
    L.emi_release.argtypes = [C.c_int]
    L.emi_finalize.argtypes = []
    L.emi_trim_cache.argtypes = []
    L.emi_last_error.restype = C.c_char_p
    L.emi_work_model.argtypes = [C.c_int, C.c_int, dp, dp, dp]
    L.emi_last_phase_ms.argtypes = [dp]
    L.emi_set_max_batch.argtypes = [C.c_int]
    L.emi_specnorm_partial.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, dp]
    L.emi_gpnorm.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, dp, dp, dp, C.c_int]
    L.emi_vordiv_to_uv.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    if hasattr(L, "emi_ptr_space"):  # (an older build loaded for an A/B run, tools/ab_libs.sh)
        L.emi_ptr_space.argtypes = [C.c_void_p]
        L.emi_wait.argtypes = [C.c_int]
    L.emi_specnorm_kvset.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, dp]
    L.emi_set_alltoallv.argtypes = [C.c_void_p, C.c_void_p]
    L.emi_set_profile.argtypes = [C.c_int]
    L.emi_last_phase_launches.argtypes = [ip]
    if hasattr(L, "emi_last_exchange"):  # (an older build loaded for an A/B run)
        L.emi_last_exchange.argtypes = [dp, ip, dp]
        L.emi_last_fft_launches.argtypes = [C.POINTER(C.c_longlong)]
    L.emi_crc64.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_ulonglong)]
    L.emi_set_host_collectives.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.emi_dist_spec.argtypes = [C.c_int, C.c_void_p, C.c_int, ip, ip, C.c_void_p]
    L.emi_gath_spec.argtypes = [C.c_int, C.c_void_p, C.c_int, ip, C.c_void_p]
    L.emi_dist_grid.argtypes = [C.c_int, C.c_void_p, C.c_int, ip, ip, C.c_int, C.c_void_p]
    L.emi_gath_grid.argtypes = [C.c_int, C.c_void_p, C.c_int, ip, C.c_int, C.c_void_p]
    L.emi_inq_tasks.argtypes = [ip, ip]
    L.emi_inq_init.argtypes = [ip, dp]
    L.emi_set_nprtrv.argtypes = [C.c_int]
    return L


def lib():
    """The HIP shared library.  No fallback: a missing library is an error."""
    global _L
    if _L is None:
        if not os.path.exists(_LIBPATH):
            raise TransError("libectrans_mi.so is not built (run __graft_entry__.build()); "
                             "ectrans_amd has no CPU implementation")
        _L = _bind(C.CDLL(_LIBPATH))
    return _L


def _use_library_for_tests(path):
    """tests/emu only: point the wrapper at the CPU functional emulator build."""
    global _L
    _L = _bind(C.CDLL(path))
    return _L


def _chk(rc):
    if rc != 0:
        raise TransError(lib().emi_last_error().decode())


def _is_torch(a):
    return a is not None and type(a).__module__.startswith("torch")


def _ptr(a, space):
    """(pointer, keep-alive object) of an array living in `space`."""
    if a is None:
        return None, None
    if _is_torch(a):
        if not a.is_contiguous() or str(a.dtype) != "torch." + space[1]:
            raise TransError("device arrays must be contiguous %s tensors" % space[1])
        if space[0] is None:
            space[0] = EMI_MEM_DEVICE if a.is_cuda else EMI_MEM_HOST
        elif space[0] != (EMI_MEM_DEVICE if a.is_cuda else EMI_MEM_HOST):
            raise TransError("all arrays of one call must live in the same memory space")
        return a.data_ptr(), a
    if not (isinstance(a, np.ndarray) and a.dtype == np.dtype(space[1]) and a.flags.c_contiguous):
        raise TransError("host arrays must be C-contiguous %s numpy arrays" % space[1])
    if space[0] is None:
        space[0] = EMI_MEM_HOST
    elif space[0] != EMI_MEM_HOST:
        raise TransError("all arrays of one call must live in the same memory space")
    return a.ctypes.data, a


_DIST = {"nproc": 1, "nprtrv": 1, "group": None, "device": None}


def setup_trans0(kmax_resol=1, kprintlev=0, prad=None, device=-1, kprtrw=1, myproc=1, group=None, alltoallv=None,
                 transport="torch", kprtrv=1, **unsupported):
    """SETUP_TRANS0 (setup_trans0.h:12-89).

    kprtrw > 1: this process is task `myproc` (1-based) of a W-set of `kprtrw` tasks, one per GPU
    (NPRTRV = NPRGPEW = 1); the all-to-all-v between them defaults to torch.distributed on `group`
    (RCCL for CUDA devices, gloo on the CPU test tier) -- see ectrans_amd.dist.
    kprtrv > 1: NPRTRV V-sets (sump_trans0_mod.F90:49): kprtrw x kprtrv tasks, task `myproc` is (MYSETW, MYSETV) =
    ((myproc - 1) // kprtrv + 1, (myproc - 1) % kprtrv + 1); fields are dealt to the V-sets by the kvset* arguments of the
    transforms.
    transport="rccl": the native transport of ectrans_amd/rccl instead (what a Fortran host attaches: grouped
    ncclSend / ncclRecv on the library's stream, no Python callback per field batch); `group` only carries the
    128-byte unique id and the small reductions of SPECNORM."""
    for k, v in unsupported.items():
        if k.lower() in ("kprgpns", "kprgpew") and v not in (None, 1, kprtrw, kprtrw * kprtrv):
            raise TransError("SETUP_TRANS0: %s=%r: only the W-set decomposition (KPRTRW tasks) is supported" % (k, v))
    nproc_all = kprtrw * max(1, kprtrv)
    if transport == "rccl":
        from . import dist as _dist
        if kprtrv > 1:
            _chk(lib().emi_set_nprtrv(int(kprtrv)))
        _dist.rccl_native_attach(nproc_all, myproc, kmax_resol, kprintlev, prad, device, group)  # calls emi_init itself
        _DIST.update(nproc=nproc_all, nprtrv=max(1, kprtrv), group=group if nproc_all > 1 else None,
                     device=("cuda:%d" % device) if nproc_all > 1 and device is not None and device >= 0 else None)
        return
    if nproc_all > 1:
        from . import dist as _dist
        dev = ("cuda:%d" % device) if device is not None and device >= 0 else "cpu"
        hook = alltoallv if alltoallv is not None else _dist.make_alltoallv_hook(group, dev, p2p=kprtrv > 1)
        _chk(lib().emi_set_alltoallv(C.cast(hook, C.c_void_p), None))
        bc, ag = _dist.make_host_collectives(group, dev)  # DIST_x / GATH_x, SPECNORM over several tasks
        _chk(lib().emi_set_host_collectives(C.cast(bc, C.c_void_p), C.cast(ag, C.c_void_p), None))
        _DIST.update(nproc=nproc_all, nprtrv=max(1, kprtrv), group=group, device=dev)
    else:
        _DIST.update(nproc=1, nprtrv=1, group=None, device=None)
    cfg = _Init(kmax_resol, kprintlev, prad if prad else 0.0, nproc_all, myproc, device if device is not None else -1, max(1, kprtrv))
    _chk(lib().emi_init(C.byref(cfg)))


def inq_init():
    """(KMAX_RESOL, PRAD) the library was initialised with (emi_inq_init): what a host that adopts an attached transport's
    initialisation compares its own SETUP_TRANS0 arguments with."""
    k, r = C.c_int(0), C.c_double(0.0)
    _chk(lib().emi_inq_init(C.byref(k), C.byref(r)))
    return k.value, r.value


_PREC = {}  # kresol -> array dtype name of that resolution


def setup_trans(ksmax, kdgl, kloen=None, kdlon=0, lduseflt=False, ldll=False, pstret=None, precision=8,
                cdio_legpol=None, cdlegpolfname=None, klegpolptr=None, klegpolptr_len=None, lduserpnm=False):
    """SETUP_TRANS (setup_trans.h:12-115); returns KRESOL.

    cdio_legpol: "writef" writes the Legendre polynomials of this setup to `cdlegpolfname`, "readf" takes
    them from that file, "membuf" from the file image `klegpolptr` (bytes-like or numpy array, or an
    address with `klegpolptr_len`) -- the reference's format (write_legpol_mod.F90), one task only.

    precision: 8 = the reference's double-precision library (libtrans_dp, JPRB=JPRD), arrays are
    float64; 4 = its single-precision library (libtrans_sp, JPRB=JPRM), arrays are float32 (setup --
    Gaussian latitudes, Legendre recurrences -- still runs in double, as in the reference)."""
    if precision not in (4, 8):
        raise TransError("SETUP_TRANS: precision must be 4 or 8")
    cfg = _Setup()
    cfg.ksmax, cfg.kdgl, cfg.kdlon, cfg.precision = int(ksmax), int(kdgl), int(kdlon), int(precision)
    keep = None
    if kloen is not None:
        keep = np.ascontiguousarray(kloen, dtype=np.int32)
        if keep.size < kdgl:
            raise TransError("SETUP_TRANS: KLOEN TOO SHORT")
        cfg.kloen = keep.ctypes.data_as(C.POINTER(C.c_int))
    cfg.lduseflt, cfg.ldll = int(bool(lduseflt)), int(bool(ldll))
    cfg.lduserpnm = int(bool(lduserpnm))  # True: Belousov's generator (the Fortran API's default); False: SUPOLF, as the benchmark
    cfg.ldstretch = int(pstret is not None and abs(pstret - 1.0) > 100 * np.finfo(float).eps)
    kresol = C.c_int(0)
    if cdio_legpol is None:
        _chk(lib().emi_setup(C.byref(cfg), C.byref(kresol)))
    else:
        io = _LegpolIO(str(cdio_legpol).encode(), None if cdlegpolfname is None else str(cdlegpolfname).encode(), None, 0)
        seg = None
        if klegpolptr is not None:
            if isinstance(klegpolptr, int):
                io.ptr, io.len = klegpolptr, int(klegpolptr_len or 0)
            else:
                seg = np.frombuffer(klegpolptr, dtype=np.uint8) if not isinstance(klegpolptr, np.ndarray) \
                    else np.ascontiguousarray(klegpolptr).view(np.uint8).reshape(-1)
                io.ptr, io.len = seg.ctypes.data, seg.size if klegpolptr_len is None else int(klegpolptr_len)
        _chk(lib().emi_setup_legpol(C.byref(cfg), C.byref(io), C.byref(kresol)))
        del seg  # the library has copied what it needs
    _PREC[kresol.value] = "float32" if precision == 4 else "float64"
    return kresol.value


def real_dtype(kresol):
    """numpy dtype name of the arrays of resolution `kresol` ("float64" or "float32")."""
    return _PREC.get(kresol, "float64")


_INT_SCALARS = ("nspec2", "nspec2g", "nspec2mx", "nspec", "nspecg", "ngptot", "ngptotg", "ngptotmx", "nump", "ndgl",
                "nsmax", "ndlon", "nproc", "myproc", "nfrstlat", "nlstlat", "nprtrw", "nprtrv", "mysetw", "mysetv", "ngptot_band")
_INT_ARRAYS = {"nloen": "ndgl", "nmen": "ndgl", "ndglu": "nsmax+1", "nasm0": "nsmax+1", "myms": "nump",
               "procm": "nsmax+1", "latlo": "nproc+1", "fftwork": "ndgl"}
_REAL_ARRAYS = {"rmu": "ndgl", "pmu": "ndgl", "rgw": "ndgl", "pgw": "ndgl", "racthe": "ndgl"}


def trans_inq(kresol, name):
    """TRANS_INQ (trans_inq.h): one quantity by (lower-case) name."""
    L = lib()
    name = name.lower()
    if name in _INT_SCALARS:
        v = C.c_int(0)
        _chk(L.emi_inq_int(kresol, name.encode(), C.byref(v)))
        return v.value
    dims = {"ndgl": trans_inq(kresol, "ndgl"), "nsmax+1": trans_inq(kresol, "nsmax") + 1,
            "nump": trans_inq(kresol, "nump"), "nproc+1": trans_inq(kresol, "nproc") + 1}
    if name in _INT_ARRAYS:
        out = np.zeros(dims[_INT_ARRAYS[name]], dtype=np.int32)
        _chk(L.emi_inq_int_array(kresol, name.encode(), out.ctypes.data_as(C.POINTER(C.c_int)), out.size))
        return out
    if name in _REAL_ARRAYS:
        out = np.zeros(dims[_REAL_ARRAYS[name]])
        _chk(L.emi_inq_real_array(kresol, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
        return out
    raise TransError("TRANS_INQ: unknown quantity %r" % name)


def legendre_panel(kresol, m, symmetric):
    """S%FA(m)%RPNMS / RPNMA as the reference lays them out; numpy [col(n desc)][lat]."""
    L = lib()
    r, c = C.c_int(), C.c_int()
    _chk(L.emi_inq_legendre(kresol, m, int(symmetric), None, C.byref(r), C.byref(c)))
    out = np.zeros((c.value, r.value))
    if out.size:
        _chk(L.emi_inq_legendre(kresol, m, int(symmetric), out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(r),
                                C.byref(c)))
    return out


def _fill_spec(a, space, keep, pspvor, pspdiv, pspscalar, pspsc3a, pspsc3b, pspsc2, nspec2):
    def chk2(x, nm):
        if x is not None and (x.ndim != 2 or x.shape[0] != nspec2):
            raise TransError("%s must have shape (nspec2=%d, nfld)" % (nm, nspec2))

    chk2(pspvor, "PSPVOR"), chk2(pspdiv, "PSPDIV"), chk2(pspscalar, "PSPSCALAR"), chk2(pspsc2, "PSPSC2")
    for x, nm in ((pspsc3a, "PSPSC3A"), (pspsc3b, "PSPSC3B")):
        if x is not None and (x.ndim != 3 or x.shape[1] != nspec2):
            raise TransError("%s must have shape (nvar, nspec2=%d, nlev)" % (nm, nspec2))
    if (pspvor is None) != (pspdiv is None):
        raise TransError("PSPVOR and PSPDIV must be given together")
    if pspvor is not None and pspvor.shape != pspdiv.shape:
        raise TransError("PSPVOR and PSPDIV shapes differ")
    for nm, x in (("spvor", pspvor), ("spdiv", pspdiv), ("spscalar", pspscalar), ("spsc3a", pspsc3a),
                  ("spsc3b", pspsc3b), ("spsc2", pspsc2)):
        p, k = _ptr(x, space)
        setattr(a, nm, p)
        keep.append(k)
    a.nf_uv = 0 if pspvor is None else pspvor.shape[1]
    a.nf_scalar = 0 if pspscalar is None else pspscalar.shape[1]
    a.nf_sc2 = 0 if pspsc2 is None else pspsc2.shape[1]
    a.sc3a_nvar, a.sc3a_nlev = (0, 0) if pspsc3a is None else (pspsc3a.shape[0], pspsc3a.shape[2])
    a.sc3b_nvar, a.sc3b_nlev = (0, 0) if pspsc3b is None else (pspsc3b.shape[0], pspsc3b.shape[2])


def _fill_grid(a, space, keep, pgp, pgpuv, pgp3a, pgp3b, pgp2, nproma, ngpblks, specs=()):
    """Grid arrays + the extents block (emi_extents_t): the library makes the reference's extent checks
    (inv_trans.F90:476-600) on the real shapes.  numpy/torch shapes are the Fortran extents reversed."""
    ext = _Ext()
    for nm, x, rank in (("gp", pgp, 3), ("gpuv", pgpuv, 4), ("gp3a", pgp3a, 4), ("gp3b", pgp3b, 4), ("gp2", pgp2, 3)):
        if x is not None:
            if x.ndim != rank:
                raise TransError("P%s must have %d dimensions, got shape %s" % (nm.upper(), rank, tuple(x.shape)))
            for i, n in enumerate(reversed(tuple(x.shape))):
                getattr(ext, nm)[i] = int(n)
        p, k = _ptr(x, space)
        setattr(a, nm, p)
        keep.append(k)
    sp2 = [x.shape[-2] if x.ndim == 3 else x.shape[0] for x in specs if x is not None]
    ext.sp_dim2 = int(min(sp2)) if sp2 else 0
    a.gp_nfld = 0 if pgp is None else pgp.shape[1]
    a.ext = C.pointer(ext)
    keep.append(ext)


def _fill_vsets(a, keep, kvsetuv, kvsetsc, kvsetsc2, kvsetsc3a, kvsetsc3b, pgp3a=None, pgp3b=None, dmul=1):
    """KVSETUV / KVSETSC / KVSETSC2 / KVSETSC3A / KVSETSC3B (inv_trans.h:84-101): the V-set (1..NPRTRV) of every GLOBAL field;
    with them the spectral arrays hold this task's V-set only, the grid arrays all fields."""
    if all(k is None for k in (kvsetuv, kvsetsc, kvsetsc2, kvsetsc3a, kvsetsc3b)):
        return
    vs = _VSets()
    for nm, cnt, k in (("kvsetuv", "nuv_g", kvsetuv), ("kvsetsc", "nsc_g", kvsetsc), ("kvsetsc2", "nsc2_g", kvsetsc2),
                       ("kvsetsc3a", "nsc3a_g", kvsetsc3a), ("kvsetsc3b", "nsc3b_g", kvsetsc3b)):
        if k is not None:
            arr = np.ascontiguousarray(k, dtype=np.int32)
            keep.append(arr)
            setattr(vs, nm, arr.ctypes.data_as(C.POINTER(C.c_int)))
            setattr(vs, cnt, int(arr.size))
    # the variable count of the 3-D arrays from the GRID side (every task holds all fields there), so that a task whose V-set owns
    # no level -- and passes no PSPSC3A -- still lists the same global fields as its peers (the reference: UBOUND(PSPSC3A,3))
    vs.nvar3a_g = 0 if pgp3a is None or pgp3a.ndim != 4 else int(pgp3a.shape[1]) // dmul
    vs.nvar3b_g = 0 if pgp3b is None or pgp3b.ndim != 4 else int(pgp3b.shape[1]) // dmul
    keep.append(vs)
    a.vsets = C.pointer(vs)
    a.ext = None  # the extents block describes one-V-set calls (local == global field counts)


def inv_trans(kresol, pspvor=None, pspdiv=None, pspscalar=None, pspsc3a=None, pspsc3b=None, pspsc2=None,
              ldscders=False, ldvorgp=False, lddivgp=False, lduvder=False, kproma=None, pgp=None, pgpuv=None,
              pgp3a=None, pgp3b=None, pgp2=None, stream=None, kvsetuv=None, kvsetsc=None, kvsetsc2=None, kvsetsc3a=None,
              kvsetsc3b=None):
    """INV_TRANS (inv_trans.h:12-163): spectral -> grid point, results written into pgp*/..."""
    a, space, keep = _Inv(), [None, real_dtype(kresol)], []
    nspec2, ngptot = trans_inq(kresol, "nspec2"), trans_inq(kresol, "ngptot")
    nproma = int(kproma) if kproma else ngptot
    ngpblks = (ngptot - 1) // nproma + 1
    _fill_spec(a, space, keep, pspvor, pspdiv, pspscalar, pspsc3a, pspsc3b, pspsc2, nspec2)
    _fill_grid(a, space, keep, pgp, pgpuv, pgp3a, pgp3b, pgp2, nproma, ngpblks,
               (pspvor, pspdiv, pspscalar, pspsc3a, pspsc3b, pspsc2))
    a.ldscders, a.ldvorgp, a.lddivgp, a.lduvder = int(ldscders), int(ldvorgp), int(lddivgp), int(lduvder)
    a.kproma = nproma
    a.mem_space = space[0] if space[0] is not None else EMI_MEM_HOST
    a.stream = stream
    _fill_vsets(a, keep, kvsetuv, kvsetsc, kvsetsc2, kvsetsc3a, kvsetsc3b, pgp3a, pgp3b, 3 if ldscders else 1)
    _chk(lib().emi_inv_trans(kresol, C.byref(a)))


def dir_trans(kresol, pspvor=None, pspdiv=None, pspscalar=None, pspsc3a=None, pspsc3b=None, pspsc2=None,
              kproma=None, pgp=None, pgpuv=None, pgp3a=None, pgp3b=None, pgp2=None, stream=None, kvsetuv=None, kvsetsc=None,
              kvsetsc2=None, kvsetsc3a=None, kvsetsc3b=None):
    """DIR_TRANS (dir_trans.h:12-140): grid point -> spectral, results written into psp*."""
    a, space, keep = _Dir(), [None, real_dtype(kresol)], []
    nspec2, ngptot = trans_inq(kresol, "nspec2"), trans_inq(kresol, "ngptot")
    nproma = int(kproma) if kproma else ngptot
    ngpblks = (ngptot - 1) // nproma + 1
    _fill_spec(a, space, keep, pspvor, pspdiv, pspscalar, pspsc3a, pspsc3b, pspsc2, nspec2)
    _fill_grid(a, space, keep, pgp, pgpuv, pgp3a, pgp3b, pgp2, nproma, ngpblks,
               (pspvor, pspdiv, pspscalar, pspsc3a, pspsc3b, pspsc2))
    a.kproma = nproma
    a.mem_space = space[0] if space[0] is not None else EMI_MEM_HOST
    a.stream = stream
    _fill_vsets(a, keep, kvsetuv, kvsetsc, kvsetsc2, kvsetsc3a, kvsetsc3b, pgp3a, pgp3b)
    _chk(lib().emi_dir_trans(kresol, C.byref(a)))


def inv_transad(kresol, pspvor=None, pspdiv=None, pspscalar=None, pspsc3a=None, pspsc3b=None, pspsc2=None,
                ldscders=False, ldvorgp=False, lddivgp=False, lduvder=False,
                kproma=None, pgp=None, pgpuv=None, pgp3a=None, pgp3b=None, pgp2=None, stream=None, kvsetuv=None, kvsetsc=None,
                kvsetsc2=None, kvsetsc3a=None, kvsetsc3b=None):
    """INV_TRANSAD (inv_transad.h:12): adjoint of INV_TRANS -- reads pgp*, writes psp* (overwritten).
    Inner products: plain sum in grid-point space, SPECNORM weights (1 for m = 0, 2 for m > 0) in
    spectral space, as tests/trans/test_invtrans_adjoint.F90:243-315.  With ldscders / ldvorgp / lddivgp / lduvder the
    grid arrays carry the derivative / vorticity / divergence inputs in INV_TRANS's layout (inv_trans.h:66-76)."""
    a, space, keep = _Inv(), [None, real_dtype(kresol)], []
    nspec2, ngptot = trans_inq(kresol, "nspec2"), trans_inq(kresol, "ngptot")
    nproma = int(kproma) if kproma else ngptot
    _fill_spec(a, space, keep, pspvor, pspdiv, pspscalar, pspsc3a, pspsc3b, pspsc2, nspec2)
    _fill_grid(a, space, keep, pgp, pgpuv, pgp3a, pgp3b, pgp2, nproma, (ngptot - 1) // nproma + 1,
               (pspvor, pspdiv, pspscalar, pspsc3a, pspsc3b, pspsc2))
    a.ldscders, a.ldvorgp, a.lddivgp, a.lduvder = int(ldscders), int(ldvorgp), int(lddivgp), int(lduvder)
    a.kproma = nproma
    a.mem_space = space[0] if space[0] is not None else EMI_MEM_HOST
    a.stream = stream
    _fill_vsets(a, keep, kvsetuv, kvsetsc, kvsetsc2, kvsetsc3a, kvsetsc3b, pgp3a, pgp3b, 3 if ldscders else 1)
    _chk(lib().emi_inv_transad(kresol, C.byref(a)))


def dir_transad(kresol, pspvor=None, pspdiv=None, pspscalar=None, pspsc3a=None, pspsc3b=None, pspsc2=None,
                kproma=None, pgp=None, pgpuv=None, pgp3a=None, pgp3b=None, pgp2=None, stream=None, kvsetuv=None, kvsetsc=None,
                kvsetsc2=None, kvsetsc3a=None, kvsetsc3b=None):
    """DIR_TRANSAD (dir_transad.h:12): adjoint of DIR_TRANS -- reads psp*, writes pgp*."""
    a, space, keep = _Dir(), [None, real_dtype(kresol)], []
    nspec2, ngptot = trans_inq(kresol, "nspec2"), trans_inq(kresol, "ngptot")
    nproma = int(kproma) if kproma else ngptot
    _fill_spec(a, space, keep, pspvor, pspdiv, pspscalar, pspsc3a, pspsc3b, pspsc2, nspec2)
    _fill_grid(a, space, keep, pgp, pgpuv, pgp3a, pgp3b, pgp2, nproma, (ngptot - 1) // nproma + 1,
               (pspvor, pspdiv, pspscalar, pspsc3a, pspsc3b, pspsc2))
    a.kproma = nproma
    a.mem_space = space[0] if space[0] is not None else EMI_MEM_HOST
    a.stream = stream
    _fill_vsets(a, keep, kvsetuv, kvsetsc, kvsetsc2, kvsetsc3a, kvsetsc3b, pgp3a, pgp3b)
    _chk(lib().emi_dir_transad(kresol, C.byref(a)))


def specnorm(kresol, pspec, kvset=None):
    """SPECNORM (specnorm.h:12): per-field spectral L2 norm, returned as a numpy array (on every
    task; the reference returns it on the master only).  kvset (NPRTRV > 1, specnorm.F90:82-101): V-set of every GLOBAL
    field; pspec holds this task's fields of its own V-set and the norms of all len(kvset) fields come back."""
    space = [None, real_dtype(kresol)]
    if pspec.shape[1] == 0:
        p, keep, space[0] = None, None, EMI_MEM_HOST
    else:
        p, keep = _ptr(pspec, space)
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    if kvset is not None and _DIST.get("nprtrv", 1) > 1:
        kv = np.ascontiguousarray(kvset, dtype=np.int32)
        out = np.zeros(kv.size)
        _chk(lib().emi_specnorm_kvset(kresol, space[0], p, pspec.shape[1], kv.ctypes.data_as(C.POINTER(C.c_int)), kv.size, pd(out)))
        return out
    out = np.zeros(pspec.shape[1])
    if _DIST["nproc"] == 1 or _DIST.get("nprtrv", 1) > 1:
        # several V-sets: the library sums over the tasks of this V-set itself (host collectives)
        _chk(lib().emi_specnorm(kresol, space[0], p, pspec.shape[1], pd(out)))
        return out
    from . import dist as _dist
    _chk(lib().emi_specnorm_partial(kresol, space[0], p, pspec.shape[1], pd(out)))
    return np.sqrt(_dist.all_reduce_sum(out, _DIST["group"], _DIST["device"]))  # every task gets the norms


def gpnorm_trans(kresol, pgp, kfields=None, kproma=None, ldave_only=False, pmin=None, pmax=None):
    """GPNORM_TRANS (gpnorm_trans.h:12): (PAVE, PMIN, PMAX) of the first `kfields` fields of pgp[ngpblks, nfld, nproma] -- the
    area-weighted average over the sphere (Gaussian weights), minimum and maximum -- as numpy arrays, on every task (the
    reference: task 1).  ldave_only: pmin / pmax are the caller's local extrema and are only reduced over the tasks."""
    space = [None, real_dtype(kresol)]
    p, keep = _ptr(pgp, space)
    if pgp.ndim != 3:
        raise TransError("PGP must have 3 dimensions (ngpblks, fields, nproma), got shape %s" % (tuple(pgp.shape),))
    nf = int(pgp.shape[1]) if kfields is None else int(kfields)
    nproma = int(kproma) if kproma else int(pgp.shape[2])
    ave, mn, mx = np.zeros(nf), np.zeros(nf), np.zeros(nf)
    if ldave_only:
        mn[:], mx[:] = np.asarray(pmin, dtype=np.float64)[:nf], np.asarray(pmax, dtype=np.float64)[:nf]
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    _chk(lib().emi_gpnorm(kresol, space[0], p, int(pgp.shape[1]), nf, nproma, pd(ave), pd(mn), pd(mx), int(bool(ldave_only))))
    return ave, mn, mx


def vordiv_to_uv(pspvor, pspdiv, ksmax, pspu=None, pspv=None):
    """VORDIV_TO_UV (vordiv_to_uv.h:12): spectral vorticity / divergence [nspec2, nfld] -> spectral (U, V) = (u, v) cos(theta), total
    wavenumbers n <= ksmax, for the zonal wavenumbers of this task.  Needs setup_trans0 only; returns (pspu, pspv)."""
    is64 = str(pspvor.dtype).endswith("float64")
    space = [None, "float64" if is64 else "float32"]
    pv_, k1 = _ptr(pspvor, space)
    pd_, k2 = _ptr(pspdiv, space)
    if pspu is None:
        pspu = pspvor.clone() if _is_torch(pspvor) else np.zeros_like(pspvor)
    if pspv is None:
        pspv = pspvor.clone() if _is_torch(pspvor) else np.zeros_like(pspvor)
    pu_, k3 = _ptr(pspu, space)
    pw_, k4 = _ptr(pspv, space)
    if not (tuple(pspvor.shape) == tuple(pspdiv.shape) == tuple(pspu.shape) == tuple(pspv.shape)) or pspvor.ndim != 2:
        raise TransError("VORDIV_TO_UV: PSPVOR, PSPDIV, PSPU, PSPV must be [nspec2, nfld] arrays of one shape")
    _chk(lib().emi_vordiv_to_uv(int(ksmax), 8 if is64 else 4, space[0], pv_, pd_, pu_, pw_, int(pspvor.shape[1]), int(pspvor.shape[0])))
    return pspu, pspv


# ---------------------------------------------------------------------------------------------
# DIST_SPEC / GATH_SPEC / DIST_GRID / GATH_GRID: global <-> distributed arrays.
# Not on the transform hot path: host (numpy) arrays, moved with torch.distributed collectives.
# Global spectral order (dist_spec_control_mod.F90:158-161, IASM0G): m = 0..N, n = m..N, (re, im).
# Global grid order: latitudes north to south = the tasks' latitude bands in task order.
# ---------------------------------------------------------------------------------------------
def _np(a):
    return a.detach().cpu().numpy() if _is_torch(a) else np.asarray(a)


def _global_spec_index(kresol):
    """indices into the global spectral array of this task's coefficients, in local order"""
    n, myms = trans_inq(kresol, "nsmax"), trans_inq(kresol, "myms")
    iasm0g = np.concatenate([[0], np.cumsum([2 * (n - m + 1) for m in range(n + 1)])])
    return np.concatenate([np.arange(iasm0g[m], iasm0g[m] + 2 * (n - m + 1)) for m in myms]) if len(myms) else \
        np.zeros(0, dtype=np.int64)


def _roots(k, nfld, what):
    k = np.ascontiguousarray(np.broadcast_to(np.asarray(k, dtype=np.int32), (nfld,))) if np.ndim(k) <= 1 else None
    if k is None or (nfld and (k.min() < 1 or k.max() > _DIST["nproc"])):
        raise TransError("%s: task numbers must be 1..%d" % (what, _DIST["nproc"]))
    return k


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def dist_spec(kresol, pspecg, kfdistg, kfrom=1):
    """DIST_SPEC (dist_spec.h:11) over emi_dist_spec: global spectral fields `pspecg` (nspec2g, kfdistg), field f held
    by task kfrom[f] (1-based; the columns of other tasks' fields are not read, the array may be None on a task that is
    the source of none), -> this task's (nspec2, kfdistg) local array."""
    dt = real_dtype(kresol)
    ns2g, ns2, me = trans_inq(kresol, "nspec2g"), trans_inq(kresol, "nspec2"), trans_inq(kresol, "myproc")
    kfrom = _roots(kfrom, kfdistg, "DIST_SPEC:KFROM")
    mine = np.flatnonzero(kfrom == me)
    g = None
    if mine.size:
        a = _np(pspecg)
        if a.shape[0] != ns2g:
            raise TransError("DIST_SPEC: PSPECG must have shape (nspec2g=%d, nfld)" % ns2g)
        g = np.ascontiguousarray(a[:, mine].T, dtype=dt)  # [n_mine][nspec2g]
    out = np.zeros((ns2, kfdistg), dtype=dt)
    _chk(lib().emi_dist_spec(kresol, None if g is None else g.ctypes.data, kfdistg, _iptr(kfrom), None, out.ctypes.data))
    return out


def gath_spec(kresol, pspec, kfgathg, kto=1):
    """GATH_SPEC (gath_spec.h:11) over emi_gath_spec: local (nspec2, kfgathg) -> global (nspec2g, n_mine) holding the
    fields this task is the target of (None if it is the target of none)."""
    dt = real_dtype(kresol)
    ns2g, me = trans_inq(kresol, "nspec2g"), trans_inq(kresol, "myproc")
    kto = _roots(kto, kfgathg, "GATH_SPEC:KTO")
    loc = np.ascontiguousarray(_np(pspec), dtype=dt)
    nm = int((kto == me).sum())
    g = np.zeros((nm, ns2g), dtype=dt) if nm else None
    _chk(lib().emi_gath_spec(kresol, None if g is None else g.ctypes.data, kfgathg, _iptr(kto), loc.ctypes.data))
    return None if g is None else np.ascontiguousarray(g.T)


def gath_grid(kresol, pgp, kfgathg, kto=1):
    """GATH_GRID (gath_grid.h:11) over emi_gath_grid: local blocked grid array (ngpblks, kfgathg, nproma) -> global
    (n_mine, ngptotg) of the fields this task is the target of (None if none)."""
    dt = real_dtype(kresol)
    ngg, me = trans_inq(kresol, "ngptotg"), trans_inq(kresol, "myproc")
    kto = _roots(kto, kfgathg, "GATH_GRID:KTO")
    loc = np.ascontiguousarray(_np(pgp), dtype=dt)
    if loc.ndim != 3 or loc.shape[1] != kfgathg:
        raise TransError("GATH_GRID: PGP must have shape (ngpblks, kfgathg=%d, nproma)" % kfgathg)
    nm = int((kto == me).sum())
    g = np.zeros((nm, ngg), dtype=dt) if nm else None
    _chk(lib().emi_gath_grid(kresol, None if g is None else g.ctypes.data, kfgathg, _iptr(kto), loc.shape[2], loc.ctypes.data))
    return g


def dist_grid(kresol, pgpg, kfdistg, kfrom=1, kproma=None):
    """DIST_GRID (dist_grid.h:11) over emi_dist_grid: global (kfdistg, ngptotg) on task kfrom[f] -> this task's blocked
    (ngpblks, kfdistg, nproma) array (padding of the last block zero)."""
    dt = real_dtype(kresol)
    ngl, ngg, me = trans_inq(kresol, "ngptot"), trans_inq(kresol, "ngptotg"), trans_inq(kresol, "myproc")
    kfrom = _roots(kfrom, kfdistg, "DIST_GRID:KFROM")
    mine = np.flatnonzero(kfrom == me)
    g = None
    if mine.size:
        a = _np(pgpg)
        if a.shape[-1] != ngg:
            raise TransError("DIST_GRID: PGPG must have shape (nfld, ngptotg=%d)" % ngg)
        g = np.ascontiguousarray(a[mine], dtype=dt)
    nproma = int(kproma) if kproma else ngl
    out = np.zeros(((ngl - 1) // nproma + 1, kfdistg, nproma), dtype=dt)
    _chk(lib().emi_dist_grid(kresol, None if g is None else g.ctypes.data, kfdistg, _iptr(kfrom), None, nproma, out.ctypes.data))
    return out


def trans_release(kresol):
    _chk(lib().emi_release(kresol))


def trim_cache():
    """Frees the idle device staging buffers of host-array calls (emi_trim_cache); TRANS_RELEASE and TRANS_END do it too."""
    _chk(lib().emi_trim_cache())


def trans_end():
    _chk(lib().emi_finalize())
    _DIST.update(nproc=1, nprtrv=1, group=None, device=None)


def work_model(kresol, nfields):
    """Algorithmic work per direction for `nfields` Fourier-space fields."""
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    _chk(lib().emi_work_model(kresol, nfields, C.byref(a), C.byref(b), C.byref(c)))
    return {"legendre_flops": a.value, "fft_flops": b.value, "fourier_bytes": c.value}


def last_phase_ms():
    """Device time (ms) of the last inv_trans/dir_trans call per phase
    [spectral pack/unpack, Legendre MFMA kernel, FFT kernels] -- needs set_profile(True)."""
    out = (C.c_double * 3)()
    lib().emi_last_phase_ms(out)
    return list(out)


def last_phase_launches():
    out = (C.c_int * 3)()
    lib().emi_last_phase_launches(out)
    return list(out)


def last_exchange():
    """(ms, calls, bytes_sent) of the all-to-all-v exchanges since set_profile(): HIP events around the hook on the exchange stream."""
    ms, n, b = C.c_double(), C.c_int(), C.c_double()
    lib().emi_last_exchange(C.byref(ms), C.byref(n), C.byref(b))
    return ms.value, n.value, b.value


def last_fft_launches():
    """kernel launches of the FFT phases since set_profile()"""
    k = C.c_longlong()
    lib().emi_last_fft_launches(C.byref(k))
    return k.value


def set_profile(on):
    """True / 1: HIP-event phase timers per call (last_phase_ms() of the last call); 2: accumulated over all calls
    since this set_profile(2) -- nothing is resolved, so nothing synchronises, between the calls of a timed loop."""
    lib().emi_set_profile(int(on))


def set_max_batch(n):
    lib().emi_set_max_batch(int(n))
