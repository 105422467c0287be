"""Multi-GPU (one task per GPU) support: the all-to-all-v hook of the C-ABI over torch.distributed.

The library shards one transform over tasks the way the reference does over its W-sets: zonal
wavenumbers zig-zag over tasks (suwavedi_mod.F90:118-137), latitudes in contiguous bands, and ONE
all-to-all-v per direction (TRLTOM/TRMTOL, trltom_mod.F90:96-136, trmtol_mod.F90:101-141) of whole
blocks of the device-resident Fourier buffer.  Here that exchange is RCCL (`backend="nccl"` on
ROCm) over xGMI; on the CPU test tier the same hook runs over gloo against the emulator build.

    import torch.distributed as dist, ectrans_amd as et
    dist.init_process_group("nccl", ...)
    et.setup_trans0(kprtrw=dist.get_world_size(), myproc=dist.get_rank() + 1, device=local_rank)
    r = et.setup_trans(nsmax, ndgl, nloen)      # local sizes: trans_inq(r, "nspec2"/"ngptot"/"myms"/...)
"""
import ctypes as C
import sys
import traceback

import numpy as np

_A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_void_p,
                   C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int, C.c_void_p)
_keep = []


class _DeviceBuffer:
    """Zero-copy view of raw device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nelem):
        self.__cuda_array_interface__ = {"shape": (nelem,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def _as_tensor(ptr, nbytes, device):
    import torch
    n = int(nbytes) // 8
    if n == 0:
        return torch.empty(0, dtype=torch.float64, device=device)
    if device.type == "cuda":
        return torch.as_tensor(_DeviceBuffer(ptr, n), device=device)
    arr = np.ctypeslib.as_array((C.c_double * n).from_address(int(ptr)))
    return torch.from_numpy(arr)


def make_alltoallv_hook(group=None, device=None, p2p=False):
    """C callback implementing emi_alltoallv_fn with torch.distributed.all_to_all_single.

    The collective is queued behind the transform kernels already on the current torch stream and the
    kernels launched afterwards wait for it (torch.distributed stream semantics) -- no host sync."""
    import torch
    import torch.distributed as dist
    device = torch.device(device) if device is not None else torch.device("cpu")

    staged = device.type == "cuda" and dist.get_backend(group) == "gloo"
    if p2p and device.type == "cuda" and dist.get_backend(group) == "nccl":
        # With V-sets the first point-to-point call may involve only some tasks (a V-set that holds no field of a call makes no
        # TRMTOL / TRLTOM exchange), but the first batch_isend_irecv on an NCCL group must be entered by ALL its ranks (it creates
        # the communicator): one collective over the whole group here, where every task passes, creates it once for all.
        warm = torch.zeros(1, dtype=torch.float64, device=device)
        dist.all_reduce(warm, group=group)
        torch.cuda.synchronize(device)

    def exchange(recv, send, osz, isz):
        """all_to_all_single over the whole group -- or, with V-sets, point-to-point transfers between the tasks that really
        exchange a block: TRMTOL / TRLTOM run inside a V-set, TRLTOG / TRGTOL inside a W-set, and two V-sets need not make the
        same number of calls (fields per V-set differ, a V-set may hold none), which a collective over all tasks cannot take"""
        if not p2p:
            dist.all_to_all_single(recv, send, output_split_sizes=osz, input_split_sizes=isz, group=group)
            return
        me = dist.get_rank(group)
        ro, so = np.concatenate([[0], np.cumsum(osz)]), np.concatenate([[0], np.cumsum(isz)])
        if isz[me]:
            recv[ro[me]:ro[me + 1]].copy_(send[so[me]:so[me + 1]])
        ops = []
        for r in range(len(isz)):
            if r == me:
                continue
            peer = r if group is None else dist.get_global_rank(group, r)
            if osz[r]:
                ops.append(dist.P2POp(dist.irecv, recv[ro[r]:ro[r + 1]], peer, group))
            if isz[r]:
                ops.append(dist.P2POp(dist.isend, send[so[r]:so[r + 1]], peer, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def hook(user, sb, sc, sd, rb, rc, rd, nproc, stream):
        try:
            scl, sdl = [int(sc[i]) for i in range(nproc)], [int(sd[i]) for i in range(nproc)]
            rcl, rdl = [int(rc[i]) for i in range(nproc)], [int(rd[i]) for i in range(nproc)]
            # blocks are dense and ordered by task: the buffers are exactly sum(counts) long
            assert all(sdl[i] == sum(scl[:i]) for i in range(nproc)) and all(rdl[i] == sum(rcl[:i]) for i in range(nproc))
            send = _as_tensor(sb, sum(scl), device)
            recv = _as_tensor(rb, sum(rcl), device)
            osz, isz = [c // 8 for c in rcl], [c // 8 for c in scl]
            if staged:
                # test configuration only (several ranks sharing one GPU, gloo has no device
                # all-to-all): stage through the host, ordered on the stream the library passed -- only
                # that stream is synchronised, so the library's event dependencies between its Legendre,
                # exchange and FFT streams are really exercised
                es = torch.cuda.ExternalStream(int(stream), device=device) if stream else torch.cuda.current_stream(device)
                es.synchronize()
                with torch.cuda.stream(es):
                    hs = send.cpu()
                hr = torch.empty(recv.shape, dtype=recv.dtype)
                exchange(hr, hs, osz, isz)
                with torch.cuda.stream(es):
                    recv.copy_(hr)
                es.synchronize()
            elif device.type == "cuda" and stream:
                # the transform runs on a caller-supplied HIP stream: make it torch's current stream so
                # that the collective is ordered behind the kernels queued on it and ahead of the next ones
                with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=device)):
                    exchange(recv, send, osz, isz)
            else:
                exchange(recv, send, osz, isz)
            return 0
        except Exception:  # never let an exception cross the C boundary
            traceback.print_exc(file=sys.stderr)
            return -1

    cb = _A2A(hook)
    _keep.append(cb)
    return cb


_BCAST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int)
_GATHV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int)


def make_host_collectives(group=None, device=None):
    """(bcast, allgatherv) callbacks for emi_set_host_collectives: host bytes over torch.distributed (the DIST_x / GATH_x
    routines and SPECNORM with several tasks; not on the transform path)."""
    import torch
    import torch.distributed as dist

    def as_np(ptr, nbytes):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(nbytes),)) if nbytes else np.zeros(0, dtype=np.uint8)

    def bcast(user, buf, nbytes, root):
        try:
            a = as_np(buf, nbytes)
            t = _comm_tensor(a, group, device)
            dist.broadcast(t, src=dist.get_global_rank(group, root) if group is not None else root, group=group)
            if dist.get_rank(group) != root:
                a[:] = t.cpu().numpy()
            return 0
        except Exception:
            traceback.print_exc(file=sys.stderr)
            return -1

    def gathv(user, sb, sbytes, rb, rbytes, displs, nproc):
        try:
            me = dist.get_rank(group)
            out = as_np(rb, max(int(displs[r]) + int(rbytes[r]) for r in range(nproc)))
            for r in range(nproc):  # one broadcast per task: pieces of any size
                piece = out[int(displs[r]):int(displs[r]) + int(rbytes[r])]
                if r == me:
                    piece[:] = as_np(sb, sbytes)
                t = _comm_tensor(piece, group, device)
                dist.broadcast(t, src=dist.get_global_rank(group, r) if group is not None else r, group=group)
                if r != me:
                    piece[:] = t.cpu().numpy()
            return 0
        except Exception:
            traceback.print_exc(file=sys.stderr)
            return -1

    cb = (_BCAST(bcast), _GATHV(gathv))
    _keep.append(cb)
    return cb


def all_reduce_sum(values, group=None, device=None):
    """Sum a small numpy vector over tasks (SPECNORM's gather, spnormc_mod.F90:49-85)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64))
    if device is not None and torch.device(device).type == "cuda" and dist.get_backend(group) != "gloo":
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def _comm_tensor(a, group, device):
    """numpy array -> tensor the group's backend can move (CUDA for RCCL, host otherwise)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(a))
    if device is not None and torch.device(device).type == "cuda" and dist.get_backend(group) != "gloo":
        t = t.to(device)
    return t


def broadcast_from(a, shape, dtype, src, group=None, device=None):
    """Task `src` (0-based) holds numpy array `a`; every task returns a copy (DIST_* helpers)."""
    import torch.distributed as dist
    buf = np.ascontiguousarray(a, dtype=dtype) if dist.get_rank(group) == src else np.empty(shape, dtype=dtype)
    if tuple(buf.shape) != tuple(shape):
        raise ValueError("broadcast_from: array of shape %s, expected %s" % (buf.shape, tuple(shape)))
    t = _comm_tensor(buf, group, device)
    dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return t.cpu().numpy()


def all_gather_padded(a, nmax, group=None, device=None):
    """Every task contributes a (n_i, ...) numpy array with n_i <= nmax; returns the list of the
    tasks' arrays padded to nmax rows (GATH_* helpers: pieces differ in length)."""
    import torch
    import torch.distributed as dist
    pad = np.zeros((nmax,) + a.shape[1:], dtype=a.dtype)
    pad[:a.shape[0]] = a
    t = _comm_tensor(pad, group, device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [o.cpu().numpy() for o in out]


_RCCL = [None]


def rccl_native_lib():
    """libectrans_mi_rccl.so (ectrans_amd/rccl): the transport a Fortran / C host attaches -- one group of ncclSend / ncclRecv
    per exchange on the library's own stream, no Python in the data path."""
    import os
    if _RCCL[0] is None:
        from . import lib
        lib()  # libectrans_mi.so first: the hook library resolves its symbols against the instance already loaded
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl", "libectrans_mi_rccl.so")
        if not os.path.exists(path):
            raise OSError("%s is missing: make -C ectrans_amd/rccl" % path)
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.emi_rccl_get_unique_id.argtypes = [C.c_void_p]
        L.emi_rccl_attach.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        L.emi_rccl_last_error.restype = C.c_char_p
        _RCCL[0] = L
    return _RCCL[0]


def rccl_native_attach(nproc, myproc, kmax_resol=1, kprintlev=0, prad=0.0, device=-1, group=None):
    """SETUP_TRANS0 through the native RCCL transport (emi_rccl_attach): task 1 draws the 128-byte RCCL unique id and the
    process group carries it to the other tasks (torch.distributed here; MPI_Bcast in a Fortran host); every
    task then creates its rank of the communicator, registers the grouped ncclSend / ncclRecv exchange and the host collectives
    of DIST_x / GATH_x, and initialises the library as task `myproc` of `nproc`.  Returns 0 or raises OSError with the
    transport's message."""
    import torch.distributed as dist
    L = rccl_native_lib()
    uid = C.create_string_buffer(128)
    err = None
    if myproc == 1 and L.emi_rccl_get_unique_id(uid):
        err = "emi_rccl_get_unique_id: " + (L.emi_rccl_last_error() or b"").decode()
    box = [(uid.raw if err is None else err) if myproc == 1 else None]
    if nproc > 1:  # a failure on task 1 travels too: every task raises, none is left waiting in the broadcast
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
    if not isinstance(box[0], bytes):
        raise OSError(str(box[0]))
    uid2 = C.create_string_buffer(box[0], 128)
    rc = L.emi_rccl_attach(uid2, int(nproc), int(myproc), int(kmax_resol), int(kprintlev), float(prad or 0.0), int(device if device is not None else -1))
    if rc:
        from . import lib
        msg = (L.emi_rccl_last_error() or b"").decode() or (lib().emi_last_error() or b"").decode()
        raise OSError("emi_rccl_attach failed (%d): %s" % (rc, msg))
    return 0
