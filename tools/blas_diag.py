"""diagnostic: where does bench.py's cpu_baseline_blas spend its time on this host?  python tools/blas_diag.py NSMAX NF THREADS"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, scipy.fft
from concurrent.futures import ThreadPoolExecutor
from oracle.oracle import Oracle
from tests.common import octahedral, random_spectrum
N, nf, thr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t0 = time.time(); o = Oracle(N, octahedral(N)); print("oracle setup %.1fs" % (time.time() - t0), flush=True)
torch.set_num_threads(thr)
nloen = octahedral(N); H = len(nloen) // 2
t0 = time.time()
Ps = []
for m in range(N + 1):
    ps = o.rpnm(m, True)
    Ps.append(torch.from_numpy(ps[::-1][:(N - m) // 2 + 1].copy()))
print("panels %.1fs" % (time.time() - t0), flush=True)
rng = np.random.default_rng(1)
xs = [torch.from_numpy(rng.standard_normal((Ps[m].shape[0], 2 * nf))) for m in range(N + 1)]
t0 = time.time()
outs = [(Ps[m].T @ xs[m]) for m in range(N + 1)]
print("%d matmuls with %d threads: %.2fs" % (N + 1, thr, time.time() - t0), flush=True)
FN = np.zeros((H, N + 1, 2 * nf))
t0 = time.time()
for m in range(N + 1):
    K = outs[m].shape[0]
    FN[H - K:, m] = outs[m].numpy()
print("scatter into FN: %.2fs" % (time.time() - t0), flush=True)
off = np.concatenate([[0], np.cumsum(nloen)])
grid = np.zeros((nf, int(off[-1])))
def row(j):
    n = int(nloen[j]); jn = j if j < H else 2 * H - 1 - j
    M = min(int(o.nmen[j]), n // 2)
    X = np.zeros((nf, n // 2 + 1), dtype=np.complex128)
    F = FN[jn, :M + 1].reshape(M + 1, 2, nf)
    X[:, :M + 1] = (F[:, 0] + 1j * F[:, 1]).T
    grid[:, off[j]:off[j + 1]] = scipy.fft.irfft(X, n, axis=1) * n
for w in (1, 8, 64):
    pool = ThreadPoolExecutor(max_workers=w)
    t0 = time.time(); list(pool.map(row, range(2 * H))); print("irfft rows, pool of %d: %.2fs" % (w, time.time() - t0), flush=True)
    pool.shutdown()
t0 = time.time()
for j in range(2 * H): pass
X = [np.zeros((nf, int(nloen[j]) // 2 + 1), dtype=np.complex128) for j in range(2 * H)]
t0 = time.time()
for j in range(2 * H): scipy.fft.irfft(X[j], int(nloen[j]), axis=1, workers=1)
print("bare irfft serial: %.2fs" % (time.time() - t0), flush=True)
