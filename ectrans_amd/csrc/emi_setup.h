// emi_setup.h -- host-side (C++) setup of one resolution: Gaussian latitudes/weights, NMEN,
// associated Legendre panels, FFT plans.  Product code: independent of oracle/ (which is test
// infrastructure); both follow the same reference routines, cited per function
// (paths relative to /root/reference/src/trans).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <thread>
#include <vector>

namespace emi {

inline void parallel_for(int n, const std::function<void(int)> &fn) {
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 4;
  if (nt > 64) nt = 64;
  if ((int)nt > n) nt = n > 0 ? n : 1;
  if (nt <= 1) {
    for (int i = 0; i < n; i++) fn(i);
    return;
  }
  std::vector<std::thread> th;
  std::atomic<int> next{0};
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&]() {
      for (;;) {
        int i = next.fetch_add(1);
        if (i >= n) break;
        fn(i);
      }
    });
  for (auto &t : th) t.join();
}

// ---------------------------------------------------------------------------------------
// Gaussian latitudes and weights: Newton iteration on the Fourier series of the ordinary
// Legendre polynomial of degree NDGL (sugaw_mod.F90:157-190 "LLOLD" branch, gawl_mod.F90,
// cpledn_mod.F90:94-129; series coefficients suleg_mod.F90:249-263).  Weights sum to 1.
// ---------------------------------------------------------------------------------------
inline void gauss_latitudes(int ndgl, std::vector<double> &mu, std::vector<double> &w) {
  const int kn = ndgl, half = ndgl / 2, odd = ndgl & 1;
  // Fourier coefficients of P_kn: fn[k] multiplies cos((2k+odd) theta) ... built from the
  // top coefficient downwards
  std::vector<double> full(kn + 1, 0.0);
  double top = 2.0;
  for (int j = 1; j <= kn; j++) top *= std::sqrt(1.0 - 0.25 / ((double)j * (double)j));
  full[kn] = top;
  for (int j = 2; j <= kn - odd; j += 2)
    full[kn - j] = full[kn - j + 2] * (double)((j - 1) * (2 * kn - j + 2)) / (double)(j * (2 * kn - j + 1));
  std::vector<double> fn;
  for (int j = odd; j <= kn; j += 2) fn.push_back(full[j]);
  mu.assign(ndgl, 0.0);
  w.assign(ndgl, 0.0);
  const double pi = 2.0 * std::asin(1.0), eps = 2.220446049250313e-16;
  std::vector<double> theta(half);
  parallel_for(half, [&](int i) {
    int jgl = i + 1;
    double z = (double)(4 * jgl - 1) * pi / (double)(4 * kn + 2);
    double x = z + 1.0 / (std::tan(z) * (double)(8 * kn * kn));
    bool last = false;
    double wt = 0.0;
    for (int it = 0; it <= 20; it++) {
      double pk = odd ? 0.0 : 0.5 * fn[0], dp = 0.0;
      int ik = 1;
      if (!last) {
        for (int jn = 2 - odd; jn <= kn; jn += 2, ik++) {
          pk += fn[ik] * std::cos((double)jn * x);
          dp -= fn[ik] * (double)jn * std::sin((double)jn * x);
        }
        double dx = -pk / dp;
        x += dx;
        if (std::fabs(dx) <= eps * 1000.0) last = true;
      } else {
        for (int jn = 2 - odd; jn <= kn; jn += 2, ik++) dp -= fn[ik] * (double)jn * std::sin((double)jn * x);
        wt = (double)(2 * kn + 1) / (dp * dp);
        break;
      }
    }
    theta[i] = x;
    w[i] = wt;
  });
  for (int i = 0; i < half; i++) {
    mu[i] = std::cos(theta[i]);
    mu[ndgl - 1 - i] = -mu[i];
    w[ndgl - 1 - i] = w[i];
  }
}

// ---------------------------------------------------------------------------------------
// NMEN per latitude (setup_geom_mod.F90:44-78) and NDGLU per wavenumber (:85-97)
// ---------------------------------------------------------------------------------------
inline void wavenumber_cutoffs(int nsmax, int ndgl, const std::vector<int> &nloen, bool reduced,
                               const std::vector<double> &cos2, std::vector<int> &nmen, std::vector<int> &ndglu) {
  const int ndgnh = (ndgl + 1) / 2, lin = ndgl - 1;
  nmen.assign(ndgl, 0);
  auto lim = [&](int j, double sq, int sub) { return (int)((double)(nloen[j] - 1) / (2.0 + sq)) - sub; };
  if (nsmax >= lin || !reduced) {
    for (int j = 0; j < ndgl; j++) nmen[j] = std::min(nsmax, (nloen[j] - 1) / 2);
  } else {
    const bool quad = nsmax >= ndgl * 2 / 3 - 1;
    const int ifac = 3 * (lin - nsmax) / ndgl;  // integer arithmetic as in the reference
    auto sq = [&](int j) { return quad ? (double)ifac * cos2[j] : cos2[j]; };
    const int sub = quad ? 0 : 1;
    nmen[0] = std::min(nsmax, lim(0, sq(0), sub));
    for (int j = 1; j < ndgnh; j++) nmen[j] = std::min(nsmax, std::max(nmen[j - 1], lim(j, sq(j), sub)));
    nmen[ndgl - 1] = std::min(nsmax, lim(ndgl - 1, sq(ndgl - 1), sub));
    for (int j = ndgl - 2; j >= ndgnh; j--) nmen[j] = std::min(nsmax, std::max(nmen[j + 1], lim(j, sq(j), sub)));
  }
  ndglu.assign(nsmax + 1, 0);
  for (int j = 0; j < ndgnh; j++)
    for (int m = 0; m <= std::min(nmen[j], nsmax); m++) ndglu[m]++;
}

// ---------------------------------------------------------------------------------------
// Normalised associated Legendre functions for one (m, mu): values for n = m + par + 2k,
// k = 0..cnt-1, following the per-m recurrence of SUPOLF (supolf_mod.F90:85-247): explicit
// first four values, 4-term n -> n+2 recurrence, 1e+-100 rescaling with the same flooring.
// coef: per-m tables dcl[n], ddl[n] (statement functions supolf_mod.F90:79-83).
// ---------------------------------------------------------------------------------------
struct LegCoef {
  int m, nmax;
  std::vector<double> dcl, ddl;  // index n
  double zfac_m;                 // sqrt(2m-1) * prod_{j<m} sqrt((2j-1)/(2j))
};

inline LegCoef legendre_coefficients(int m, int nmax) {
  LegCoef c;
  c.m = m;
  c.nmax = nmax;
  c.dcl.assign(nmax + 1, 0.0);
  c.ddl.assign(nmax + 1, 0.0);
  for (int k = std::max(m, 1); k <= nmax; k++) {
    c.dcl[k] = std::sqrt(((double)(k - m + 1) * (double)(k - m + 2) * (double)(k + m + 1) * (double)(k + m + 2)) /
                         ((double)(2 * k + 1) * (double)(2 * k + 3) * (double)(2 * k + 3) * (double)(2 * k + 5)));
    c.ddl[k] = (2.0 * (double)k * (double)(k + 1) - 2.0 * (double)(m * m) - 1.0) / ((double)(2 * k - 1) * (double)(2 * k + 3));
  }
  double zfac = 1.0;
  for (int j = 1; j <= m - 1; j++) {
    zfac *= std::sqrt((double)(2 * j - 1));
    zfac /= std::sqrt((double)(2 * j));
  }
  if (m >= 1) zfac *= std::sqrt((double)(2 * m - 1));
  c.zfac_m = zfac;
  return c;
}

// out[n] for n = 0..nmax (only the entries of parity `par` relative to m are defined for m>=2)
inline void legendre_column(const LegCoef &c, double mu_in, int par, double *out, int *corr /* [nmax+1] scratch */) {
  const int m = c.m, nmax = c.nmax;
  const double eps = 2.220446049250313e-16;
  double x = mu_in;
  double c2 = 1.0 - x * x, cs = std::sqrt(c2), csr;
  if (std::fabs(cs) <= eps) {
    x = 1.0;
    cs = 0.0;
    csr = 0.0;
    c2 = 0.0;
  } else {
    csr = 1.0 / cs;
  }
  if (m <= 1) {
    // ordinary Legendre 3-term recurrence (supolf_mod.F90:124-142; DFA/DFB/DFF/DFG/DFI of
    // tpm_pol.F90:73-80)
    double km2 = 1.0, km1 = x;
    if (m == 0) {
      out[0] = km2;
      if (nmax >= 1) out[1] = km1 * std::sqrt(3.0 / 2.0) / (1.0 / std::sqrt(2.0));
    } else {
      out[0] = 0.0;
      if (nmax >= 1) out[1] = cs * std::sqrt(3.0 / 2.0);
    }
    for (int n = 2; n <= nmax; n++) {
      double dff = (double)(2 * n - 1) / (double)n, dfg = (double)(n - 1) / (double)n;
      double dfb = std::sqrt((double)(2 * n + 1) / (double)(n * (n + 1)));
      double k = dff * x * km1 - dfg * km2;
      if (m == 0) {
        double dfa = 1.0 / std::sqrt((double)(n * (n + 1)));
        out[n] = k * dfb / dfa;
      } else {
        out[n] = ((double)n * (km1 - x * k) * csr) * dfb;
      }
      km2 = km1;
      km1 = k;
    }
    return;
  }
  const double big = 1.0e+100, small = 1.0e-100;
  int corr3 = 0;
  double lsita = 1.0;
  for (int j = 1; j <= m / 2; j++) {
    lsita *= c2;
    if (std::fabs(lsita) < small) {
      lsita *= big;
      corr3++;
    }
  }
  if (m & 1) lsita *= cs;
  double zfac = c.zfac_m, zfac0 = 1.0, zfac1 = 1.0, mult = 0.0;
  const int icmax = std::min(nmax - m, 3);
  for (int ic = 0; ic <= icmax; ic++) {
    zfac0 *= (double)(2 * m + ic);
    switch (ic) {
      case 0: zfac1 = 1.0; mult = zfac; break;
      case 1: zfac1 = 1.0; zfac *= (double)(2 * m + 1); mult = zfac * x; break;
      case 2: zfac1 = 2.0; mult = 0.5 * zfac * ((double)(2 * m + 3) * x * x - 1.0); break;
      case 3: zfac1 = 6.0; zfac *= (double)(2 * m + 3); mult = (1.0 / 6.0) * x * zfac * ((double)(2 * m + 5) * x * x - 3.0); break;
    }
    out[m + ic] = lsita * mult * std::sqrt(2.0 * ((double)(m + ic) + 0.5) * zfac1 / zfac0);
  }
  for (int n = 0; n <= nmax; n++) corr[n] = corr3;
  for (int n = m + par + 4; n <= nmax; n += 2) {
    if (std::fabs(out[n - 4]) > big) {
      for (int j = n - 4; j <= n - 1; j++) out[j] /= big;
      for (int j = n - 4; j <= nmax; j++) corr[j] -= 1;
    }
    out[n] = ((x * x - c.ddl[n - 2]) * out[n - 2] - c.dcl[n - 4] * out[n - 4]) / c.dcl[n - 2];
  }
  for (int n = m + par; n <= nmax; n += 2)
    for (int j = 1; j <= corr[n]; j++) {
      out[n] /= big;
      if (out[n] < eps) out[n] = eps;  // sic: the reference has no ABS here (supolf_mod.F90:241-243)
    }
}

// ---------------------------------------------------------------------------------------
// Belousov's generator (SETUP_TRANS with LDUSERPNM=.TRUE., the default of the Fortran API): every
// P_n^m(mu), m <= n <= nmax, of ONE latitude (supol_mod.F90:86-167): m = 0, 1 from the Fourier series
// of the ordinary Legendre polynomials (coefficient table of suleg_mod.F90:251-263), the diagonal by
// Belousov's equation (23), the rest by his three-term recurrence (17) with the coefficients of
// tpm_pol.F90:43-52.  pol[m * (nmax + 1) + n].
// ---------------------------------------------------------------------------------------
struct BelousovTables {
  int nmax = 0;
  std::vector<double> fn;             // [(nmax+1)^2]  fn[n * (nmax+1) + k]: coefficient of cos(k theta) in P_n
  std::vector<double> c17, d17, e17;  // [(nmax+1)^2]  recurrence coefficients, index m * (nmax+1) + n
  std::vector<double> a1, h23;        // [nmax+1]      1 / sqrt(n (n+1)),  sqrt((2n+1) / (2n))
};
inline BelousovTables belousov_tables(int nmax) {
  BelousovTables T;
  T.nmax = nmax;
  const size_t ld = (size_t)nmax + 1;
  T.fn.assign(ld * ld, 0.0);
  T.fn[0] = 2.0;
  for (int n = 1; n <= nmax; n++) {
    double top = 2.0;
    for (int j = 1; j <= n; j++) top *= std::sqrt(1.0 - 0.25 / ((double)j * (double)j));
    const int odd = n & 1;
    double *row = T.fn.data() + (size_t)n * ld;
    row[n] = top;
    for (int j = 2; j <= n - odd; j += 2) row[n - j] = row[n - j + 2] * (double)((j - 1) * (2 * n - j + 2)) / (double)(j * (2 * n - j + 1));
  }
  T.c17.assign(ld * ld, 0.0);
  T.d17.assign(ld * ld, 0.0);
  T.e17.assign(ld * ld, 0.0);
  for (int n = 3; n <= nmax; n++)
    for (int m = 2; m <= n - 1; m++) {
      const double dn = n, dm = m;
      T.c17[(size_t)m * ld + n] = std::sqrt(((2 * dn + 1) * (dn + dm - 1) * (dn + dm - 3)) / ((2 * dn - 3) * (dn + dm) * (dn + dm - 2)));
      T.d17[(size_t)m * ld + n] = std::sqrt(((2 * dn + 1) * (dn + dm - 1) * (dn - dm + 1)) / ((2 * dn - 1) * (dn + dm) * (dn + dm - 2)));
      T.e17[(size_t)m * ld + n] = std::sqrt(((2 * dn + 1) * (dn - dm)) / ((2 * dn - 1) * (dn + dm)));
    }
  T.a1.assign(ld, 0.0);
  T.h23.assign(ld, 0.0);
  for (int n = 1; n <= nmax; n++) {
    T.a1[n] = 1.0 / std::sqrt((double)n * (double)(n + 1));
    T.h23[n] = std::sqrt((double)(2 * n + 1) / (double)(2 * n));
  }
  return T;
}
inline void belousov_latitude(const BelousovTables &T, double mu, double *pol) {
  const int nmax = T.nmax;
  const size_t ld = (size_t)nmax + 1;
  double x = mu;
  const double theta = std::acos(x);
  double sita = std::sqrt(1.0 - x * x), rsita;
  if (std::fabs(sita) <= std::sqrt(2.220446049250313e-16)) {  // closer than a metre to the pole
    x = 1.0;
    sita = 0.0;
    rsita = 0.0;
  } else {
    rsita = 1.0 / sita;
  }
  pol[0] = 1.0;
  // m = 0, 1: series over the cosines / sines of the same parity as n
  for (int n = 1; n <= nmax; n++) {
    const double *f = T.fn.data() + (size_t)n * ld;
    const int odd = n & 1;
    double p0 = odd ? 0.0 : 0.5 * f[0], p1 = 0.0;
    for (int k = 2 - odd; k <= n; k += 2) {
      p0 += f[k] * std::cos((double)k * theta);
      p1 += T.a1[n] * f[k] * (double)k * std::sin((double)k * theta);
    }
    pol[n] = p0;
    pol[ld + n] = p1;
  }
  // diagonal, with the reference's flush of values that have underflowed relative to 1 / sin
  const double floor_ = rsita * 2.2250738585072014e-308;
  for (int n = 2; n <= nmax; n++) {
    double v = pol[(size_t)(n - 1) * ld + (n - 1)] * sita * T.h23[n];
    if (std::fabs(v) < floor_) v = 0.0;
    pol[(size_t)n * ld + n] = v;
  }
  for (int n = 3; n <= nmax; n++)
    for (int m = 2; m <= n - 1; m++)
      pol[(size_t)m * ld + n] = T.c17[(size_t)m * ld + n] * pol[(size_t)(m - 2) * ld + (n - 2)] -
                                T.d17[(size_t)m * ld + n] * pol[(size_t)(m - 2) * ld + (n - 1)] * x +
                                T.e17[(size_t)m * ld + n] * pol[(size_t)m * ld + (n - 1)] * x;
}

// ---------------------------------------------------------------------------------------
// FFT planning
// ---------------------------------------------------------------------------------------
// Factor list in DIT order.  Powers of two first (radix 8, then one of 4/2), odd radices last, so
// that every pass stride of a 2-3-5-smooth size is a power of two.  Radix 8 keeps the butterfly
// at 32 data VGPRs, so the FFT kernels fit 128 VGPRs and every CU holds 16 waves.
inline bool factorize_smooth(int s, std::vector<int> &fac) {
  fac.clear();
  int e2 = 0;
  while (s % 2 == 0 && s > 1) {
    s /= 2;
    e2++;
  }
  // radix 8 keeps the butterfly at 32 data VGPRs: the FFT kernels then fit 128 VGPRs without
  // spills and every CU holds 16 waves (a radix-16 variant was measured 40% slower: it spills)
  for (; e2 >= 3; e2 -= 3) fac.push_back(8);
  if (e2 == 2) fac.push_back(4);
  if (e2 == 1) fac.push_back(2);
  static const int rad[] = {3, 5, 7};
  for (int r : rad)
    while (s % r == 0 && s > 1) {
      fac.push_back(r);
      s /= r;
    }
  return s == 1;
}
// The last two factors (2,3), (3,3), (2,5) as one composite radix 6, 9, 10 (specialised FFT kernels only:
// their butterflies do the two steps in registers).  Returns false when the list does not end that way.
inline bool merge_tail(const std::vector<int> &fac, std::vector<int> &out) {
  const size_t n = fac.size();
  if (n < 2) return false;
  const int a = fac[n - 2], b = fac[n - 1];
  if (!((a == 2 && b == 3) || (a == 3 && b == 3) || (a == 2 && b == 5))) return false;
  if (n >= 3 && fac[n - 3] != 8) return false;  // e.g. 8,8,4,... 3,3,3 or 2,3,5 tails stay as they are
  out.assign(fac.begin(), fac.end() - 2);
  out.push_back(a * b);
  return true;
}
// Bluestein work length: among 2^a * {1,3,5,9,15} >= n, the one minimising L * (passes + 1)
inline int next_235(int n) {
  static const int odd[] = {1, 3, 5, 9, 15};
  long long best = -1, bestcost = 0;
  for (int o : odd) {
    long long v = o;
    while (v < n) v *= 2;
    std::vector<int> f;
    factorize_smooth((int)v, f);
    long long cost = v * (long long)(f.size() + 1);
    if (best < 0 || cost < bestcost) {
      best = v;
      bestcost = cost;
    }
  }
  return (int)best;
}
inline void dit_positions(int S, const std::vector<int> &fac, std::vector<uint16_t> &perm) {
  perm.assign(S, 0);
  for (int i = 0; i < S; i++) {
    int rem = i, size = S, pos = 0;
    for (int q = (int)fac.size() - 1; q >= 0; q--) {
      int r = fac[q];
      size /= r;
      pos += (rem % r) * size;
      rem /= r;
    }
    perm[i] = (uint16_t)pos;
  }
}

}  // namespace emi
