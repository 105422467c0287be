// emi_mr_body.h -- direct mixed-radix FFT kernels k_fft_dir_mr / k_fft_inv_mr (round 3), included by emi_kernels_body.h
// inside namespace emi_f64 / emi_f32.  (No include guard on purpose.)
// ==========================================================================================
// Rows whose half-length sz = NLOEN / 2 is a product of at most three radices A B C, each from {2..16, 17, 19, 23} (fp32 library: also 18, 20, 21), need no
// chirp-z convolution: at TCo1279 that is 24 % of the grid by weight (7.7 % with a 7-smooth half-length), and a direct
// transform does about a sixth of the arithmetic of the three work-length transforms of a Bluestein row.  What the
// reference does for every length (FFTW plans, tpm_fftw.F90:251-377; ftdir_mod.F90:67-84 / ftinv_mod.F90:65-84) --
// here with register-resident butterflies:
//   * decimation in frequency, in place: point n = n1 BC + n2 C + n3, coefficient k = k1 + A k2 + AB k3;
//       pass 1: BC butterflies of radix A over n1
//       pass 2: AC butterflies of radix B over n2, input n2 times the twiddle w_N^(C k1 n2)
//       pass 3: AB butterflies of radix C over n3, input n3 times the twiddle w_N^((k1 + A k2) n3)
//     a butterfly's twiddles are the powers of ONE table value (w_N^(C k1) / w_N^(k1 + A k2)), loaded beside its LDS reads and
//     raised by a product chain (a table value per output behind a branch serialised R second-level-cache round trips per
//     butterfly: 25 ps per point and field);
//     every butterfly writes where it read, so a pass needs no barrier between its loads and stores and the LDS holds the row once:
//     element (n1 | k1, n2 | k2, n3 | k3) at n1 P1 + n2 C + n3, P1 = BC made odd -- the three access patterns (lanes along
//     j = n2 C + n3; along n3 then k1; along k1 then k2) are then free of bank conflicts up to the wrap of the faster index;
//   * a butterfly lives in registers: odd primes as the symmetric DFT matrix (sums / differences of x_j, x_(P-j); 4 m^2 + 10 m
//     operations, m = (P-1)/2, the cos / sin entries instruction literals), 2 and 4 by hand, composite radices (6 ... 21) as two such
//     stages with constant twiddles; outputs are stored as they are produced (radix 19 fits 125 VGPRs, radix 23 spills 24);
//   * the first pass of the direct transform reads the grid row straight from memory (TRGTOL local copy), the last pass of the
//     inverse transform writes it (TRLTOG local copy): thread j handles points j + cnt r, coalesced for every r;
//   * the inverse transform is conj(DFT(conj Z)): one set of (forward) butterflies and twiddle tables;
//   * several fields per workgroup for short rows (the 40-KiB rule of the generic kernels): thread -> (field, butterfly) by a
//     multiply-high division, so that lanes stay busy when a pass has fewer butterflies than the workgroup threads.
// The coefficients end up digit-reversed, k at (k mod A) P1 + ((k / A) mod B) C + k / (AB).
// ==========================================================================================

// -DEMI_MR_STAMP (experiments only): wave 0 of every workgroup adds the clock ticks it spent between consecutive MR_STAMP points to
// emi_mr_stamp[] (read with emi_debug_mr_stamps)
#if defined(EMI_MR_STAMP) && !defined(EMI_LEG_STAMP) && !defined(EMI_CPU_EMU)
#define MR_STAMP_BEGIN()                  \
  unsigned long long st_acc[10] = {0};    \
  unsigned long long st_prev = __builtin_readcyclecounter()
#define MR_STAMP(i_)                                                  \
  do {                                                                \
    const unsigned long long st_now = __builtin_readcyclecounter();   \
    st_acc[i_] += st_now - st_prev;                                   \
    st_prev = st_now;                                                 \
  } while (0)
#define MR_STAMP_END()                                                                \
  do {                                                                                \
    if (EMI_TID == 0) {                                                               \
      for (int i_ = 0; i_ < 10; i_++) atomicAdd(&emi_mr_stamp[i_ < 7 ? i_ : i_ + 1], st_acc[i_]); \
      atomicAdd(&emi_mr_stamp[7], 1ull);                                              \
    }                                                                                 \
  } while (0)
#else
#define MR_STAMP_BEGIN() ((void)0)
#define MR_STAMP(i_) ((void)0)
#define MR_STAMP_END() ((void)0)
#endif

template <int I, int N, class F>
EMI_DEVFN void mr_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    mr_for<I + 1, N>(f);
  }
}

// x *= exp(-2 pi i T / R), T and R compile-time constants
template <int R, int T>
EMI_DEVFN void mr_mulw(real_t &xr, real_t &xi) {
  constexpr int t = ((T % R) + R) % R;
  if constexpr (t == 0) {
  } else if constexpr (4 * t == R) {  // -i
    const real_t a = xr;
    xr = xi, xi = -a;
  } else if constexpr (2 * t == R) {
    xr = -xr, xi = -xi;
  } else if constexpr (4 * t == 3 * R) {  // +i
    const real_t a = xr;
    xr = -xi, xi = a;
  } else if constexpr (8 * t == R) {  // (1 - i) / sqrt 2
    constexpr real_t h = (real_t)0.70710678118654752440;
    const real_t a = (xr + xi) * h;
    xi = (xi - xr) * h, xr = a;
  } else if constexpr (8 * t == 3 * R) {  // (-1 - i) / sqrt 2
    constexpr real_t h = (real_t)0.70710678118654752440;
    const real_t a = (xi - xr) * h;
    xi = -(xr + xi) * h, xr = a;
  } else if constexpr (8 * t == 5 * R) {  // (-1 + i) / sqrt 2
    constexpr real_t h = (real_t)0.70710678118654752440;
    const real_t a = -(xr + xi) * h;
    xi = (xr - xi) * h, xr = a;
  } else if constexpr (8 * t == 7 * R) {  // (1 + i) / sqrt 2
    constexpr real_t h = (real_t)0.70710678118654752440;
    const real_t a = (xr - xi) * h;
    xi = (xr + xi) * h, xr = a;
  } else {
    constexpr real_t c = (real_t)mr_cos<R>(t), s = (real_t)mr_sin<R>(t);
    const real_t a = xr * c + xi * s;
    xi = xi * c - xr * s, xr = a;
  }
}

// how a composite radix splits: R = mr_split_a(R) * (R / mr_split_a(R)); 0: prime (or 4), a butterfly of its own
EMI_DEVFN constexpr int mr_split_a(int R) {
  return R == 6 ? 2 : R == 8 ? 2 : R == 9 ? 3 : R == 10 ? 2 : R == 12 ? 3 : R == 14 ? 2 : R == 15 ? 3 : R == 16 ? 4 : R == 18 ? 2 : R == 20 ? 4 : R == 21 ? 3 : 0;
}

// forward DFT of R values in registers; emit(integral_constant<k>, re, im) is called once per output
template <int R>
struct MrDft {
  template <class Emit>
  static EMI_DEVFN void run(const real_t *xr, const real_t *xi, Emit &&emit) {
    constexpr int SA = mr_split_a(R);
    if constexpr (R == 1) {
      emit(std::integral_constant<int, 0>{}, xr[0], xi[0]);
    } else if constexpr (R == 2) {
      emit(std::integral_constant<int, 0>{}, xr[0] + xr[1], xi[0] + xi[1]);
      emit(std::integral_constant<int, 1>{}, xr[0] - xr[1], xi[0] - xi[1]);
    } else if constexpr (R == 4) {
      const real_t ar = xr[0] + xr[2], ai = xi[0] + xi[2], br = xr[0] - xr[2], bi = xi[0] - xi[2];
      const real_t cr = xr[1] + xr[3], ci = xi[1] + xi[3], dr = xr[1] - xr[3], di = xi[1] - xi[3];
      emit(std::integral_constant<int, 0>{}, ar + cr, ai + ci);
      emit(std::integral_constant<int, 1>{}, br + di, bi - dr);  // b - i d
      emit(std::integral_constant<int, 2>{}, ar - cr, ai - ci);
      emit(std::integral_constant<int, 3>{}, br - di, bi + dr);  // b + i d
    } else if constexpr (SA == 0) {
      // odd prime: X_k = x_0 + sum_j cos(jk) s_j - i sum_j sin(jk) d_j, X_(R-k) the same with + i; s_j = x_j + x_(R-j), d_j = x_j - x_(R-j)
      constexpr int M = (R - 1) / 2;
      real_t sr[M], si[M], dr[M], di[M];
      real_t y0r = xr[0], y0i = xi[0];
      mr_for<0, M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        sr[j] = xr[j + 1] + xr[R - 1 - j], si[j] = xi[j + 1] + xi[R - 1 - j];
        dr[j] = xr[j + 1] - xr[R - 1 - j], di[j] = xi[j + 1] - xi[R - 1 - j];
        y0r += sr[j], y0i += si[j];
      });
      emit(std::integral_constant<int, 0>{}, y0r, y0i);
      mr_for<1, M + 1>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        real_t ar = xr[0], ai = xi[0], br = 0, bi = 0;
        mr_for<0, M>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int t = ((j + 1) * k) % R;
          constexpr real_t c = (real_t)mr_cos<R>(t), s = (real_t)mr_sin<R>(t);
          ar += c * sr[j], ai += c * si[j];
          if constexpr (j == 0)
            br = s * dr[j], bi = s * di[j];
          else
            br += s * dr[j], bi += s * di[j];
        });
        emit(std::integral_constant<int, k>{}, ar + bi, ai - br);
        emit(std::integral_constant<int, R - k>{}, ar - bi, ai + br);
      });
    } else {
      // R = SA * SB, n = na SB + nb, k = ka + SA kb: SB transforms of length SA, constant twiddles w_R^(nb ka), SA of length SB
      constexpr int SB = R / SA;
      real_t tr[R], ti[R];  // [nb][ka]
      mr_for<0, SB>([&](auto nbc) {
        constexpr int nb = decltype(nbc)::value;
        real_t ir[SA], ii[SA];
        mr_for<0, SA>([&](auto nac) {
          constexpr int na = decltype(nac)::value;
          ir[na] = xr[na * SB + nb], ii[na] = xi[na * SB + nb];
        });
        MrDft<SA>::run(ir, ii, [&](auto kac, real_t re, real_t im) {
          constexpr int ka = decltype(kac)::value;
          mr_mulw<R, nb * ka>(re, im);
          tr[nb * SA + ka] = re, ti[nb * SA + ka] = im;
        });
      });
      mr_for<0, SA>([&](auto kac) {
        constexpr int ka = decltype(kac)::value;
        real_t ir[SB], ii[SB];
        mr_for<0, SB>([&](auto nbc) {
          constexpr int nb = decltype(nbc)::value;
          ir[nb] = tr[nb * SA + ka], ii[nb] = ti[nb * SA + ka];
        });
        MrDft<SB>::run(ir, ii, [&](auto kbc, real_t re, real_t im) {
          constexpr int kb = decltype(kbc)::value;
          emit(std::integral_constant<int, ka + SA * kb>{}, re, im);
        });
      });
    }
  }
};

// u = j / D for j, D < 65536 (m = mr_magic(D)); D = 1: m = 0, callers test
EMI_DEVFN unsigned mr_magic(unsigned D) { return D > 1 ? 0xFFFFFFFFu / D + 1u : 0u; }
EMI_DEVFN unsigned mr_div(unsigned j, unsigned m) { return m ? (unsigned)(((unsigned long long)j * m) >> 32) : j; }

struct MrPassArgs {
  int cnt;             // butterflies per field
  unsigned mcnt, mD;   // magic numbers of cnt and of D
  int D, SU, SV, ES;   // butterfly j -> (u, v) = (j / D, j mod D); its elements at u SU + v SV + r ES
  const real2 *tw;     // input r of butterfly j = (u, v) is multiplied by w^r, w = tw[tw_by_u ? u : j]; null: no twiddles
  int tw_by_u;
};

// the grid row of one field when it lies inside one NPROMA block: its first element
EMI_DEVFN real_t *mr_row(const GridFld &gf, long long blk0, unsigned rem0, int nproma) {
  return (real_t *)gf.base + (blk0 * gf.nf_arr + gf.fidx) * (long long)nproma + rem0;
}
EMI_DEVFN bool mr_row_flat(const GridFld &gf, long long blk0, unsigned rem0, int nproma, int n) {
  return (rem0 + (unsigned)n <= (unsigned)nproma) && ((((uintptr_t)mr_row(gf, blk0, rem0, nproma)) & (2 * sizeof(real_t) - 1)) == 0);
}

// One pass.  IO = 0: LDS -> LDS; 1: grid -> LDS (first pass of the direct transform: z_l = x_2l + i x_(2l+1), l = j + cnt r);
// 2: LDS -> grid (last pass of the inverse transform: x_2i = Re y_i, x_(2i+1) = -Im y_i, i = j + cnt k).  IO = 1, 2 only for rows that
// lie inside one NPROMA block, 2-element aligned (mr_row_flat; the kernels copy other rows through the LDS).
template <int R, int IO>
EMI_DEVFN void mr_pass(real2 *a, int fs, int nfl, const MrPassArgs &pa, const GridFld *flds, long long blk0, unsigned rem0, int nproma) {
  const int ntot = nfl * pa.cnt;
  for (int gi = EMI_TID; gi < ntot; gi += EMI_NTHREADS) {
    const int f = nfl > 1 ? (int)mr_div((unsigned)gi, pa.mcnt) : 0;
    const int j = gi - f * pa.cnt;
    const int u = pa.D > 1 ? (int)mr_div((unsigned)j, pa.mD) : j;
    const int v = j - u * pa.D;
    real2 *p = a + (long long)f * fs + u * pa.SU + v * pa.SV;
    real_t xr[R], xi[R];
    real2 w = mk2(1, 0);
    if (IO != 1 && pa.tw) {
      EMI_GLOBAL_AS const real2 *tw = (EMI_GLOBAL_AS const real2 *)pa.tw + (pa.tw_by_u ? u : j);
      w = mk2(tw->x, tw->y);
    }
    EMI_GLOBAL_AS real2 *row = nullptr;
    if constexpr (IO != 0) row = (EMI_GLOBAL_AS real2 *)mr_row(flds[f], blk0, rem0, nproma) + j;
    if constexpr (IO == 1) {
      mr_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        xr[r] = row[pa.cnt * r].x, xi[r] = row[pa.cnt * r].y;
      });
    } else {
      mr_for<0, R>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const real2 z = p[r * pa.ES];
        xr[r] = z.x, xi[r] = z.y;
      });
    }
    if (IO != 1 && pa.tw) {  // (uniform) x_r *= w^r
      if constexpr (sizeof(real_t) == 4 && (R > 8)) {
        // fp32 library: four interleaved chains w^r = w^(r-4) w^4 instead of one chain w^r = w^(r-1) w -- the rounding error of a
        // power grows with the number of products behind it, 22 at radix 23 (about 1e-6 relative in single precision, which the
        // yardstick of tests/common.py sees on the long rows of TCo2559), 2 + r / 4 here, for the same one product per power.
        // The fp64 kernels keep the single chain: 20 double ulps are far below their tolerance and radix 19 / 23 have no
        // registers for four live powers.
        real_t qr[4], qi[4];
        qr[1] = w.x, qi[1] = w.y;
        mr_for<1, R>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          if constexpr (r == 2) {
            qr[2] = w.x * w.x - w.y * w.y, qi[2] = 2 * w.x * w.y;
          } else if constexpr (r == 3) {
            qr[3] = qr[2] * w.x - qi[2] * w.y, qi[3] = qr[2] * w.y + qi[2] * w.x;
          } else if constexpr (r == 4) {
            qr[0] = qr[2] * qr[2] - qi[2] * qi[2], qi[0] = 2 * qr[2] * qi[2];
            w = mk2(qr[0], qi[0]);  // w^4 from here on
          } else if constexpr (r > 4) {
            const real_t t0 = qr[r & 3] * w.x - qi[r & 3] * w.y;
            qi[r & 3] = qr[r & 3] * w.y + qi[r & 3] * w.x, qr[r & 3] = t0;
          }
          const real_t t1 = xr[r] * qr[r & 3] - xi[r] * qi[r & 3];
          xi[r] = xr[r] * qi[r & 3] + xi[r] * qr[r & 3], xr[r] = t1;
        });
      } else {
        real_t cr = w.x, ci = w.y;
        mr_for<1, R>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          if constexpr (r > 1) {
            const real_t t0 = cr * w.x - ci * w.y;
            ci = cr * w.y + ci * w.x, cr = t0;
          }
          const real_t t1 = xr[r] * cr - xi[r] * ci;
          xi[r] = xr[r] * ci + xi[r] * cr, xr[r] = t1;
        });
      }
    }
    MrDft<R>::run(xr, xi, [&](auto kc, real_t re, real_t im) {
      constexpr int k = decltype(kc)::value;
      if constexpr (IO == 2)
        row[pa.cnt * k].x = re, row[pa.cnt * k].y = -im;
      else
        p[k * pa.ES] = mk2(re, im);
    });
  }
}

template <int IO>
EMI_DEVFN void mr_pass_any(int R, real2 *a, int fs, int nfl, const MrPassArgs &pa, const GridFld *flds, long long blk0, unsigned rem0, int nproma) {
  switch (R) {
#define EMI_MR_CASE(r_) \
  case r_: mr_pass<r_, IO>(a, fs, nfl, pa, flds, blk0, rem0, nproma); break;
    EMI_MR_RADICES(EMI_MR_CASE)
    EMI_MR_EXTRA(EMI_MR_CASE)
#undef EMI_MR_CASE
    default: break;
  }
}

// the three passes of a plan (A, B, C; B or C may be 1): `first_io` / `last_io` select the grid-side variants of the first and of the
// last pass that exists
struct MrGeom {
  int A, B, C, P1, fs;
};
EMI_DEVFN MrGeom mr_geom(int abc) {  // FftRowDev.mr_abc
  MrGeom m;
  m.A = abc & 255, m.B = (abc >> 8) & 255, m.C = (abc >> 16) & 255;
  const int bc = m.B * m.C;
  m.P1 = bc | 1;
  m.fs = m.A * m.P1;
  return m;
}
EMI_DEVFN MrPassArgs mr_args(const MrGeom &m, int ip, const real2 *tw1, const real2 *tw2) {
  MrPassArgs pa;
  if (ip == 0) {
    pa.cnt = m.B * m.C, pa.D = 1, pa.SU = 1, pa.SV = 0, pa.ES = m.P1, pa.tw = nullptr, pa.tw_by_u = 0;
  } else if (ip == 1) {
    pa.cnt = m.A * m.C, pa.D = m.C, pa.SU = m.P1, pa.SV = 1, pa.ES = m.C, pa.tw = tw1, pa.tw_by_u = 1;
  } else {
    pa.cnt = m.A * m.B, pa.D = m.A, pa.SU = m.C, pa.SV = m.P1, pa.ES = 1, pa.tw = tw2, pa.tw_by_u = 0;
  }
  pa.mcnt = mr_magic((unsigned)pa.cnt);
  pa.mD = mr_magic((unsigned)pa.D);
  return pa;
}

// (four waves per SIMD; three -- 146 registers, no scratch instead of 76 bytes -- were measured: the long rows +-1 %, the short ones 20 % slower)
EMI_KERNEL_MR(4) void k_fft_dir_mr(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, real_t *FB, int ldf,
                                                  int nproma) {
  EMI_LDS_DECL;
  real2 *a = (real2 *)EMI_LDS_PTR;
  const int bid = (int)xcd_swizzle(EMI_BID, Lc.nblocks, 8);
  const int li = bid / Lc.nchunk;
  const FftRowDev rw_ = Lc.rows[li];  // one 64-byte record instead of the chain lats -> planid -> plans -> ..., nmen / fbase / gpoff [lat]
  const FftPlanDev &pl = T.plans[rw_.planid];  // (the permutation table of rows cut by NPROMA blocks only)
  const int fbk = (int)((unsigned)rw_.mr_abc >> 24);
  const int f0 = (bid - li * Lc.nchunk) * fbk;
  const int nfl = (nfld - f0) < fbk ? (nfld - f0) : fbk;
  const int n = rw_.n, sz = rw_.sz, nmen = rw_.nmen;
  const MrGeom m = mr_geom(rw_.mr_abc);
  const int fb0 = rw_.fb0;
  const int *frow = g.fftrow ? g.fftrow + fb0 : nullptr;
  const real2 *rtw = (const real2 *)T.rtw + rw_.rtw_off;
  const real2 *tw1 = (const real2 *)T.ptw + rw_.ptw_off0, *tw2 = tw1 + m.A;
  const long long gp0 = rw_.gpoff;
  const long long blk0 = gp0 / nproma;
  const unsigned rem0 = (unsigned)(gp0 - blk0 * nproma);
  // stages 1 + 2 (TRGTOL local copy + FTDIR): the first pass reads the grid rows -- unless NPROMA blocks cut them (or a field is not
  // 2-element aligned): those go to the LDS first, point l at (l / BC) P1 + l mod BC
  bool flat = true;
  for (int fl = 0; fl < nfl; fl++) flat = flat && mr_row_flat(flds[f0 + fl], blk0, rem0, nproma, n);
  MR_STAMP_BEGIN();
  if (flat) {
    mr_pass_any<1>(m.A, a, m.fs, nfl, mr_args(m, 0, tw1, tw2), flds + f0, blk0, rem0, nproma);
    MR_STAMP(0);
  } else {
    const int bc = m.B * m.C, pad = m.P1 - bc;
    const unsigned mbc = mr_magic((unsigned)bc);
    for (int fl = 0; fl < nfl; fl++) {
      const GridRow gr = grid_row(flds[f0 + fl], gp0, nproma);
      real2 *af = a + (long long)fl * m.fs;
      for (int l = EMI_TID; l < sz; l += EMI_NTHREADS) {
        real2 z;
        if (grid_pair_ok(gr, 2u * l)) {
          z = *(const real2 *)grid_ptr(gr, 2u * l);
        } else {
          z.x = *grid_ptr(gr, 2u * l);
          z.y = *grid_ptr(gr, 2u * l + 1);
        }
        af[l + (pad && bc > 1 ? (int)mr_div((unsigned)l, mbc) * pad : 0)] = z;
      }
    }
    EMI_SYNC();
    mr_pass_any<0>(m.A, a, m.fs, nfl, mr_args(m, 0, tw1, tw2), flds + f0, blk0, rem0, nproma);
  }
  if (m.B > 1) {
    EMI_SYNC();
    MR_STAMP(1);
    mr_pass_any<0>(m.B, a, m.fs, nfl, mr_args(m, 1, tw1, tw2), flds + f0, blk0, rem0, nproma);
    MR_STAMP(2);
  }
  if (m.C > 1) {
    EMI_SYNC();
    MR_STAMP(3);
    mr_pass_any<0>(m.C, a, m.fs, nfl, mr_args(m, 2, tw1, tw2), flds + f0, blk0, rem0, nproma);
    MR_STAMP(4);
  }
  EMI_SYNC();
  MR_STAMP(5);
  // stage 3 (FOURIER_OUT): X_k = 1/2 [ (Z_k + conj Z_{sz-k}) - i exp(-2 pi i k/n) (Z_k - conj Z_{sz-k}) ], k <= NMEN.
  // Threads run over (field, k) together; four elements per thread with all their loads first (each element of the plain loop paid a
  // table load, a dependent LDS read and a store in sequence); coefficient k sits at (k mod A) P1 + ((k / A) mod B) C + k / (AB).
  {
    const int nk = nmen + 1, ntot = nfl * nk, NT = EMI_NTHREADS;
    const unsigned mnf = mr_magic((unsigned)nfl), mA = mr_magic((unsigned)m.A), mAB = mr_magic((unsigned)(m.A * m.B));
    const real_t sc0 = (real_t)0.5 * (Lc.adj ? (real_t)1.0 : (real_t)(rw_.rw / (double)n)), racthe = (real_t)rw_.racthe;
    auto pos = [&](int k) {
      const int q1 = m.A > 1 ? (int)mr_div((unsigned)k, mA) : k;
      const int k3 = (int)mr_div((unsigned)k, mAB);
      return (k - q1 * m.A) * m.P1 + (q1 - k3 * m.B) * m.C + k3;
    };
    for (int g0 = EMI_TID; g0 < ntot; g0 += 4 * NT) {
      real2 za[4], zb[4], w4[4];
      int kk[4], ff[4];
      real_t sc[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int gi = g0 + e * NT;
        if (gi < ntot) {
          const int k = nfl > 1 ? (int)mr_div((unsigned)gi, mnf) : gi, fl = gi - k * nfl;  // field fastest: the lanes of a k write adjacent fields
          const real2 *af = a + (long long)fl * m.fs;
          kk[e] = k, ff[e] = fl;
          w4[e] = rtw[k];
          sc[e] = sc0 * fft_dir_mode_scale(flds[f0 + fl].mode, racthe);
          za[e] = af[pos(k)], zb[e] = af[pos(k == 0 ? 0 : sz - k)];
        }
      }
#if defined(EMI_MR_STAMP) && !defined(EMI_LEG_STAMP) && !defined(EMI_CPU_EMU)
      if (w4[0].x + za[0].x + zb[0].x + sc[0] == (real_t)1.2345e300) kk[0] = 0;  // wait for the loads here
      MR_STAMP(7);
#endif
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (g0 + e * NT < ntot) {
          const real2 s1 = cadd(za[e], cconj(zb[e])), d1 = csub(za[e], cconj(zb[e]));
          const real2 tt = cmuli(cmul(w4[e], d1));
          *(real2 *)(FB + (unsigned long long)(unsigned)FROW(kk[e]) * (unsigned)ldf + 2 * (f0 + ff[e])) = mk2((s1.x - tt.x) * sc[e], (s1.y - tt.y) * sc[e]);
        }
      }
      MR_STAMP(8);
    }
  }
  MR_STAMP(6);
  MR_STAMP_END();
}

EMI_KERNEL_MR(4) void k_fft_inv_mr(EmiGeomDev g, FftTabDev T, FftLaunchDev Lc, const GridFld *flds, int nfld, const real_t *FB,
                                                  int ldf, int nproma) {
  EMI_LDS_DECL;
  real2 *a = (real2 *)EMI_LDS_PTR;
  const int bid = (int)xcd_swizzle(EMI_BID, Lc.nblocks, 8);
  const int li = bid / Lc.nchunk;
  const int lat = Lc.lats[li];
  // (the per-latitude record of k_fft_dir_mr was tried here too and LOST 13 %: this kernel sits on a register-allocation edge -- 128
  // vector registers, 76 bytes of scratch, scalar registers spilled to lanes -- and the compiler placed the spills worse)
  const FftPlanDev &pl = T.plans[T.planid[lat]];
  const int fbk = pl.fbk;
  const int f0 = (bid - li * Lc.nchunk) * fbk;
  const int nfl = (nfld - f0) < fbk ? (nfld - f0) : fbk;
  const int n = pl.n, sz = pl.sz, nmen = g.nmen[lat];
  MrGeom m;
  m.A = pl.fac[0], m.B = pl.fac[1], m.C = pl.fac[2];
  m.P1 = (m.B * m.C) | 1;
  m.fs = m.A * m.P1;
  const real_t racthe = (real_t)g.racthe[lat];
  const real_t adjw = (real_t)(g.rw[lat] / (double)pl.n);  // DIR_TRANSAD only (Lc.adj)
  const int fb0 = g.fbase[lat];
  const int *frow = g.fftrow ? g.fftrow + fb0 : nullptr;
  const real2 *rtw = (const real2 *)T.rtw + pl.rtw_off;
  const real2 *tw1 = (const real2 *)T.ptw + pl.ptw_off[0], *tw2 = (const real2 *)T.ptw + pl.ptw_off[1];
  const long long gp0 = g.gpoff[lat];
  const long long blk0 = gp0 / nproma;
  const unsigned rem0 = (unsigned)(gp0 - blk0 * nproma);
  // stage 1 (FOURIER_IN + FSC): Z_k = (X_k + conj X_{sz-k}) + i w^k (X_k - conj X_{sz-k}); its conjugate to LDS, point l at
  // (l / BC) P1 + l mod BC.  Threads run over (field, pair) together, four pairs per thread with all their loads first.
  {
    const int bc = m.B * m.C, pad = (bc > 1) ? m.P1 - bc : 0;
    const unsigned mbc = mr_magic((unsigned)bc);
    const int npair = sz / 2 + 1, ntot = nfl * npair, NT = EMI_NTHREADS;
    const unsigned mnf = mr_magic((unsigned)nfl);
    if (fbk == 1) {  // (a property of the row length, not of the chunk: the same code path in every decomposition)
      // one field per workgroup (the long rows): branch-free, four pairs per thread with all their loads first (round 4, as
      // k_fft_inv_r16: behind the `k <= nmen` / FSC-mode branches of the loop below every load waits alone).  One task: the Fourier rows
      // through a buffer descriptor (k > NMEN reads zero); several tasks: clamped look-up in the exchange-order table and a select.  ONE
      // copy of the arithmetic for both, so the results do not depend on the decomposition.
      constexpr unsigned SZ2 = sizeof(real2);
      const GridFld gf = flds[f0];
      const unsigned rowb = (unsigned)ldf * (unsigned)sizeof(real_t);
      const EmiBuf b_fb = emi_buf(FB + (unsigned long long)(unsigned)fb0 * (unsigned)ldf + 2 * gf.src, (unsigned)nmen * rowb + SZ2);
      const EmiBuf b_rtw = emi_buf(rtw, (unsigned)(sz + 1) * SZ2);
      real_t fa, fb;
      fin_factors(gf.mode, racthe, fa, fb);
      const real_t fsc = Lc.adj ? adjw : (real_t)1.0;
      for (unsigned g0 = (unsigned)EMI_TID; g0 < (unsigned)npair; g0 += 4u * (unsigned)NT) {
        real2 xa[4], xb[4], w4[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const unsigned k = g0 + (unsigned)(e * NT), k2 = (unsigned)sz - k;  // k >= npair: loaded (or zero) and dropped
          if (!frow) {
            xa[e] = emi_buf_ld<real2>(b_fb, k * rowb, 0);
            xb[e] = emi_buf_ld<real2>(b_fb, k2 * rowb, 0);
          } else {
            const unsigned ka = k < (unsigned)nmen ? k : (unsigned)nmen, kb = k2 < (unsigned)nmen ? k2 : (unsigned)nmen;
            const real2 va = fin_raw(FB, frow[ka], ldf, gf.src), vb = fin_raw(FB, frow[kb], ldf, gf.src);
            xa[e] = k <= (unsigned)nmen ? va : mk2(0, 0);
            xb[e] = k2 <= (unsigned)nmen ? vb : mk2(0, 0);
          }
          w4[e] = emi_buf_ld<real2>(b_rtw, k * SZ2, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const unsigned k = g0 + (unsigned)(e * NT), k2 = (unsigned)sz - k;
          if (k < (unsigned)npair) {
            real2 zk, zk2;
            fin_pair(xa[e], xb[e], k, k2, fa, fb, fsc, cconj(w4[e]), mk2((real_t)1.0, (real_t)0.0), mk2((real_t)1.0, (real_t)0.0), zk, zk2);
            a[k + (pad ? mr_div(k, mbc) * (unsigned)pad : 0u)] = cconj(zk);
            if (k2 != k && k2 < (unsigned)sz) a[k2 + (pad ? mr_div(k2, mbc) * (unsigned)pad : 0u)] = cconj(zk2);
          }
        }
      }
    } else
    for (int g0 = EMI_TID; g0 < ntot; g0 += 4 * NT) {
      real2 xa[4], xb[4], w4[4];
      real_t fa4[4], fb4[4];
      int kk[4], ff[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int gi = g0 + e * NT;
        if (gi < ntot) {
          const int k = nfl > 1 ? (int)mr_div((unsigned)gi, mnf) : gi, fl = gi - k * nfl, k2 = sz - k;  // field fastest
          const GridFld gf = flds[f0 + fl];
          kk[e] = k, ff[e] = fl;
          xa[e] = (k <= nmen) ? fin_raw(FB, FROW(k), ldf, gf.src) : mk2(0, 0);
          xb[e] = (k2 <= nmen) ? fin_raw(FB, FROW(k2), ldf, gf.src) : mk2(0, 0);
          w4[e] = rtw[k];
          fin_factors(gf.mode, racthe, fa4[e], fb4[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (g0 + e * NT < ntot) {
          const int k = kk[e], k2 = sz - k;
          real2 *af = a + (long long)ff[e] * m.fs;
          real2 zk, zk2;
          fin_pair(xa[e], xb[e], (unsigned)k, (unsigned)k2, fa4[e], fb4[e], Lc.adj ? adjw : (real_t)1.0, cconj(w4[e]), mk2((real_t)1.0, (real_t)0.0), mk2((real_t)1.0, (real_t)0.0), zk, zk2);
          af[k + (pad ? (int)mr_div((unsigned)k, mbc) * pad : 0)] = cconj(zk);
          if (k2 != k && k2 < sz) af[k2 + (pad ? (int)mr_div((unsigned)k2, mbc) * pad : 0)] = cconj(zk2);
        }
      }
    }
  }
  EMI_SYNC();
  // stages 2 + 3 (FTINV + TRLTOG local copy): the last pass writes the grid rows -- unless NPROMA blocks cut them (or a field is not
  // 2-element aligned): those are copied from the LDS, y_i at perm[i]
  bool flat = true;
  for (int fl = 0; fl < nfl; fl++) flat = flat && mr_row_flat(flds[f0 + fl], blk0, rem0, nproma, n);
  const int last = m.C > 1 ? 2 : (m.B > 1 ? 1 : 0);
  if (last == 0 && flat)
    mr_pass_any<2>(m.A, a, m.fs, nfl, mr_args(m, 0, tw1, tw2), flds + f0, blk0, rem0, nproma);
  else
    mr_pass_any<0>(m.A, a, m.fs, nfl, mr_args(m, 0, tw1, tw2), flds + f0, blk0, rem0, nproma);
  if (m.B > 1) {
    EMI_SYNC();
    if (last == 1 && flat)
      mr_pass_any<2>(m.B, a, m.fs, nfl, mr_args(m, 1, tw1, tw2), flds + f0, blk0, rem0, nproma);
    else
      mr_pass_any<0>(m.B, a, m.fs, nfl, mr_args(m, 1, tw1, tw2), flds + f0, blk0, rem0, nproma);
  }
  if (m.C > 1) {
    EMI_SYNC();
    if (flat)
      mr_pass_any<2>(m.C, a, m.fs, nfl, mr_args(m, 2, tw1, tw2), flds + f0, blk0, rem0, nproma);
    else
      mr_pass_any<0>(m.C, a, m.fs, nfl, mr_args(m, 2, tw1, tw2), flds + f0, blk0, rem0, nproma);
  }
  if (!flat) {
    EMI_SYNC();
    const unsigned short *perm = T.perm + pl.perm_off;
    for (int fl = 0; fl < nfl; fl++) {
      const GridRow gr = grid_row(flds[f0 + fl], gp0, nproma);
      const real2 *af = a + (long long)fl * m.fs;
      for (int i = EMI_TID; i < sz; i += EMI_NTHREADS) {
        const real2 y = af[perm[i]];
        if (grid_pair_ok(gr, 2u * i)) {
          *(real2 *)grid_ptr(gr, 2u * i) = mk2(y.x, -y.y);
        } else {
          *grid_ptr(gr, 2u * i) = y.x;
          *grid_ptr(gr, 2u * i + 1) = -y.y;
        }
      }
    }
  }
}
