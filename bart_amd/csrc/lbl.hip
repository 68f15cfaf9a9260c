// Line-by-line Voigt extinction on the GPU (gather form: one lane per output
// wavenumber, the tile's window of the sorted line list staged through LDS in
// chunks; no atomics, fixed summation order).
//
// Per line j of isotope i at a layer state (T, p, composition):
//   S_j   = SIGCTE gf_j scale_i exp(-EXPCTE E_j / T) (1 - exp(-EXPCTE nu_j / T))
//   aD    = nu_j sqrt(2 ln2 k T / m_i) / c        (Doppler HWHM, scripts/broadening.py:143)
//   aL    = sqrt(2) / (c sqrt(pi k T)) p sum_{c in H2,He} q_c ((d+d_c)/2)^2 sqrt(1/m_i + 1/m_c)
//                                                 (Lorentz HWHM, scripts/broadening.py:121-127)
//   e(nu) += S_j sqrt(ln2/pi)/aD  K(sqrt(ln2)|nu-nu_j|/aD, sqrt(ln2) aL/aD)
// for |nu - nu_j| <= nwidth max(aD, aL) and S_j >= ethresh max_j S_j (per state
// and database).  scale_i = n_i / Z_i(T) for extinction (cm-1) or
// ratio_i / (Z_i(T) m_mol AMU) for opacity per gram of the molecule (cm2/g).
#include "lbl.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "engine.hpp"
#include "imw_tab.hpp"
#include "voigt_coef.hpp"

namespace bartrt {

constexpr double kSIGCTE = 8.852821681767784e-13;  // pi e^2 / (m_e c^2), cm
constexpr double kEXPCTE = kH * kLS / kKB;          // h c / k_B, cm K
constexpr double kSqrtLn2 = 0.8325546111576977;
constexpr double kInvSqrtPi = 0.5641895835477563;

// a * b rounded, then added: never fused (the header's __dmul_rn / __dadd_rn are
// plain operators the compiler may contract again after inlining)
__device__ __forceinline__ double mul_rounded(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double add_rounded(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}

// |z| >= 100 (almost every line-point pair of a pressure-broadened layer): three terms
// of voigt_k's asymptotic series in real arithmetic.  With 1/z = c - i s, C = c^2, S = s^2:
//   sqrt(pi) K = s [1 + (3 C - S)/2 + 3/4 (5 C^2 - 10 C S + S^2)]   (<= 1.4e-11)
// and, as C + S = 1/|z|^2 =: v, a polynomial in C alone,
//   = s [(1 - v/2 + 3 v^2/4) + (2 - 9 v) C + 12 C^2]:  12 operations after the reciprocal instead of 17.
__device__ __forceinline__ double voigt_far(double x2, double y, double r2) {
#pragma clang fp contract(off)
  const double v = rcp_n1(r2);                        // reciprocal + Newton: no IEEE divide per pair
  const double C = x2 * (v * v);
  const double t1 = fma(-9.0, v, 2.0), t0 = fma(fma(0.75, v, -0.5), v, 1.0);
  return (kInvSqrtPi * (y * v)) * fma(fma(12.0, C, t1), C, t0);
}

// ---------------------------------------------------------------------------
// Small y, |z| < 8 (Doppler cores: nine in ten core samples of a hot-Jupiter column have y <= 0.1):
// Taylor expansion of w about the real axis,
//     w(x + i y) = sum_n a_n,   a_n = (i y)^n w^(n)(x) / n!,
//     a_0 = w(x) = exp(-x^2) + i D(x),          a_1 = i y (2 i / sqrt(pi) - 2 x w(x)),
//     a_(n+1) = (2 y^2 a_(n-1) - 2 i x y a_n) / (n + 1)            (from w' = -2 z w + 2 i / sqrt(pi)),
// K = sum Re a_n.  exp(-x^2) is the kernels' own exp; D(x) = Im w(x) = 2 F(x) / sqrt(pi) (Dawson's integral)
// comes from a table of degree-7 polynomials on pieces of width 1/16 (imw_tab.hpp), copied into LDS by every
// kernel that evaluates the function.  The lanes of a step are points of one line a few pieces apart, 64
// different rows: a row is 48 bytes (four doubles, four floats), so that neighbouring rows start on different
// banks -- 64-byte rows cost 1.2e10 bank-conflict cycles per spectrum of config 5, these 1.6e9.
// 7 operations per order; the order follows y alone (voigt_order: 2 for y <= 6e-6 ... 10 for y <= 0.13, each
// chosen so that the first omitted term is below 3e-12 of K over the whole of |z| < 8): <= 2.5e-12 relative
// everywhere, against 1e-10 of the rational approximation it replaces there, at 40 - 93 VALU operations
// per sample against 96.  (The same expansion for the wings, 8 <= |z| < 17 with exp(-x^2) dropped, was built
// and measured: the table index and the cancellation-safe form cost what the eleven-term series costs, 40.)
constexpr double kTwoInvSqrtPi = 1.1283791670955126;
constexpr double kTaylorYmax = 0.13;
constexpr double kSeriesR2 = 17.0 * 17.0;   // from here six terms of the asymptotic series do

// Order of the expansion for a line of this y; 0: not small (Weideman's approximation, asymptotic series).
__device__ double g_taylor_ymax = kTaylorYmax;   // BARTRT_VOIGT_TAYLOR=0 sets it to 0: no expansion for any y
__device__ __forceinline__ int voigt_order(double y) {
  if (!(y <= g_taylor_ymax)) return 0;
  return 2 + (y > 6.0e-6) + (y > 5.0e-4) + (y > 3.0e-3) + (y > 9.5e-3) + (y > 2.2e-2) + (y > 4.0e-2) +
         (y > 6.5e-2) + (y > 9.5e-2);
}

// The table in LDS (kImwDoubles doubles at `tab`, rows of 48 bytes); callers put a barrier after it.
constexpr int kImwDoubles = kImwPieces * (int)(sizeof(ImwRow) / sizeof(double));
static_assert(sizeof(ImwRow) == 48, "ImwRow: four doubles and four floats");
__device__ __forceinline__ void load_imw_table(double *tab) {
  const double *src = reinterpret_cast<const double *>(kImwTab);
  for (int i = threadIdx.x; i < kImwDoubles; i += blockDim.x) tab[i] = src[i];
}

// D at local coordinate t of row i
__device__ __forceinline__ double imw_row(const double *tab, int i, double t) {
#pragma clang fp contract(off)
  // (an LDS pointer and a 32-bit row offset: through the generic pointer the row address is a 64-bit multiply-add)
  typedef float f4_t __attribute__((ext_vector_type(4)));
  typedef double d2_t __attribute__((ext_vector_type(2)));
  const unsigned base = (unsigned)(unsigned long long)tab;   // the low half of a generic pointer into LDS is its LDS address
  const unsigned row = base + __umul24((unsigned)i, (unsigned)sizeof(ImwRow));
  const f4_t hi = *(const f4_t __attribute__((address_space(3))) *)(size_t)(row + 32);
  const d2_t c01 = *(const d2_t __attribute__((address_space(3))) *)(size_t)(row),
             c23 = *(const d2_t __attribute__((address_space(3))) *)(size_t)(row + 16);
  double D = fma((double)hi.w, t, (double)hi.z);
  D = fma(D, t, (double)hi.y);
  D = fma(D, t, (double)hi.x);
  D = fma(D, t, c23.y);
  D = fma(D, t, c23.x);
  D = fma(D, t, c01.y);
  return fma(D, t, c01.x);
}

// exp_rt's arithmetic (kernels.hpp) with the Horner steps spelled as three-operand v_fma_f64: in these kernels
// the nine coefficients live in VGPRs (no scalar registers left), and the compiler's two-address form
// (v_fmac after a copy of the coefficient) costs a v_mov per step.  Same operations, same bits.
__device__ __forceinline__ double fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ double exp_rt_fma3(double x) {
  const double t = fma(x, 1.4426950408889634074, kExpShift);
  const double r = fma(t - kExpShift, -6.93147180559945286227e-01, x);
  constexpr double cf[9] = {2.4867870179687727e-05, 0.00019841224599656011, 0.0013888839110572009,
                            0.0083333333442029804,  0.041666666786265731,   0.16666666666662586,
                            0.49999999999955108,    1.0,                    1.0};
  double q = 2.7617564785876086e-06;
#pragma unroll
  for (int j = 0; j < 9; j++) q = fma3(q, r, cf[j]);
  return exp_scale(q, t);
}

constexpr double kInvN[12] = {0.0, 1.0, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8, 1.0 / 9, 1.0 / 10, 1.0 / 11};

// |z| < 8: straight-line code per order (a wave-uniform order takes one scalar branch; where the lanes of a
// wave hold different lines -- lbl_accumulate_pairs -- the orders present run one after the other, which
// measured faster than predicating the terms of one unrolled loop: 3.5 against 3.9 ms on config 5).
template <int ORDER>
__device__ __forceinline__ double voigt_taylor_n(double x, double y, double x2, const double *tab) {
#pragma clang fp contract(off)
  const double xs = x * (double)(kImwPieces / 8);
  const int i = min((int)xs, kImwPieces - 1);
  const double D = imw_row(tab, i, xs - ((double)i + 0.5));
  const double E = exp_rt_fma3(-x2);
  const double al = (x + x) * y, be = (y + y) * y;
  // (p0, q0) and (p1, q1) hold Re, Im of two successive terms; each step overwrites the older one
  double p0 = E, q0 = D;
  double p1 = fma(al, D, -(kTwoInvSqrtPi * y)), q1 = -(al * E);
  double K = E + p1;
#pragma unroll
  for (int n = 1; n < ORDER; n += 2) {
    const double pe = fma(al, q1, be * p0) * kInvN[n + 1];
    if (n + 1 < ORDER) q0 = fma(-al, p1, be * q0) * kInvN[n + 1];     // the last term's imaginary part is not needed
    p0 = pe;
    K = K + p0;
    if (n + 1 < ORDER) {
      const double po = fma(al, q0, be * p1) * kInvN[n + 2];
      if (n + 2 < ORDER) q1 = fma(-al, p0, be * q1) * kInvN[n + 2];
      p1 = po;
      K = K + p1;
    }
  }
  return K;
}

__device__ __forceinline__ double voigt_taylor(double x, double y, double x2, int order, const double *tab) {
  switch (order) {
    case 2: return voigt_taylor_n<2>(x, y, x2, tab);
    case 3: return voigt_taylor_n<3>(x, y, x2, tab);
    case 4: return voigt_taylor_n<4>(x, y, x2, tab);
    case 5: return voigt_taylor_n<5>(x, y, x2, tab);
    case 6: return voigt_taylor_n<6>(x, y, x2, tab);
    case 7: return voigt_taylor_n<7>(x, y, x2, tab);
    case 8: return voigt_taylor_n<8>(x, y, x2, tab);
    case 9: return voigt_taylor_n<9>(x, y, x2, tab);
    default: return voigt_taylor_n<10>(x, y, x2, tab);
  }
}

// s(t) = sum_k (2k-1)!!/2^k t^k, NT terms, for complex t = tr + i ti: synthetic division by the argument's real
// quadratic (see voigt_k); the value is sr + i si.
template <int NT>
__device__ __forceinline__ void asym_series(double tr, double ti, double &sr, double &si) {
#pragma clang fp contract(off)
  constexpr double c[11] = {1.0, 0.5, 0.75, 1.875, 6.5625, 29.53125, 162.421875, 1055.7421875, 7918.06640625,
                            67303.564453125, 639383.8623046875};
  const double p = tr + tr, q = fma(tr, tr, ti * ti);
  double a = c[NT - 1], b = c[NT - 2];
#pragma unroll
  for (int k = NT - 3; k >= 0; k--) {
    const double an = fma(p, a, b);
    b = fma(-q, a, c[k]);
    a = an;
  }
  sr = fma(a, tr, b);
  si = a * ti;
}

// Re w(x + i y), x >= 0, y > 0.  Contraction is switched off inside and every
// fused multiply-add is written out: the function is inlined into kernels with
// different surroundings, and which products the compiler fuses must not depend
// on them (a layer state can be summed by either accumulation kernel).  Which
// branch a point takes depends on (x, y) alone, never on the other lanes.
//
// The asymptotic series and Weideman's approximation are polynomials with REAL
// coefficients in a complex argument (t = 1/z^2, resp. Weideman's Z).  Those are
// evaluated by synthetic division by the argument's real quadratic
// u^2 - 2 Re(u) u + |u|^2  (Knuth, TAOCP 4.6.4): two real FMAs per coefficient
// instead of the five operations of a complex Horner step; the remainder a u + b is
// the value.  `order` hands in voigt_order(y) where the caller has it per line (it is
// called only for |z| < 8); `tab` is the LDS copy of the Im w table.
template <class OrderFn>
__device__ __forceinline__ double voigt_k(double x, double y, const double *tab, OrderFn order) {
#pragma clang fp contract(off)
  const double x2 = x * x, y2 = y * y;
  const double r2 = x2 + y2;
  if (r2 >= 1.0e4) return voigt_far(x2, y, r2);
  if (r2 >= 64.0) {
    // asymptotic series  w = i/(sqrt(pi) z) * s(1/z^2): eleven terms from |z| = 8 (<= 2.6e-12 there; the
    // Laplace continued fraction with eight levels that stood here gave 1.7e-12 at eight reciprocals), six
    // from |z| = 17 (<= 3e-13)
    const double inv = rcp_n1(r2), inv2 = inv * inv;
    const double tr = (x2 - y2) * inv2, ti = (-2.0 * x) * y * inv2;  // t = 1/z^2
    double sr, si;
    if (r2 >= kSeriesR2) asym_series<6>(tr, ti, sr, si);
    else asym_series<11>(tr, ti, sr, si);
    // Re[i s / z] = (y s_r - x s_i) / |z|^2
    return kInvSqrtPi * fma(y, sr, -(x * si)) * inv;
  }
  const int ord = order();
  if (ord) return voigt_taylor(x, y, x2, ord, tab);
  // y > 0.13: Weideman N = 36: Z = ((L - y) + i x) / ((L + y) - i x)
  const double ar = kWeidL - y, br = kWeidL + y;
  const double den = rcp_core(fma(br, br, x2));
  const double Zr = fma(ar, br, -x2) * den, Zi = fma(x, br, ar * x) * den;
  const double p = Zr + Zr, q = fma(Zr, Zr, Zi * Zi);
  double a = kWeidA[0], b = kWeidA[1];
#pragma unroll
  for (int k = 2; k < kWeidN; k++) {
    const double an = fma(p, a, b);
    b = fma(-q, a, kWeidA[k]);
    a = an;
  }
  const double pr = fma(a, Zr, b), pi = a * Zi;
  // 1/(L - iz) = (br + i x) den ; its square
  const double qr = br * den, qi = x * den;
  const double q2r = fma(qr, qr, -(qi * qi)), q2i = 2.0 * qr * qi;
  return fma(2.0, fma(pr, q2r, -(pi * q2i)), kInvSqrtPi * qr);
}
__device__ __forceinline__ double voigt_k(double x, double y, const double *tab) {
  return voigt_k(x, y, tab, [&] { return voigt_order(y); });
}

// Diagnostics (bartrt_voigt): the kernels' Voigt function on n (x, y) pairs.
__global__ void voigt_probe(const double *x, const double *y, double *k, long n) {
  __shared__ double s_tab[kImwDoubles];
  load_imw_table(s_tab);
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) k[i] = voigt_k(x[i], y[i], s_tab);
}

void lbl_voigt_probe(const double *x, const double *y, double *k, long n) {
  if (n <= 0) return;
  double *d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(double) * 3 * n));
  HIPCHK(hipMemcpy(d, x, sizeof(double) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + n, y, sizeof(double) * n, hipMemcpyHostToDevice));
  voigt_probe<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(d, d + n, d + 2 * n, n);
  hipError_t err = hipDeviceSynchronize();
  if (err == hipSuccess) err = hipMemcpy(k, d + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIPCHK(err);
}

struct StateArgs {
  int L, S, nstate, table_mode, iH2, iHe, Nt, layer0;
  const double *prof;    // extinction mode: [nw][(S+1)][L]
  const double *abund0;  // table mode: base abundances [L][S]
  const double *press;   // [L] barye
  const double *mass;    // [S] amu
  const double *diam;    // [S] cm
  const double *tgrid;   // table mode temperatures
  double *state;         // [nstate][2 + 3*niso]: T, oversampling factor dv, then (dopfac, alphaL, scale) per isotope
  int *dvmax;            // largest dv of the launch (sizes lbl_accumulate_fine's LDS), or null
};

// One workgroup per state, lanes over isotopes.
__global__ void lbl_states(StateArgs a, LblDev d) {
  const int st = blockIdx.x, L = a.L, S = a.S;
  double T, p;
  const double *q;  // mixing ratios with stride qs
  size_t qs;
  int l;
  if (a.table_mode) {
    l = a.layer0 + st / a.Nt;
    T = a.tgrid[st % a.Nt];
    q = a.abund0 + (size_t)l * S; qs = 1;
  } else {
    const int w = st / L;
    l = st % L;
    const double *pr = a.prof + (size_t)w * (S + 1) * L;
    T = pr[l];
    q = pr + L + l; qs = L;
  }
  p = a.press[l];
  double *out = a.state + (size_t)st * (2 + 3 * d.niso);
  const int i = threadIdx.x;
  const bool mine = i < d.niso;
  const int g = d.iso_group[mine ? i : 0];
  const double mi = d.iso_mass[mine ? i : 0] * kAMU;
  const double dop = sqrt(2.0 * 0.6931471805599453 * kKB * T / mi) / kLS;
  double sum = 0.0;
  const double dm = d.gdiam[g];
  if (a.iH2 >= 0) {
    const double dd = 0.5 * (dm + a.diam[a.iH2]);
    sum += q[(size_t)a.iH2 * qs] * dd * dd * sqrt(1.0 / mi + 1.0 / (a.mass[a.iH2] * kAMU));
  }
  if (a.iHe >= 0) {
    const double dd = 0.5 * (dm + a.diam[a.iHe]);
    sum += q[(size_t)a.iHe * qs] * dd * dd * sqrt(1.0 / mi + 1.0 / (a.mass[a.iHe] * kAMU));
  }
  const double aL = sqrt(2.0) / (kLS * sqrt(kPI * kKB * T)) * p * sum;
  {
    // oversampling factor of this state (one wave per state: a 64-lane minimum of the
    // isotopes' line half-widths, Doppler at the low end of the full grid)
    double wmin = mine ? fmax(d.wn_first * dop, aL) : 1e300;
    for (int o = 32; o > 0; o >>= 1) wmin = fmin(wmin, __shfl_xor(wmin, o));
    if (threadIdx.x == 0) {
      int dv = 1;
      if (d.osamp > 1) {
        dv = d.osamp;
        if (d.osamp_rule == 0)
          for (int k = 0; k < d.ndiv; k++)
            if (d.wndelt / d.odiv[k] <= 0.5 * wmin) { dv = d.odiv[k]; break; }
      }
      out[0] = T;
      out[1] = (double)dv;
      if (a.dvmax && dv > 1) atomicMax(a.dvmax, dv);
    }
  }
  if (!mine) return;
  // partition function: linear in T on the database grid, clamped
  const double *zt = d.ztemp + d.iso_toff[i], *zz = d.ztab + d.iso_zoff[i];
  const int nt = d.iso_nt[i];
  double Z;
  if (nt == 1 || T <= zt[0]) Z = zz[0];
  else if (T >= zt[nt - 1]) Z = zz[nt - 1];
  else {
    int j = 0;
    while (j < nt - 2 && zt[j + 1] <= T) j++;
    const double f = (T - zt[j]) / (zt[j + 1] - zt[j]);
    Z = zz[j] * (1.0 - f) + zz[j + 1] * f;
  }
  double scale;
  if (a.table_mode) {
    const int sp = d.gspecies[g];
    scale = d.iso_ratio[i] / (Z * a.mass[sp] * kAMU);
  } else {
    const double nd = p / (kKB * T);
    scale = d.iso_ratio[i] * q[(size_t)d.gspecies[g] * qs] * nd / Z;
  }
  out[2 + 3 * i] = dop;
  out[3 + 3 * i] = aL;
  out[4 + 3 * i] = kSIGCTE * scale;
}

// invT = 1 / T.  exp_core: the arguments are <= 0; far below -708 the line is
// negligible at any threshold
__device__ inline double line_strength(double gf, double elow, double nu0, double scale, double invT) {
#pragma clang fp contract(off)  // same bits in every kernel that inlines it (see voigt_k)
  const double a = fmax(-kEXPCTE * elow * invT, kExpMin), b = fmax(-kEXPCTE * nu0 * invT, kExpMin);
  return gf * scale * exp_core(a) * (1.0 - exp_core(b));
}

// max line strength per (state, group); positive doubles order like their bits
__global__ __launch_bounds__(256) void lbl_smax(LblDev d, const double *state, double *smax,
                                                int nstate) {
  const int st = blockIdx.y, g = blockIdx.z;
  const double *sv = state + (size_t)st * (2 + 3 * d.niso);
  const double invT = 1.0 / sv[0];
  double m = 0.0;
  for (long j = d.gstart[g] + blockIdx.x * blockDim.x + threadIdx.x; j < d.gend[g];
       j += (long)gridDim.x * blockDim.x) {
    const int i = d.liso[j];
    m = fmax(m, line_strength(d.gf[j], d.elow[j], d.nu0[j], sv[4 + 3 * i], invT));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.0)
    atomicMax(reinterpret_cast<unsigned long long *>(smax + (size_t)st * d.ngroup + g),
              (unsigned long long)__double_as_longlong(m));
}

struct AccArgs {
  int W, nstate, per_group;   // per_group: out[state][g][W] else out[state][W] summed over groups
  int pair_reach;             // states whose widest cut spans at most this many points go to lbl_accumulate_pairs
  int mid_reach;              // ... at most this many (and more than pair_reach): the line-major walk of lbl_accumulate_fine; 0: off
  const double *wn;
  const double *state, *smax;
  double *out;
  // width-grid mode (LblDev::voigt_grid; Lbl's d_g* arrays)
  const int *ginfo, *gK;
  const long *goff, *gbase;   // gbase[state]: first double of the state's profiles in ptab
  const double *ptab;
};

// ---------------------------------------------------------------------------
// Window of group g's sorted line list for the points [nu_a, nu_b] of a state:
// centres within the widest cut of the group's isotopes, rounded out to the
// 1 cm-1 bucket index (every lane does the same two scalar loads; a single
// lane's binary search over global memory would hold the others at a barrier
// for ~40 dependent loads).  The exact per-line cut is applied by the callers.
__device__ __forceinline__ void line_window(const LblDev &d, const double *sv, int g, double nu_a,
                                            double nu_b, long &j0, long &j1, double &cmax) {
  cmax = 0.0;
  for (int k = 0; k < d.niso; k++)
    if (d.iso_group[k] == g)
      cmax = fmax(cmax, d.nwidth * fmax(sv[3 + 3 * k], (nu_b + 1.0) * sv[2 + 3 * k] * 1.001));
  cmax *= 1.01;
  int b0 = (int)floor((nu_a - cmax - d.bmin) / d.bstep), b1 = (int)floor((nu_b + cmax - d.bmin) / d.bstep) + 1;
  b0 = b0 < 0 ? 0 : (b0 > d.nbucket ? d.nbucket : b0);
  b1 = b1 < 0 ? 0 : (b1 > d.nbucket ? d.nbucket : b1);
  j0 = d.bucket[d.boff[g] + b0];
  j1 = d.bucket[d.boff[g] + b1];
}

// Widest cut of any isotope at this state / smallest point spacing of the tile:
// how many points a line can reach on either side.  Decides which of the two
// accumulation kernels owns the state (both are launched over all states).
__device__ __forceinline__ bool narrow_state(const LblDev &d, const AccArgs &a, const double *sv, int tile0) {
  const int ilast = min(tile0 + 255, a.W - 1);
  if (ilast <= tile0) return false;
  const double dnu = fmin(a.wn[tile0 + 1] - a.wn[tile0], a.wn[ilast] - a.wn[ilast - 1]);
  double cany = 0.0;
  for (int k = 0; k < d.niso; k++)
    cany = fmax(cany, d.nwidth * fmax(sv[3 + 3 * k], (a.wn[ilast] + 1.0) * sv[2 + 3 * k] * 1.011));
  return cany <= a.pair_reach * dnu;
}

// Moderately broad states evaluated on the output points (no oversampling): the widest cut of any isotope, taken at
// the top of the FULL grid, reaches at most mid_reach points either side.  With lane = point (lbl_accumulate) a line
// of +-48 points keeps 38 % of a 256-point tile's lanes inside its cut; the line-major walk of lbl_accumulate_fine on
// 64-point sub-tiles keeps nearly all of them (round 4).  A function of the state alone; where a tile of such a state
// is narrow enough for the pair kernel (narrow_state, by 256-point tile), the pair kernel keeps it.
__device__ __forceinline__ bool mid_state(const LblDev &d, const AccArgs &a, const double *sv) {
  if (a.mid_reach <= 0 || sv[1] > 1.0) return false;
  const double wn_last = d.wn_first + (double)(d.wfull - 1) * d.wndelt;
  double cany = 0.0;
  for (int k = 0; k < d.niso; k++)
    cany = fmax(cany, d.nwidth * fmax(sv[3 + 3 * k], (wn_last + 1.0) * sv[2 + 3 * k] * 1.011));
  return cany <= a.mid_reach * d.wndelt;
}

// One line's staged record, or cut < 0 for a line below the strength threshold.
struct LineRec { double nu0, amp, xs, y, cut; };
__device__ __forceinline__ LineRec stage_line(const LblDev &d, const double *sv, double invT, double thresh, long j) {
#pragma clang fp contract(off)
  LineRec r{1e300, 0.0, 0.0, 1.0, -1.0};
  const int k = d.liso[j];
  const double n0 = d.nu0[j];
  const double Sj = line_strength(d.gf[j], d.elow[j], n0, sv[4 + 3 * k], invT);
  if (Sj >= thresh && Sj > 0.0) {
    const double aD = n0 * sv[2 + 3 * k], aL = sv[3 + 3 * k];
    r.nu0 = n0;
    r.cut = d.nwidth * fmax(aD, aL);
    r.xs = kSqrtLn2 / aD;
    r.amp = Sj * kSqrtLn2 * kInvSqrtPi / aD;
    r.y = aL * r.xs;
  }
  return r;
}

// Broad states (pressure-broadened layers: a line reaches many points).  One
// workgroup per (256-point tile, state); lane = point.  The window of the line
// list is staged through LDS 256 lines at a time; lines below the strength
// threshold (two thirds of a typical list at ethresh 1e-6) are dropped while
// staging (ordered compaction: ballot + prefix), so no lane ever tests them.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void lbl_accumulate(LblDev d, AccArgs a) {
  __shared__ double s_nu0[256], s_amp[256], s_xs[256], s_y[256], s_cut[256];
  __shared__ double s_tab[kImwDoubles];
  __shared__ int s_ord[256];
  __shared__ int s_wcount[4];
  const int st = blockIdx.y;
  const int tile0 = blockIdx.x * 256;
  const double *sv = a.state + (size_t)st * (2 + 3 * d.niso);
  if (sv[1] > 1.0) return;                    // an oversampled state: lbl_accumulate_fine owns it
  if (narrow_state(d, a, sv, tile0)) return;  // lbl_accumulate_pairs owns it
  if (mid_state(d, a, sv)) return;            // moderately broad: lbl_accumulate_fine's line-major walk owns it
  load_imw_table(s_tab);
  __syncthreads();
  const double invT = 1.0 / sv[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = tile0 + threadIdx.x;
  const double nu = a.wn[min(i, a.W - 1)];
  const double nu_a = a.wn[tile0], nu_b = a.wn[min(tile0 + 255, a.W - 1)];
  const int g_lo = a.per_group ? blockIdx.z : 0, g_hi = a.per_group ? blockIdx.z + 1 : d.ngroup;
  double acc = 0.0;
  for (int g = g_lo; g < g_hi; g++) {
    long j0, j1;
    double cmax;
    line_window(d, sv, g, nu_a, nu_b, j0, j1, cmax);
    const double thresh = d.ethresh * a.smax[(size_t)st * d.ngroup + g];
    for (long base = j0; base < j1; base += 256) {
      const long j = base + threadIdx.x;
      LineRec r{1e300, 0.0, 0.0, 1.0, -1.0};
      if (j < j1) r = stage_line(d, sv, invT, thresh, j);
      // ordered compaction of the kept lines
      const unsigned long long keep = __ballot(r.cut >= 0.0);
      if (lane == 0) s_wcount[wave] = __popcll(keep);
      __syncthreads();
      int pos = __popcll(keep & ((1ull << lane) - 1ull));
      for (int w = 0; w < wave; w++) pos += s_wcount[w];
      const int cnt = s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3];
      if (r.cut >= 0.0) {
        s_nu0[pos] = r.nu0; s_amp[pos] = r.amp; s_xs[pos] = r.xs; s_y[pos] = r.y; s_cut[pos] = r.cut;
        s_ord[pos] = voigt_order(r.y);
      }
      __syncthreads();
      // product rounded on its own (no fma into the sum): the pair kernel stores
      // the same product before adding it, and a state may change owner between
      // two tilings of the grid -- the bits must not
      auto one = [&](int t) {
        const double dv = fabs(nu - s_nu0[t]);
        if (dv <= s_cut[t])
          acc = add_rounded(acc, mul_rounded(s_amp[t], voigt_k(dv * s_xs[t], s_y[t], s_tab, [&] { return __builtin_amdgcn_readfirstlane(s_ord[t]); })));
      };
      for (int t = 0; t < cnt; t++) one(t);
      __syncthreads();
    }
  }
  if (i < a.W) {
    if (a.per_group) a.out[((size_t)st * d.ngroup + blockIdx.z) * a.W + i] = acc;
    else a.out[(size_t)st * a.W + i] = acc;
  }
}

// Narrow and moderately broad states (every cut reaches at most kPairReach points).
// With lane = point a step would evaluate the Voigt function -- and the core of
// a Doppler line is its costly branch, a 40-term rational -- with the few lanes
// a line reaches and the rest of the wave idle, and a workgroup-wide staging
// loop would leave three of four waves waiting at its barriers (a chunk of the
// sorted list overlaps one wave's points).  Here every wave works alone on its 64
// points, no workgroup barrier:
//   A1 lane = line, 64 lines of the window at a time: stage it; the lines above the
//      strength threshold whose cut reaches the wave's span join a queue in LDS (a ring), each
//      with its range of the wave's points (binary search, once per line)
//   A2 (the queue holds 64 lines, or the window is exhausted) lane = queued line: prefix sum
//      of the range lengths -> pair offsets; the round takes the leading lines whose pairs fit
//      the LDS buffer (kPairCap), the rest stay queued
//   B  lane = (line, point) pair, 64 pairs per pass, all lanes busy: the pair's
//      contribution goes to the buffer in pair order
//   C  lane = point: adds its pairs in line order (fixed summation order, no
//      atomics), looping over the round's lines that reach any point at all, their
//      ranges broadcast from the registers of their A2 lanes
// (A lane = point gather over each point's own window of the sorted list was costed and
// dropped: the wave runs the longest of its 64 windows, about four times the mean for
// Doppler cores.)
// Ownership: states whose widest cut spans at most kPairReach points (measured on
// config 5, the 80 layers above 3 bar: 7: 4.5 ms, 12: 4.0, 20: 3.9, 31: 3.5, 48: 3.5 --
// with lane = point a line of 2 * 31 points still leaves half of a wave's lanes idle).
constexpr int kPairReach = 31;
constexpr int kPairCap = 240;                       // pairs per round (LDS buffer; four waves' scratch + the Im w table fit 40 KB)

struct PairScratch {                        // per wave
  double wnu[64];                           // the wave's points
  // queue of kept lines that reach the wave, list order: a RING of 128 slots (round 6: a round takes the ~9 leading
  // lines whose pairs fit the buffer -- 25 points per line on config 5's Doppler cores -- and moving the other ~55 up
  // the queue every round was a tenth of the kernel)
  double nu0[128], amp[128], xs[128], y[128], cut[128];
  double val[kPairCap];                     // pair contributions of the round
  unsigned short off[64];                   // pair offset of each line of the round
  unsigned char qfirst[128], qn[128];       // a queued line's first point and point count, found ONCE when it is staged
                                            // (round 6: every round searched the ranges of all 64 queued lines again)
  unsigned char ord[128];                   // voigt_order(y) of the queued lines
  unsigned char line[kPairCap];             // the line (position in the round) a pair belongs to
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void lbl_accumulate_pairs(LblDev d, AccArgs a) {
  __shared__ PairScratch s_pairs[4];
  __shared__ double s_tab[kImwDoubles];
  const int st = blockIdx.y;
  const int tile0 = blockIdx.x * 256;
  const double *sv = a.state + (size_t)st * (2 + 3 * d.niso);
  if (sv[1] > 1.0) return;                     // an oversampled state: lbl_accumulate_fine owns it
  if (!narrow_state(d, a, sv, tile0)) return;  // lbl_accumulate owns it
  load_imw_table(s_tab);
  __syncthreads();                             // the only workgroup barrier: the waves part ways here
  const double invT = 1.0 / sv[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w0 = tile0 + wave * 64;
  if (w0 >= a.W) return;
  PairScratch &ws = s_pairs[wave];
  const int i = w0 + lane;
  // points beyond the grid's end are pushed out of every line's reach
  ws.wnu[lane] = i < a.W ? a.wn[i] : 1e300;
  wave_sync();
  const double nu_a = a.wn[w0], nu_b = a.wn[min(w0 + 63, a.W - 1)];
  const int g_lo = a.per_group ? blockIdx.z : 0, g_hi = a.per_group ? blockIdx.z + 1 : d.ngroup;
  const int nmax = 2 * a.pair_reach + 3;   // points of a line incl. one of slack either side (<= kPairCap)
  double acc = 0.0;
  int pending = 0, head = 0;               // queued lines and the ring's first slot (wave-uniform)
  // Rounds B / C over the head of the queue: while it holds 64 lines (all of it when flushing)
  auto rounds = [&](bool flush) {
    while (pending >= 64 || (flush && pending > 0)) {
      const int nq = min(pending, 64);
      // ---- A2  lane = queued line: its range of the wave's points, as found when it was staged
      const int slot = (head + lane) & 127;
      int first = 0, n = 0;
      if (lane < nq) { first = ws.qfirst[slot]; n = ws.qn[slot]; }
      int incl = n;  // inclusive prefix sum over the lanes
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      // the round takes the leading lines whose pairs fit the buffer (at least one: a
      // line has at most nmax <= kPairCap); the others stay at the head of the queue
      const unsigned long long fit = __ballot(incl <= kPairCap);
      const int m = min(~fit ? __builtin_ctzll(~fit) : 64, nq);
      if (lane >= m) n = 0;
      const int total = __builtin_amdgcn_readlane(incl, m - 1);
      const int offv = incl - n;
      if (total > 0) {
        ws.off[lane] = (unsigned short)offv;
        for (int k = 0; k < n; k++) ws.line[offv + k] = (unsigned char)lane;
        wave_sync();
        // ---- B
        for (int p = lane; p < total; p += 64) {
          const int x = ws.line[p], xs_ = (head + x) & 127;
          const double dv = fabs(ws.wnu[ws.qfirst[xs_] + (p - ws.off[x])] - ws.nu0[xs_]);
          ws.val[p] = dv <= ws.cut[xs_]
                          ? mul_rounded(ws.amp[xs_], voigt_k(dv * ws.xs[xs_], ws.y[xs_], s_tab, [&] { return (int)ws.ord[xs_]; }))
                          : 0.0;
        }
        wave_sync();
        // ---- C  (a line's range comes from the registers of its A2 lane, so a point no
        // line reaches reads nothing from LDS)
        for (unsigned long long live = __ballot(n > 0); live; live &= live - 1) {
          const int t = __builtin_ctzll(live);
          const unsigned k = (unsigned)(lane - __builtin_amdgcn_readlane(first, t));
          if (k < (unsigned)__builtin_amdgcn_readlane(n, t))
            acc = add_rounded(acc, ws.val[__builtin_amdgcn_readlane(offv, t) + k]);
        }
        wave_sync();   // (the next round's fill of `line` / `off` must not overtake this round's readers)
      }
      // the ring moves on by the m lines taken
      head = (head + m) & 127;
      pending -= m;
    }
  };
  for (int g = g_lo; g < g_hi; g++) {
    long j0, j1;
    double cmax;
    line_window(d, sv, g, nu_a, nu_b, j0, j1, cmax);
    const double thresh = d.ethresh * a.smax[(size_t)st * d.ngroup + g];
    for (long base = j0; base < j1; base += 64) {
      // ---- A1  lane = line: stage it; the lines above the strength threshold whose cut
      // reaches the wave's span join the queue (at most 63 wait there) with their range of the wave's points
      const long j = base + lane;
      LineRec r{1e300, 0.0, 0.0, 1.0, -1.0};
      if (j < j1) r = stage_line(d, sv, invT, thresh, j);
      const bool in = r.cut >= 0.0 && r.nu0 + r.cut >= nu_a && r.nu0 - r.cut <= nu_b;
      const unsigned long long keep = __ballot(in);
      if (!keep) continue;
      if (in) {
        const int pos = (head + pending + __popcll(keep & ((1ull << lane) - 1ull))) & 127;
        ws.nu0[pos] = r.nu0; ws.amp[pos] = r.amp; ws.xs[pos] = r.xs; ws.y[pos] = r.y; ws.cut[pos] = r.cut;
        ws.ord[pos] = (unsigned char)voigt_order(r.y);
        const double lo = r.nu0 - r.cut, hi = r.nu0 + r.cut;
        int b = 0, e = 64;  // b -> first point >= lo
        while (b < e) { const int m = (b + e) >> 1; if (ws.wnu[m] < lo) b = m + 1; else e = m; }
        int c = b;          // c -> first point > hi
        e = 64;
        while (c < e) { const int m = (c + e) >> 1; if (ws.wnu[m] <= hi) c = m + 1; else e = m; }
        int first = 0, n = 0;
        if (c > b) {
          // one point of slack either side: the exact |nu - nu0| <= cut test is per pair
          first = b > 0 ? b - 1 : 0;
          n = min((c < 64 ? c + 1 : 64) - first, nmax);
        }
        ws.qfirst[pos] = (unsigned char)first;
        ws.qn[pos] = (unsigned char)n;
      }
      pending += __popcll(keep);
      wave_sync();
      rounds(false);
    }
  }
  rounds(true);
  if (i < a.W) {
    if (a.per_group) a.out[((size_t)st * d.ngroup + blockIdx.z) * a.W + i] = acc;
    else a.out[(size_t)st * a.W + i] = acc;
  }
}

// Oversampled states (cfg `wnosamp` > 1 and lines narrower than twice the output
// spacing; LblDev, "Sampling"): the line sums are evaluated on the state's fine grid,
// dv points per output spacing, and reduced to the output points -- an odd dv averages
// the dv fine points centred on the output point, an even dv takes dv + 1 with the
// two ends at half weight; the first / last point of the full grid use the half of the
// window that lies on the grid, normalised by its own weights.  On its fine grid every
// state is broad: dv is chosen so that a half-width spans two fine points or more, and
// a line is cut at nwidth half-widths -- 80 fine points and up.
// Line-major: ONE wave per sub-tile of about kFineTarget fine points, whose sums live
// in LDS; the wave stages 64 lines at a time (lane = line: strength, cut, and the
// line's index ranges on the fine grid), then takes the kept lines one after the other
// with lane = fine point of THAT line: first both wings packed into one run of lanes,
// then the core.  (Round 2's first form, lane = point looping over the lines, left the
// lanes beyond a line's cut idle and, where a wave straddles |z| = 8, ran both branches
// of the Voigt function for all of them: 67 against 55 ms on config 5.  Here the lanes
// of a step are consecutive points of one line, on one side of |z| = 8 up to the
// rounding of the range ends -- the branch itself stays per point.)  Every fine point
// adds its lines in list order, whatever the tiling.  No workgroup barrier anywhere;
// smaller sub-tiles (more resident waves) measured faster than larger ones (512: 55 ms,
// 1 024: 56, 2 048: 63, 4 096: 92).
constexpr int kFineTarget = 512;
constexpr int kMidReach = 0;   // (0: off until measured; BARTRT_MID_REACH) see mid_state
// Doubles of LDS one wave of lbl_accumulate_fine needs: the fine sums, a staged batch of 64 lines (five
// doubles, four range ints and the Voigt order each).
__host__ __device__ inline size_t fine_wave_doubles(int nfmax) { return (size_t)nfmax + 5 * 64 + (4 * 64 + 64) / 2; }

__global__ __launch_bounds__(1024) void lbl_accumulate_fine(LblDev d, AccArgs a, int nfmax, int ftarget) {
  extern __shared__ double s_dyn[];
  // the workgroup's waves (1 or 4, the host decides by nfmax) share the Im w table and nothing else
  double *s_tab = s_dyn;
  const int st = blockIdx.y;
  const double *sv = a.state + (size_t)st * (2 + 3 * d.niso);
  // states evaluated on the output points belong to the other two kernels, except the moderately broad ones
  const bool mid = mid_state(d, a, sv);
  if (!(sv[1] > 1.0) && !mid) return;
  load_imw_table(s_tab);
  __syncthreads();                        // the only workgroup barrier
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile0 = (blockIdx.x * (int)(blockDim.x >> 6) + wave) * 64;
  if (tile0 >= a.W) return;
  if (mid && narrow_state(d, a, sv, tile0 & ~255)) return;   // (this 256-point tile: the pair kernel's)
  double *s_fine = s_dyn + kImwDoubles + (size_t)wave * fine_wave_doubles(nfmax);   // [nfmax]
  double *s_nu0 = s_fine + nfmax, *s_amp = s_nu0 + 64, *s_xs = s_amp + 64, *s_y = s_xs + 64, *s_cut = s_y + 64;
  int *s_rng = reinterpret_cast<int *>(s_cut + 64);                 // [64][4]: first point, core first / last, last point
  int *s_ord = s_rng + 4 * 64;                                      // [64]
  const int dv = (int)sv[1], h = dv / 2;
  const double invT = 1.0 / sv[0], step = d.wndelt / dv, inv_step = dv / d.wndelt;
  const long kmax = (long)(d.wfull - 1) * dv;            // last fine point of the full grid
  const int tile_end = min(tile0 + 64, a.W);
  const int TO = min(64, max(1, ftarget / dv));          // output points per sub-tile
  const int g_lo = a.per_group ? blockIdx.z : 0, g_hi = a.per_group ? blockIdx.z + 1 : d.ngroup;
  for (int o0 = tile0; o0 < tile_end; o0 += TO) {
    const int o1 = min(o0 + TO, tile_end);               // output points [o0, o1)
    const long c0 = (long)(d.i_off + o0) * dv, c1 = (long)(d.i_off + o1 - 1) * dv;
    const long fa = max(c0 - h, 0L), fb = min(c1 + h, kmax);
    const int nf = (int)(fb - fa + 1);                   // <= max(kFineTarget, dv) + 1 <= nfmax
    for (int p = lane; p < nf; p += 64) s_fine[p] = 0.0;
    // fine point k of the full grid: wn_first + k * step, product and sum rounded one
    // after the other (a fused multiply-add would move points that sit exactly on a
    // line's cut to the other side of it)
    const double fa_d = (double)fa;       // fine indices are integers below 2^53: fa_d + k is exact
    const double nu_a = add_rounded(d.wn_first, mul_rounded(fa_d, step));
    const double nu_b = add_rounded(d.wn_first, mul_rounded((double)fb, step));
    wave_sync();
    for (int g = g_lo; g < g_hi; g++) {
      long j0, j1;
      double cmax;
      line_window(d, sv, g, nu_a, nu_b, j0, j1, cmax);
      const double thresh = d.ethresh * a.smax[(size_t)st * d.ngroup + g];
      for (long base = j0; base < j1; base += 64) {
        const long j = base + lane;
        LineRec r{1e300, 0.0, 0.0, 1.0, -1.0};
        if (j < j1) r = stage_line(d, sv, invT, thresh, j);
        int r0 = 0, r1 = 0, r2 = -1, r3 = -1;
        if (r.cut >= 0.0) {
          // the line's points of this sub-tile, one point of slack either side (the exact
          // |nu - nu0| <= cut test is per point)
          const double ctr = (r.nu0 - d.wn_first) * inv_step, wd = r.cut * inv_step;
          long lo = (long)floor(ctr - wd) - 1, hi = (long)ceil(ctr + wd) + 1;
          lo = max(lo, fa); hi = min(hi, fb);
          if (lo > hi) r.cut = -1.0;      // out of this sub-tile's reach
          else {
            long a0 = hi + 1, a1 = hi;    // core [a0, a1]: the points with |z| < 8, none if y >= 8
            if (r.y < 8.0) {
              const double d8 = sqrt(64.0 - r.y * r.y) / r.xs * inv_step;
              a0 = (long)ceil(ctr - d8); a1 = (long)floor(ctr + d8);
              a0 = min(max(a0, lo), hi + 1); a1 = max(min(a1, hi), a0 - 1);
            }
            r0 = (int)(lo - fa); r1 = (int)(a0 - fa); r2 = (int)(a1 - fa); r3 = (int)(hi - fa);
          }
        }
        const unsigned long long keep = __ballot(r.cut >= 0.0);
        const int cnt = __popcll(keep);
        if (cnt == 0) continue;
        if (r.cut >= 0.0) {
          const int pos = __popcll(keep & ((1ull << lane) - 1ull));   // list order
          s_nu0[pos] = r.nu0; s_amp[pos] = r.amp; s_xs[pos] = r.xs; s_y[pos] = r.y; s_cut[pos] = r.cut;
          s_rng[4 * pos] = r0; s_rng[4 * pos + 1] = r1; s_rng[4 * pos + 2] = r2; s_rng[4 * pos + 3] = r3;
          s_ord[pos] = voigt_order(r.y);
        }
        wave_sync();
        for (int t = 0; t < cnt; t++) {
          const double l_nu0 = s_nu0[t], l_amp = s_amp[t], l_xs = s_xs[t], l_y = s_y[t], l_cut = s_cut[t];
          const int l_ord = __builtin_amdgcn_readfirstlane(s_ord[t]);
          const int klo = __builtin_amdgcn_readfirstlane(s_rng[4 * t]);
          const int kc0 = __builtin_amdgcn_readfirstlane(s_rng[4 * t + 1]);
          const int kc1 = __builtin_amdgcn_readfirstlane(s_rng[4 * t + 2]);
          const int khi = __builtin_amdgcn_readfirstlane(s_rng[4 * t + 3]);
          auto point = [&](int k) {
            const double nu = add_rounded(d.wn_first, mul_rounded(add_rounded(fa_d, (double)k), step));
            const double dx = fabs(nu - l_nu0);
            if (dx <= l_cut)
              s_fine[k] = add_rounded(s_fine[k], mul_rounded(l_amp, voigt_k(dx * l_xs, l_y, s_tab, [&] { return l_ord; })));
          };
          const int nl = kc0 - klo, nw = nl + (khi - kc1);
          const int gap = kc1 + 1 - nl - klo;     // what the right wing's points lie beyond the left wing's run
          for (int b = lane; b < nw; b += 64) point(klo + b + (b < nl ? 0 : gap));
          for (int k = kc0 + lane; k <= kc1; k += 64) point(k);
        }
        wave_sync();
      }
    }
    wave_sync();
    if (lane < o1 - o0) {
      const int o = o0 + lane;
      const long c = (long)(d.i_off + o) * dv;
      const long lo = max(c - h, 0L), hi = min(c + h, kmax);
      const bool even = (dv & 1) == 0;
      double sum = 0.0, wsum = 0.0;
      for (long f = lo; f <= hi; f++) {
        const double wgt = (even && (f == c - h || f == c + h)) ? 0.5 : 1.0;
        sum = fma(wgt, s_fine[f - fa], sum);
        wsum += wgt;
      }
      const double v = sum / wsum;
      if (a.per_group) a.out[((size_t)st * d.ngroup + blockIdx.z) * a.W + o] = v;
      else a.out[(size_t)st * a.W + o] = v;
    }
    wave_sync();
  }
}


// ---------------------------------------------------------------------------
// Width-grid Voigt evaluation (LblDev::voigt_grid).
//
// Nearest index of a half-width on a log-spaced grid (ln0, dln, n).
__device__ __forceinline__ int grid_index(double a, double ln0, double dln, int n) {
  if (n <= 1 || !(dln > 0.0)) return 0;
  const double t = floor((log(a) - ln0) / dln + 0.5);
  return t < 0.0 ? 0 : (t > (double)(n - 1) ? n - 1 : (int)t);
}

// Sampling point nearest to a line centre on a grid of spacing 1 / inv_step from wn_first:
// product and sum rounded one after the other (the scalar restatement does the same).
__device__ __forceinline__ long centre_index(double nu0, double wn_first, double inv_step) {
  return (long)floor(add_rounded(mul_rounded(nu0 - wn_first, inv_step), 0.5));
}

// One wave per state, lane = isotope: which profiles the state needs -- the isotope's Lorentz
// index, the run of Doppler indices its lines' centres span, the reach of each profile on the
// state's sampling grid -- and where they go in the state's block of the profile buffer.
__global__ __launch_bounds__(64) void lbl_grid_layout(LblDev d, const double *state, int *ginfo, int *gK, long *goff,
                                                      long *gsize) {
  const int st = blockIdx.x, k = threadIdx.x;
  const double *sv = state + (size_t)st * (2 + 3 * d.niso);
  const double step = d.wndelt / sv[1];
  long mine = 0;
  int iL = 0, iDa = 0, nD = 0;
  if (k < d.niso) {
    iL = grid_index(sv[3 + 3 * k], d.lor_ln0, d.lor_dln, d.nlor);
    iDa = grid_index(d.nu_lo * sv[2 + 3 * k], d.dop_ln0, d.dop_dln, d.ndop);
    const int iDb = grid_index(d.nu_hi * sv[2 + 3 * k], d.dop_ln0, d.dop_dln, d.ndop);
    nD = min(iDb - iDa + 1, d.dspan);
    for (int j = 0; j < nD; j++) {
      const int K = (int)floor(d.nwidth * fmax(d.dgrid[iDa + j], d.lgrid[iL]) / step);
      gK[((size_t)st * d.niso + k) * d.dspan + j] = K;
      goff[((size_t)st * d.niso + k) * d.dspan + j] = mine;   // within the isotope's run; shifted below
      mine += K + 1;
    }
    ginfo[((size_t)st * d.niso + k) * 3] = iL;
    ginfo[((size_t)st * d.niso + k) * 3 + 1] = iDa;
    ginfo[((size_t)st * d.niso + k) * 3 + 2] = nD;
  }
  long incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const long v = __shfl_up(incl, o);
    if (k >= o) incl += v;
  }
  if (k < d.niso)
    for (int j = 0; j < nD; j++) goff[((size_t)st * d.niso + k) * d.dspan + j] += incl - mine;
  if (k == 63) gsize[st] = incl;
}

// exclusive scan of the states' sizes (in place: gsize[st] -> first double of state st; gsize[nstate] = total)
__global__ void lbl_grid_scan(long *gsize, int nstate) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long run = 0;
  for (int s = 0; s < nstate; s++) { const long n = gsize[s]; gsize[s] = run; run += n; }
  gsize[nstate] = run;
}

// The profiles: block (Doppler index of the run, isotope, state), lanes over the offsets 0 .. K.
// P_j = sqrt(ln 2 / pi) / aD * Re w(sqrt(ln 2) (j step + i aL) / aD) at the GRID widths.
__global__ __launch_bounds__(256) void lbl_grid_profiles(LblDev d, const double *state, const int *ginfo, const int *gK,
                                                         const long *goff, const long *gbase, double *ptab) {
  __shared__ double s_tab[kImwDoubles];
  const int j = blockIdx.x, k = blockIdx.y, st = blockIdx.z;
  const int *gi = ginfo + ((size_t)st * d.niso + k) * 3;
  if (j >= gi[2]) return;
  load_imw_table(s_tab);
  __syncthreads();
  const double *sv = state + (size_t)st * (2 + 3 * d.niso);
  const double step = d.wndelt / sv[1];
  const double aD = d.dgrid[gi[1] + j], aL = d.lgrid[gi[0]];
  const double xs = kSqrtLn2 / aD, amp = kSqrtLn2 * kInvSqrtPi / aD, y = aL * xs;
  const int K = gK[((size_t)st * d.niso + k) * d.dspan + j];
  double *out = ptab + gbase[st] + goff[((size_t)st * d.niso + k) * d.dspan + j];
  for (int o = threadIdx.x; o <= K; o += blockDim.x)
    out[o] = mul_rounded(amp, voigt_k(mul_rounded(mul_rounded((double)o, step), xs), y, s_tab));
}

// Accumulation by table lookup, every state (oversampled or not).  Line-major like
// lbl_accumulate_fine: one wave per sub-tile of sampling points whose sums live in LDS; 64 lines
// are staged at a time (lane = line: strength, threshold, the profile its widths select, the
// sampling point nearest to its centre), then each kept line adds strength x profile[|k - kc|]
// with lane = sampling point -- one load and one multiply-add per (line, point) instead of a
// Faddeeva evaluation.  Every point adds its lines in list order, whatever the tiling.
constexpr int kGridTile = 512;   // output points per workgroup of lbl_accumulate_grid on coarsely sampled states
__global__ __launch_bounds__(64) void lbl_accumulate_grid(LblDev d, AccArgs a, int nfmax) {
  extern __shared__ double s_dyn[];
  double *s_fine = s_dyn;                                     // [nfmax]
  const int st = blockIdx.y;
  const double *sv = a.state + (size_t)st * (2 + 3 * d.niso);
  const int dv = (int)sv[1], h = dv / 2;
  // The launch has one workgroup per 64 output points.  A state sampled on (or near) the output
  // points gives a line one step of 64 lanes per such tile: there every eighth workgroup takes
  // 512 points and the others leave (eight steps per staged line: 7.4 -> 6.1 ms on config 5 at
  // wnosamp 1).  Finely sampled states keep the small tiles (many short workgroups balance the
  // chip better: 512-point tiles took config 5 at wnosamp 2160 from 22.6 to 27.7 ms).
  const int tile = dv <= 8 ? kGridTile : 64;
  if ((blockIdx.x * 64) % tile != 0) return;
  const int tile0 = blockIdx.x * 64;
  const double invT = 1.0 / sv[0], step = d.wndelt / dv, inv_step = dv / d.wndelt;
  const int lane = threadIdx.x;
  const long kmax = (long)(d.wfull - 1) * dv;            // last sampling point of the full grid
  const int tile_end = min(tile0 + tile, a.W);
  // output points per sub-tile: about kFineTarget sampling points
  const int TO = min(tile, max(1, kFineTarget / dv));
  const int g_lo = a.per_group ? blockIdx.z : 0, g_hi = a.per_group ? blockIdx.z + 1 : d.ngroup;
  const int *gi_st = a.ginfo + (size_t)st * d.niso * 3;
  const long base_st = a.gbase[st];
  for (int o0 = tile0; o0 < tile_end; o0 += TO) {
    const int o1 = min(o0 + TO, tile_end);               // output points [o0, o1)
    const long c0 = (long)(d.i_off + o0) * dv, c1 = (long)(d.i_off + o1 - 1) * dv;
    const long fa = max(c0 - h, 0L), fb = min(c1 + h, kmax);
    const int nf = (int)(fb - fa + 1);
    for (int p = lane; p < nf; p += 64) s_fine[p] = 0.0;
    const double nu_a = d.wn_first + (double)fa * step, nu_b = d.wn_first + (double)fb * step;
    wave_sync();
    for (int g = g_lo; g < g_hi; g++) {
      // window of the sorted list: the widest profile of the group's isotopes, plus the snap
      double cmax = 0.0;
      for (int k = 0; k < d.niso; k++)
        if (d.iso_group[k] == g) {
          const int *gi = gi_st + 3 * k;
          cmax = fmax(cmax, d.nwidth * fmax(d.dgrid[gi[1] + gi[2] - 1], d.lgrid[gi[0]]));
        }
      cmax += 2.0 * step;
      int b0 = (int)floor((nu_a - cmax - d.bmin) / d.bstep), b1 = (int)floor((nu_b + cmax - d.bmin) / d.bstep) + 1;
      b0 = b0 < 0 ? 0 : (b0 > d.nbucket ? d.nbucket : b0);
      b1 = b1 < 0 ? 0 : (b1 > d.nbucket ? d.nbucket : b1);
      const long j0 = d.bucket[d.boff[g] + b0], j1 = d.bucket[d.boff[g] + b1];
      const double thresh = d.ethresh * a.smax[(size_t)st * d.ngroup + g];
      for (long base = j0; base < j1; base += 64) {
        const long j = base + lane;
        bool keep1 = false;
        double Sj = 0.0;
        long toff = 0;
        int r0 = 0, rc = 0, r1 = -1;
        if (j < j1) {
          const int k = d.liso[j];
          const double n0 = d.nu0[j];
          Sj = line_strength(d.gf[j], d.elow[j], n0, sv[4 + 3 * k], invT);
          if (Sj >= thresh && Sj > 0.0) {
            const int *gi = gi_st + 3 * k;
            int jd = grid_index(n0 * sv[2 + 3 * k], d.dop_ln0, d.dop_dln, d.ndop) - gi[1];
            jd = jd < 0 ? 0 : (jd >= gi[2] ? gi[2] - 1 : jd);
            const size_t slot = ((size_t)st * d.niso + k) * d.dspan + jd;
            const int K = a.gK[slot];
            const long kc = centre_index(n0, d.wn_first, inv_step);
            const long lo = max(kc - K, fa), hi = min(kc + K, fb);
            if (lo <= hi) {
              keep1 = true;
              toff = base_st + a.goff[slot];
              r0 = (int)(lo - fa); r1 = (int)(hi - fa);
              rc = (int)max(min(kc - fa, (long)INT_MAX / 2), (long)INT_MIN / 2);
            }
          }
        }
        // the kept lines one after the other, in list order; a line's strength, table and range
        // come out of its staging lane's registers (v_readlane): no LDS traffic, no compaction
        const int t_lo = (int)(unsigned)toff, t_hi = (int)(unsigned)((unsigned long long)toff >> 32);
        const int s_lo = __double2loint(Sj), s_hi = __double2hiint(Sj);
        for (unsigned long long live = __ballot(keep1); live; live &= live - 1) {
          const int t = __builtin_ctzll(live);
          const double l_amp = __hiloint2double(__builtin_amdgcn_readlane(s_hi, t), __builtin_amdgcn_readlane(s_lo, t));
          const long off = (long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(t_hi, t) << 32) |
                                  (unsigned)__builtin_amdgcn_readlane(t_lo, t));
          const double *tab = a.ptab + off;
          const int klo = __builtin_amdgcn_readlane(r0, t), kc = __builtin_amdgcn_readlane(rc, t);
          const int khi = __builtin_amdgcn_readlane(r1, t);
          for (int k = klo + lane; k <= khi; k += 64)
            s_fine[k] = add_rounded(s_fine[k], mul_rounded(l_amp, tab[abs(k - kc)]));
        }
      }
    }
    wave_sync();
    for (int ob = o0 + lane; ob < o1; ob += 64) {
      const int o = ob;
      const long c = (long)(d.i_off + o) * dv;
      const long lo = max(c - h, 0L), hi = min(c + h, kmax);
      const bool even = (dv & 1) == 0;
      double sum = 0.0, wsum = 0.0;
      for (long f = lo; f <= hi; f++) {
        const double wgt = (even && (f == c - h || f == c + h)) ? 0.5 : 1.0;
        sum = fma(wgt, s_fine[f - fa], sum);
        wsum += wgt;
      }
      const double v = sum / wsum;
      if (a.per_group) a.out[((size_t)st * d.ngroup + blockIdx.z) * a.W + o] = v;
      else a.out[(size_t)st * a.W + o] = v;
    }
    wave_sync();
  }
}

// Fused lazy line-by-line + eclipse RT (see lbl.hpp).  One workgroup of 256
// lanes per (wavenumber tile, walker); same per-layer arithmetic as
// rt_eclipse, with the layer's line sum computed in place.
__global__ __launch_bounds__(256) void lbl_rt_eclipse_k(LblDev d, const double *state,
                                                         const double *smax, RtArgs p) {
  extern __shared__ double smem[];
  __shared__ double s_nu0[256], s_amp[256], s_xs[256], s_y[256], s_cut[256];
  __shared__ double s_tab[kImwDoubles];
  const int C = p.C, L = p.L, W = p.W, A = p.A;
  const int NC = coef_stride(0, C), NI = idx_stride(C);
  const int w = blockIdx.y;
  const int tile0 = blockIdx.x * 256;
  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  load_imw_table(s_tab);
  {
    const double *gC = p.coef + (size_t)w * L * NC;
    const idx_t *gI = p.idx + (size_t)w * L * NI;
    stage2_to_lds(sC, gC, L * NC, sI, gI, L * NI, threadIdx.x, 256);
  }
  __syncthreads();
  const int i = tile0 + threadIdx.x;
  const bool valid = i < W;
  const int ii = valid ? i : W - 1;
  const double nu = p.wn[ii];
  const double nu_a = p.wn[tile0], nu_b = p.wn[min(tile0 + 255, W - 1)];
  const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
  const double nu4 = (nu * nu) * (nu * nu);
  double I[kMaxAngles], fprev[kMaxAngles];
#pragma unroll
  for (int a = 0; a < kMaxAngles; a++) { I[a] = 0.0; fprev[a] = 1.0; }
  double tau = 0.0, eprev = 0.0, Bprev = 0.0;
  bool active = true;
  const int kraw = p.kstop[w], kend = kstop_layer(kraw);
  for (int k = 0; k <= kend; ++k) {
    const int l = L - 1 - k;
    const int st = w * L + l;
    const double *sv = state + (size_t)st * (2 + 3 * d.niso);
    const double invT = 1.0 / sv[0];
    double acc = 0.0;
    for (int g = 0; g < d.ngroup; g++) {
      double cmax = 0.0;
      for (int q = 0; q < d.niso; q++)
        if (d.iso_group[q] == g)
          cmax = fmax(cmax, d.nwidth * fmax(sv[3 + 3 * q], (nu_b + 1.0) * sv[2 + 3 * q] * 1.001));
      // window from the bucket index (a bucket too wide on either side; the
      // exact per-line cut is applied below)
      const double lo = nu_a - cmax * 1.01, hi = nu_b + cmax * 1.01;
      int b0 = (int)floor((lo - d.bmin) / d.bstep), b1 = (int)floor((hi - d.bmin) / d.bstep) + 1;
      b0 = b0 < 0 ? 0 : (b0 > d.nbucket ? d.nbucket : b0);
      b1 = b1 < 0 ? 0 : (b1 > d.nbucket ? d.nbucket : b1);
      const long j0 = d.bucket[d.boff[g] + b0], j1 = d.bucket[d.boff[g] + b1];
      const double thresh = d.ethresh * smax[(size_t)st * d.ngroup + g];
      for (long base = j0; base < j1; base += 256) {
        const long j = base + threadIdx.x;
        double cut = -1.0, n0 = 0.0, amp = 0.0, xs = 0.0, yy = 1.0;
        if (j < j1) {
          const int q = d.liso[j];
          n0 = d.nu0[j];
          const double S = line_strength(d.gf[j], d.elow[j], n0, sv[4 + 3 * q], invT);
          if (S >= thresh && S > 0.0) {
            const double aD = n0 * sv[2 + 3 * q], aL = sv[3 + 3 * q];
            cut = d.nwidth * fmax(aD, aL);
            xs = kSqrtLn2 / aD;
            amp = S * kSqrtLn2 * kInvSqrtPi / aD;
            yy = aL * xs;
          }
        }
        s_nu0[threadIdx.x] = n0; s_amp[threadIdx.x] = amp; s_xs[threadIdx.x] = xs;
        s_y[threadIdx.x] = yy; s_cut[threadIdx.x] = cut;
        __syncthreads();
        if (active) {   // lanes (and whole waves) already past toomuch skip the line sum
          const int cnt = (int)min((long)256, j1 - base);
          for (int t = 0; t < cnt; t++) {
            const double dv = fabs(nu - s_nu0[t]);
            if (dv <= s_cut[t]) acc += s_amp[t] * voigt_k(dv * s_xs[t], s_y[t], s_tab);
          }
        }
        __syncthreads();
      }
    }
    const double *c = sC + k * NC;
    const idx_t *ix = sI + k * NI;
    double e = acc + c[2 + 2 * C] * nu4 + c[3 + 2 * C];   // + Rayleigh + grey cloud
    for (int cc = 0; cc < C; cc++) {
      const double *ab = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p.cia) + ix[1 + cc]) + 2 * (size_t)ii;
      e += c[2 + 2 * cc] * ab[0] + c[3 + 2 * cc] * ab[1];   // CIA pair plane [W][2] (kernels.hpp)
    }
    const double dtau = active ? 0.5 * (eprev + e) * c[0] : 0.0;
    tau += dtau;
    const double B = bnum / (exp(c[1] * nu) - 1.0);
    const double hb = active ? 0.5 * (Bprev + B) : 0.0;
#pragma unroll
    for (int a = 0; a < kMaxAngles; a++) {
      if (a >= A) break;
      const double E = exp(-tau * p.invmu[a]);
      I[a] += hb * (fprev[a] - E);
      fprev[a] = E;
    }
    Bprev = B;
    eprev = e;
    if (active && tau > p.toomuch) active = false;
    if (!__syncthreads_or(active ? 1 : 0)) break;
  }
  double F = 0.0;
  const bool surf = kstop_deck(kraw) && active;
#pragma unroll
  for (int a = 0; a < kMaxAngles; a++) {
    if (a >= A) break;
    F += p.wgt[a] * (I[a] + (surf ? Bprev * fprev[a] : 0.0));
  }
  if (valid) p.spec[(size_t)w * W + i] = F;
}

// ---------------------------------------------------------------------------
Lbl::~Lbl() {
  auto fr = [](void *p) { if (p) (void)hipFree(p); };
  fr(d_nu0); fr(d_elow); fr(d_gf); fr(d_ztab); fr(d_ztemp); fr(d_liso);
  fr(d_state); fr(d_smax); fr(d_ext); fr(d_bucket); fr(d_dvmax);
  fr(d_dgrid); fr(d_lgrid); fr(d_ginfo); fr(d_gK); fr(d_goff); fr(d_gsize); fr(d_ptab);
}

template <class T>
static T *upv(const std::vector<T> &v) {
  T *d = nullptr;
  HIPCHK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

void lbl_init(Engine &e, const std::string &paths) {
  Lbl *b = new Lbl();
  delete e.lbl;
  e.lbl = b;
  // `linedb` may name several TLI files (comma separated; one line each in the cfg,
  // code/makecfg.py:93-104): their databases are merged in file order.  A molecule may come
  // in several databases (the usual per-band split of a line list): each stays a group of
  // its own here (own partition functions, own ethresh reference); lbl_write_opacity sums
  // the groups of one molecule into that molecule's plane.
  Tli &t = b->tli;
  for (const std::string &one_path : split_file_list(paths)) {
    Tli one = read_tli(one_path);
    if (one.db.empty()) throw IoError{"TLI file '" + one_path + "' holds no database"};
    if (t.db.empty()) { t.wn_lo = one.wn_lo; t.wn_hi = one.wn_hi; }
    t.wn_lo = std::min(t.wn_lo, one.wn_lo);
    t.wn_hi = std::max(t.wn_hi, one.wn_hi);
    for (auto &db : one.db) t.db.push_back(std::move(db));
  }
  const std::string &path = paths;
  if (t.db.empty()) throw IoError{"linedb '" + path + "' names no TLI file"};
  if ((int)t.db.size() > kMaxGroup) throw IoError{"TLI file(s): too many databases"};
  LblDev &d = b->dev;
  std::vector<double> nu0, elow, gf, ztab, ztemp;
  std::vector<int> liso;
  d.ngroup = (int)t.db.size();
  d.niso = 0;
  for (int g = 0; g < d.ngroup; g++) {
    TliDb &db = t.db[g];
    auto it = std::find(e.atm.species.begin(), e.atm.species.end(), db.molecule);
    if (it == e.atm.species.end())
      throw IoError{"TLI database molecule '" + db.molecule + "' is not in the atmosphere file"};
    d.gspecies[g] = (int)(it - e.atm.species.begin());
    int mj = e.mol.find_name(db.molecule);
    d.gdiam[g] = e.mol.diam[mj] * 1e-8;
    const int toff = (int)ztemp.size();
    ztemp.insert(ztemp.end(), db.temp.begin(), db.temp.end());
    const int iso0 = d.niso;
    for (auto &iso : db.iso) {
      if (d.niso >= kMaxIso) throw IoError{"TLI file: too many isotopes"};
      const int k = d.niso++;
      d.iso_group[k] = g;
      d.iso_mass[k] = iso.mass;
      d.iso_ratio[k] = iso.ratio;
      d.iso_toff[k] = toff;
      d.iso_nt[k] = (int)db.temp.size();
      d.iso_zoff[k] = (int)ztab.size();
      ztab.insert(ztab.end(), iso.Z.begin(), iso.Z.end());
    }
    d.gstart[g] = (long)nu0.size();
    // keep only transitions that can reach the spectral range (generous margin)
    const double lo = e.wn_full.front() - 500.0, hi = e.wn_full.back() + 500.0;
    for (size_t k = 0; k < db.wn.size(); k++) {
      if (db.wn[k] < lo || db.wn[k] > hi) continue;
      nu0.push_back(db.wn[k]); elow.push_back(db.elow[k]); gf.push_back(db.gf[k]);
      liso.push_back(iso0 + db.isoid[k]);
    }
    d.gend[g] = (long)nu0.size();
  }
  b->nlines = (long)nu0.size();
  {
    // coarse bucket index per group (1 cm-1 buckets over the kept range)
    d.bmin = std::floor(e.wn_full.front() - 500.0);
    d.bstep = 1.0;
    d.nbucket = (int)std::ceil((e.wn_full.back() + 500.0 - d.bmin) / d.bstep) + 1;
    std::vector<long> bucket;
    for (int g = 0; g < d.ngroup; g++) {
      d.boff[g] = (int)bucket.size();
      long j = d.gstart[g];
      for (int q = 0; q <= d.nbucket; q++) {
        const double edge = d.bmin + q * d.bstep;
        while (j < d.gend[g] && nu0[j] < edge) j++;
        bucket.push_back(q == d.nbucket ? d.gend[g] : j);
      }
    }
    b->d_bucket = upv(bucket);
    d.bucket = b->d_bucket;
  }
  b->d_nu0 = upv(nu0); b->d_elow = upv(elow); b->d_gf = upv(gf); b->d_liso = upv(liso);
  b->d_ztab = upv(ztab); b->d_ztemp = upv(ztemp);
  d.nu0 = b->d_nu0; d.elow = b->d_elow; d.gf = b->d_gf; d.liso = b->d_liso;
  d.ztab = b->d_ztab; d.ztemp = b->d_ztemp;
  d.nwidth = cfg_num(e.cfg, "nwidth", 20.0);
  d.ethresh = cfg_num(e.cfg, "ethresh", 1e-6);
  // sampling of the line sums (LblDev, "Sampling")
  {
    const double os = cfg_num(e.cfg, "wnosamp", 1.0);
    if (!(os >= 1.0) || os > 2160.0 || os != std::floor(os))
      throw IoError{"transit cfg: wnosamp must be an integer between 1 and 2160"};
    d.osamp = (int)os;
    d.ndiv = 0;
    for (int k = 1; k <= d.osamp; k++)
      if (d.osamp % k == 0) {
        if (d.ndiv >= 64) throw IoError{"wnosamp: too many divisors"};
        d.odiv[d.ndiv++] = k;
      }
    const char *rule = std::getenv("BARTRT_OSAMP_RULE");   // divisor (default) | full
    d.osamp_rule = rule && std::string(rule) == "full" ? 1 : 0;
    d.wn_first = e.wn_full.front();
    d.wndelt = cfg_num(e.cfg, "wndelt", 1.0) * cfg_num(e.cfg, "wnfct", 1.0);   // the grid's own spacing (Engine::setup)
    d.i_off = e.lo;
    d.wfull = e.Wfull;
  }
  // Voigt evaluation (LblDev, "Voigt evaluation"): `voigt exact` (default) or
  // `voigt grid`, BARTRT_VOIGT overrides; grid sizes `ndop` / `nlor` (40 each), ranges `dmin` /
  // `dmax` / `lmin` / `lmax` in cm-1.  Default ranges: Doppler half-widths of the TLI's isotopes
  // over the line centres kept and the temperatures tlow .. thigh (defaults 500 / 3000 K, the
  // opacity grid's keys); Lorentz half-widths over the atmosphere file's layers and
  // abundances at those two temperatures.
  {
    // A/B switch: BARTRT_VOIGT_TAYLOR=0 evaluates every |z| < 8 sample with Weideman's approximation
    const char *ty = std::getenv("BARTRT_VOIGT_TAYLOR");
    const double ymax = ty && std::string(ty) == "0" ? 0.0 : kTaylorYmax;
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_taylor_ymax), &ymax, sizeof(double)));
  }
  {
    std::string v = cfg_has(e.cfg, "voigt") ? e.cfg["voigt"] : "exact";
    if (const char *ev = std::getenv("BARTRT_VOIGT")) if (*ev) v = ev;
    if (v != "exact" && v != "grid") throw IoError{"voigt: '" + v + "' is neither exact nor grid"};
    d.voigt_grid = v == "grid";
    d.nu_lo = nu0.empty() ? e.wn_full.front() : *std::min_element(nu0.begin(), nu0.end());
    d.nu_hi = nu0.empty() ? e.wn_full.back() : *std::max_element(nu0.begin(), nu0.end());
    d.ndop = d.nlor = 1;
    d.dspan = 1;
    if (d.voigt_grid) {
      const int nd = (int)cfg_num(e.cfg, "ndop", 40.0), nl = (int)cfg_num(e.cfg, "nlor", 40.0);
      if (nd < 1 || nd > 4096 || nl < 1 || nl > 4096) throw IoError{"transit cfg: ndop / nlor must lie in 1 .. 4096"};
      const double tlo = cfg_num(e.cfg, "tlow", 500.0), thi = cfg_num(e.cfg, "thigh", 3000.0);
      if (!(tlo > 0) || !(thi >= tlo)) throw IoError{"transit cfg: bad tlow / thigh"};
      double dmin = 1e300, dmax = 0.0, lmin = 1e300, lmax = 0.0;
      for (int k = 0; k < d.niso; k++) {
        const double mi = d.iso_mass[k] * kAMU;
        auto dop = [&](double T) { return std::sqrt(2.0 * 0.6931471805599453 * kKB * T / mi) / kLS; };
        dmin = std::min(dmin, d.nu_lo * dop(tlo));
        dmax = std::max(dmax, d.nu_hi * dop(thi));
        const int g = d.iso_group[k];
        for (int l = 0; l < e.L; l++) {
          double sum = 0.0;
          for (int c : {e.iH2, e.iHe}) {
            if (c < 0) continue;
            const double dd = 0.5 * (d.gdiam[g] + e.mol.diam[e.mol.find_name(e.atm.species[c])] * 1e-8);
            sum += e.atm.abund[(size_t)l * e.S + c] * dd * dd * std::sqrt(1.0 / mi + 1.0 / (e.mass[c] * kAMU));
          }
          auto lor = [&](double T) { return std::sqrt(2.0) / (kLS * std::sqrt(kPI * kKB * T)) * e.atm.press[l] * sum; };
          if (sum > 0.0) { lmin = std::min(lmin, lor(thi)); lmax = std::max(lmax, lor(tlo)); }
        }
      }
      if (!(lmax > 0.0)) { lmin = lmax = 1e-30; }   // no collider in the atmosphere file: pure Doppler profiles
      dmin = cfg_num(e.cfg, "dmin", dmin); dmax = cfg_num(e.cfg, "dmax", dmax);
      lmin = cfg_num(e.cfg, "lmin", lmin); lmax = cfg_num(e.cfg, "lmax", lmax);
      if (!(dmin > 0) || !(dmax >= dmin) || !(lmin > 0) || !(lmax >= lmin))
        throw IoError{"transit cfg: bad Voigt width-grid range (dmin / dmax / lmin / lmax)"};
      auto logspace = [](double a, double b, int n, double &ln0, double &dln) {
        std::vector<double> g(n);
        ln0 = n > 1 ? std::log(a) : 0.5 * (std::log(a) + std::log(b));
        dln = n > 1 ? (std::log(b) - std::log(a)) / (n - 1) : 0.0;
        for (int i = 0; i < n; i++) g[i] = std::exp(ln0 + dln * i);
        return g;
      };
      const std::vector<double> dg = logspace(dmin, dmax, nd, d.dop_ln0, d.dop_dln);
      const std::vector<double> lg = logspace(lmin, lmax, nl, d.lor_ln0, d.lor_dln);
      b->d_dgrid = upv(dg); b->d_lgrid = upv(lg);
      d.dgrid = b->d_dgrid; d.lgrid = b->d_lgrid;
      d.ndop = nd; d.nlor = nl;
      // at one state an isotope's Doppler widths follow its lines' centres: a factor nu_hi / nu_lo
      d.dspan = d.dop_dln > 0.0 ? std::min(nd, (int)std::ceil(std::log(d.nu_hi / d.nu_lo) / d.dop_dln) + 2) : 1;
    }
  }
  // free the host copy of the big arrays
  for (auto &db : t.db) { db.wn.clear(); db.wn.shrink_to_fit(); db.elow.clear(); db.elow.shrink_to_fit();
                          db.gf.clear(); db.gf.shrink_to_fit(); db.isoid.clear(); db.isoid.shrink_to_fit(); }
}

static void ensure_states(Engine &e, long nstate, bool need_ext) {
  Lbl *b = e.lbl;
  if (nstate <= b->cap_state && (!need_ext || b->d_ext)) return;
  HIPCHK(hipDeviceSynchronize());
  auto re = [&](double *&p, size_t n) {
    if (p) HIPCHK(hipFree(p));
    p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(double)));
  };
  if (!b->d_dvmax) HIPCHK(hipMalloc(&b->d_dvmax, sizeof(int)));
  long cap = std::max(nstate, b->cap_state);
  re(b->d_state, (size_t)cap * (2 + 3 * b->dev.niso));
  re(b->d_smax, (size_t)cap * b->dev.ngroup);
  if (need_ext) re(b->d_ext, (size_t)cap * e.W());
  b->cap_state = cap;
}

static void run_states(Engine &e, StateArgs &sa, AccArgs &aa, hipStream_t st) {
  Lbl *b = e.lbl;
  const LblDev &d = b->dev;
  sa.L = e.L; sa.S = e.S; sa.iH2 = e.iH2; sa.iHe = e.iHe;
  sa.press = e.d_press; sa.mass = e.d_mass; sa.diam = e.d_diam;
  sa.state = b->d_state;
  sa.dvmax = b->d_dvmax;
  HIPCHK(hipMemsetAsync(b->d_dvmax, 0, sizeof(int), st));
  hipLaunchKernelGGL(lbl_states, dim3(sa.nstate), dim3(64), 0, st, sa, d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(b->d_smax, 0, sizeof(double) * (size_t)sa.nstate * d.ngroup, st));
  const int nb = (int)std::max<long>(1, std::min<long>(64, b->nlines / 4096 + 1));
  hipLaunchKernelGGL(lbl_smax, dim3(nb, sa.nstate, d.ngroup), dim3(256), 0, st, d, b->d_state,
                     b->d_smax, sa.nstate);
  HIPCHK(hipGetLastError());
  aa.W = e.W(); aa.nstate = sa.nstate; aa.wn = e.d_wn; aa.state = b->d_state; aa.smax = b->d_smax;
  aa.pair_reach = kPairReach;
  static const int mid_reach = [] { const char *v = std::getenv("BARTRT_MID_REACH"); return v && *v ? std::max(0, atoi(v)) : kMidReach; }();
  aa.mid_reach = mid_reach;
  if (d.voigt_grid) {
    // width-grid mode: lay the states' profile tables out, size the buffer, tabulate, accumulate
    const long ns = sa.nstate;
    if (ns > b->cap_gstate) {
      HIPCHK(hipDeviceSynchronize());
      auto re = [&](auto *&p, size_t n) {
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        HIPCHK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(*p)));
      };
      re(b->d_ginfo, (size_t)ns * d.niso * 3);
      re(b->d_gK, (size_t)ns * d.niso * d.dspan);
      re(b->d_goff, (size_t)ns * d.niso * d.dspan);
      re(b->d_gsize, (size_t)ns + 1);
      b->cap_gstate = ns;
    }
    hipLaunchKernelGGL(lbl_grid_layout, dim3(sa.nstate), dim3(64), 0, st, d, b->d_state, b->d_ginfo, b->d_gK,
                       b->d_goff, b->d_gsize);
    hipLaunchKernelGGL(lbl_grid_scan, dim3(1), dim3(1), 0, st, b->d_gsize, sa.nstate);
    HIPCHK(hipGetLastError());
    long total = 0;
    int dvmax = 0;
    HIPCHK(hipMemcpyAsync(&total, b->d_gsize + ns, sizeof(long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&dvmax, b->d_dvmax, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (total > b->cap_ptab) {
      if (b->d_ptab) HIPCHK(hipFree(b->d_ptab));
      b->d_ptab = nullptr;
      HIPCHK(hipMalloc(&b->d_ptab, std::max<size_t>((size_t)total, 1) * sizeof(double)));
      b->cap_ptab = total;
    }
    hipLaunchKernelGGL(lbl_grid_profiles, dim3(d.dspan, d.niso, sa.nstate), dim3(256), 0, st, d, b->d_state,
                       b->d_ginfo, b->d_gK, b->d_goff, b->d_gsize, b->d_ptab);
    HIPCHK(hipGetLastError());
    aa.ginfo = b->d_ginfo; aa.gK = b->d_gK; aa.goff = b->d_goff; aa.gbase = b->d_gsize; aa.ptab = b->d_ptab;
    const int nfmax = std::max(kFineTarget, std::max(dvmax, 1)) + 1;
    const size_t sh = sizeof(double) * (size_t)nfmax;
    hipLaunchKernelGGL(lbl_accumulate_grid, dim3((aa.W + 63) / 64, sa.nstate, aa.per_group ? d.ngroup : 1),
                       dim3(64), sh, st, d, aa, nfmax);
    HIPCHK(hipGetLastError());
    return;
  }
  const int ntile = (aa.W + 255) / 256;
  hipLaunchKernelGGL(lbl_accumulate, dim3(ntile, sa.nstate, aa.per_group ? d.ngroup : 1), dim3(256),
                     0, st, d, aa);
  hipLaunchKernelGGL(lbl_accumulate_pairs, dim3(ntile, sa.nstate, aa.per_group ? d.ngroup : 1), dim3(256),
                     0, st, d, aa);
  HIPCHK(hipGetLastError());
  if (d.osamp > 1 || aa.mid_reach > 0) {   // the states evaluated on a finer grid, and the moderately broad ones (each state is owned by one of the three)
    int dvmax = 1;     // decided on the device (lbl_states); the other two kernels run meanwhile
    if (d.osamp > 1) {
      HIPCHK(hipMemcpyAsync(&dvmax, b->d_dvmax, sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    if (dvmax > 1 || aa.mid_reach > 0) {
      static const int ftarget = [] { const char *v = std::getenv("BARTRT_FINE_TARGET"); return v && *v ? std::max(64, atoi(v)) : kFineTarget; }();
      static const int fwaves = [] { const char *v = std::getenv("BARTRT_FINE_WAVES"); return v && *v ? std::min(16, std::max(1, atoi(v))) : 4; }();
      // (no oversampled state: a sub-tile is 64 output points -- a quarter of the LDS, more resident waves)
      const int nfmax = dvmax > 1 ? std::max(ftarget, dvmax) + 1 : 65;
      // four waves share a workgroup's copy of the Im w table unless their fine sums would not fit
      const size_t per_wave = sizeof(double) * fine_wave_doubles(nfmax), tab = sizeof(double) * kImwDoubles;
      const int nw = tab + fwaves * per_wave <= 64 * 1024 ? fwaves : 1;
      const int ntile64 = (aa.W + 63) / 64;
      hipLaunchKernelGGL(lbl_accumulate_fine, dim3((ntile64 + nw - 1) / nw, sa.nstate, aa.per_group ? d.ngroup : 1),
                         dim3(64 * nw), tab + nw * per_wave, st, d, aa, nfmax, ftarget);
    }
    HIPCHK(hipGetLastError());
  }
}

void lbl_extinction(Engine &e, const double *d_prof, int nwalkers, hipStream_t st) {
  const long nstate = (long)nwalkers * e.L;
  ensure_states(e, nstate, true);
  StateArgs sa{};
  sa.nstate = (int)nstate; sa.table_mode = 0; sa.prof = d_prof;
  AccArgs aa{};
  aa.per_group = 0; aa.out = e.lbl->d_ext;
  run_states(e, sa, aa, st);
}

void lbl_rt_eclipse(Engine &e, const double *d_prof, int nwalkers, const RtArgs &r, hipStream_t st) {
  Lbl *b = e.lbl;
  const LblDev &d = b->dev;
  const long nstate = (long)nwalkers * e.L;
  ensure_states(e, nstate, false);
  StateArgs sa{};
  sa.nstate = (int)nstate; sa.table_mode = 0; sa.prof = d_prof;
  sa.L = e.L; sa.S = e.S; sa.iH2 = e.iH2; sa.iHe = e.iHe;
  sa.press = e.d_press; sa.mass = e.d_mass; sa.diam = e.d_diam;
  sa.state = b->d_state;
  hipLaunchKernelGGL(lbl_states, dim3(sa.nstate), dim3(64), 0, st, sa, d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(b->d_smax, 0, sizeof(double) * (size_t)nstate * d.ngroup, st));
  const int nb = (int)std::max<long>(1, std::min<long>(64, b->nlines / 4096 + 1));
  hipLaunchKernelGGL(lbl_smax, dim3(nb, sa.nstate, d.ngroup), dim3(256), 0, st, d, b->d_state,
                     b->d_smax, sa.nstate);
  HIPCHK(hipGetLastError());
  const int ntile = (r.W + 255) / 256;
  const size_t sh = sizeof(double) * (size_t)e.L * coef_stride(0, e.C) +
                    sizeof(idx_t) * (size_t)e.L * idx_stride(e.C);
  hipLaunchKernelGGL(lbl_rt_eclipse_k, dim3(ntile, nwalkers), dim3(256), sh, st, d, b->d_state,
                     b->d_smax, r);
  HIPCHK(hipGetLastError());
}

void lbl_write_opacity(Engine &e, const std::string &path, const std::vector<double> &tgrid) {
  Lbl *b = e.lbl;
  const int Nt = (int)tgrid.size(), L = e.L, G = b->dev.ngroup, W = e.W();
  if (e.lo != 0 || e.hi != e.Wfull) throw IoError{"the opacity grid is generated on an unsharded engine"};
  // one plane per MOLECULE (the file's layout): the databases of one molecule
  // -- several TLI files, or several databases in one -- are summed into its plane
  std::vector<int> ids, slot(G);
  for (int g = 0; g < G; g++) {
    const int id = e.mol.id[e.mol.find_name(b->tli.db[g].molecule)];
    auto it = std::find(ids.begin(), ids.end(), id);
    slot[g] = (int)(it - ids.begin());
    if (it == ids.end()) ids.push_back(id);
  }
  const int M = (int)ids.size();
  double *d_tg = nullptr, *d_ab = nullptr, *d_out = nullptr;
  HIPCHK(hipMalloc(&d_tg, sizeof(double) * Nt));
  HIPCHK(hipMemcpy(d_tg, tgrid.data(), sizeof(double) * Nt, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&d_ab, sizeof(double) * e.atm.abund.size()));
  HIPCHK(hipMemcpy(d_ab, e.atm.abund.data(), sizeof(double) * e.atm.abund.size(), hipMemcpyHostToDevice));
  FILE *fp = std::fopen(path.c_str(), "wb");
  if (!fp) { (void)hipFree(d_tg); (void)hipFree(d_ab); throw IoError{"cannot create opacity file '" + path + "'"}; }
  long dims[4] = {M, Nt, L, W};
  std::fwrite(dims, sizeof(long), 4, fp);
  std::fwrite(ids.data(), sizeof(int), M, fp);
  std::fwrite(tgrid.data(), sizeof(double), Nt, fp);
  std::fwrite(e.atm.press.data(), sizeof(double), L, fp);
  std::fwrite(e.wn_full.data(), sizeof(double), W, fp);
  // layers in slabs so the device buffer stays bounded: [nl][Nt][M][W]
  const int slab = std::max(1, std::min(L, (int)((size_t)256 * 1024 * 1024 / ((size_t)Nt * G * W * 8) + 1)));
  std::vector<double> host((size_t)slab * Nt * G * W), merged(M == G ? 0 : (size_t)slab * Nt * M * W);
  HIPCHK(hipMalloc(&d_out, host.size() * sizeof(double)));
  ensure_states(e, (long)slab * Nt, false);
  try {
    for (int l0 = 0; l0 < L; l0 += slab) {
      const int nl = std::min(slab, L - l0);
      StateArgs sa{};
      sa.nstate = nl * Nt; sa.table_mode = 1; sa.Nt = Nt; sa.tgrid = d_tg;
      sa.abund0 = d_ab; sa.layer0 = l0;
      AccArgs aa{};
      aa.per_group = 1; aa.out = d_out;
      run_states(e, sa, aa, e.stream);
      HIPCHK(hipStreamSynchronize(e.stream));
      HIPCHK(hipMemcpy(host.data(), d_out, sizeof(double) * (size_t)nl * Nt * G * W, hipMemcpyDeviceToHost));
      if (M == G) {
        std::fwrite(host.data(), sizeof(double), (size_t)nl * Nt * M * W, fp);
      } else {
        std::fill(merged.begin(), merged.begin() + (size_t)nl * Nt * M * W, 0.0);
        for (size_t pl = 0; pl < (size_t)nl * Nt; pl++)
          for (int g = 0; g < G; g++) {
            const double *src = host.data() + (pl * G + g) * W;
            double *dst = merged.data() + (pl * M + slot[g]) * W;
            for (int i = 0; i < W; i++) dst[i] += src[i];
          }
        std::fwrite(merged.data(), sizeof(double), (size_t)nl * Nt * M * W, fp);
      }
    }
  } catch (...) {
    std::fclose(fp); std::remove(path.c_str());
    (void)hipFree(d_tg); (void)hipFree(d_ab); (void)hipFree(d_out);
    throw;
  }
  std::fclose(fp);
  (void)hipFree(d_tg); (void)hipFree(d_ab); (void)hipFree(d_out);
}

}  // namespace bartrt
