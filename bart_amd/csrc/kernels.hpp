// Device-side argument blocks shared by the kernels and the host engine.
#pragma once
// (The kernel headers -- this file, integ.hpp, prep.hpp, rt_eclipse*.hpp -- also compile under hiprtc, which knows no
// host headers: csrc/rtc.hip builds the shapes the ahead-of-time set does not hold from these same sources.  Host-only
// parts sit behind !__HIPCC_RTC__; the few std:: names the kernels use are declared below for that compiler.)
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#else
typedef struct ihipEvent_t *hipEvent_t;
namespace std {
template <class T, T v> struct integral_constant { static constexpr T value = v; using value_type = T; constexpr operator T() const { return v; } };
using true_type = integral_constant<bool, true>;
using false_type = integral_constant<bool, false>;
template <bool B, class T, class F> struct conditional { using type = T; };
template <class T, class F> struct conditional<false, T, F> { using type = F; };
template <bool B, class T, class F> using conditional_t = typename conditional<B, T, F>::type;
}  // namespace std
#endif

// (molecules, CIA pairs) the specialised kernels are instantiated for AHEAD OF TIME
// (four CIA slots = two cross-section files under the default `cia_interp spline`: BART's usual H2-H2 + H2-He).
// Up to six table molecules (the reference's examples use one to four); seven and more are instantiated at their
// first launch (csrc/rtc.hpp) since round 6 -- the (7, *) / (8, *) pairs were 8 MB of the library.
#define BARTRT_MC_LIST(X) \
  X(1, 0) X(1, 1) X(1, 2) X(2, 0) X(2, 1) X(2, 2) X(3, 0) X(3, 1) X(3, 2) X(4, 0) X(4, 1) X(4, 2) \
  X(5, 0) X(5, 1) X(5, 2) X(6, 0) X(6, 1) X(6, 2)                                                 \
  X(1, 4) X(2, 4) X(3, 4) X(4, 4) X(5, 4) X(6, 4)
// ... of the adjacent-rows layer-parallel kernel (rt_eclipse_qadj.hpp): the same list since round 6 -- the measured table
// (kernel_table.inc) names that kernel for the few-walker launches of every shape, and a run-time compile of it at a
// shape's first launch is 5-10 s (the whole list is 5 MB of code objects)
#define BARTRT_QADJ_LIST(X) BARTRT_MC_LIST(X)
// CIA slot counts of the line-by-line hand-off kernels (no table molecules)
#define BARTRT_EXT_C_LIST(X) X(0) X(1) X(2) X(4)

// Every launch of an RT kernel goes through this: the kernel's own dispatch carries the timing
// events of RtArgs (null: a plain launch).
#define BARTRT_RT_LAUNCH(kernel, grid, block, sh, st, args) \
  hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(sh), st, (args).ev_start, (args).ev_stop, 0, args)

namespace bartrt {

// Physical constants, cgs.  H, LS, KB: the values BART copies from transit
// (reference code/constants.py:13-16).
constexpr double kH = 6.6260755e-27;
constexpr double kLS = 2.99792458e10;
constexpr double kKB = 1.380658e-16;
constexpr double kAMU = 1.66053886e-24;
constexpr double kAMAGAT = 2.68679e19;
constexpr double kPI = 3.141592653589793;
constexpr double kRaySigma0 = 2.52e-28, kRayLambda0 = 7.5e-5;
constexpr double kPolH2 = 0.8059e-24, kPolHe = 0.2051e-24;

constexpr int kMaxAngles = 16;
constexpr int kMaxCia = 8;
constexpr int kMaxMol = 16;

// Per-layer coefficient record produced by prep_profiles, consumed by the RT
// kernel from LDS.  Layer order is top -> bottom (k = 0 is the top layer).
//   [0]            dr_k   = r_{k-1} - r_k (cm), 0 for k = 0
//   [1]            c2/T_k = (h c / k_B) / T_k  (cm)
//   [2 .. 2+2M)    (rho_m (1-f), rho_m f) per table molecule
//   [.. +2C)       (n1 n2 (1-f), n1 n2 f) / amagat^2 per CIA table
//   [last - 1]     Rayleigh coefficient (multiplies wn^4)
//   [last]         grey extinction of the layer, cm-1 (the radius-ramp cloud)
__host__ __device__ inline int coef_stride(int M, int C) { return 4 + 2 * M + 2 * C; }
// chord table of the transit geometry: [row tile][step][lane], (L/16)^2 * 256 doubles
__host__ __device__ inline size_t chord_table_size(int L) {
  const size_t nkt = (size_t)(L + 15) / 16;
  return nkt * (4 * nkt) * 64;
}
__host__ __device__ inline size_t chord_table_index(int L, int k, int j) {
  const size_t nkt = (size_t)(L + 15) / 16;
  return ((size_t)(k >> 4) * (4 * nkt) + (size_t)(j >> 2)) * 64 + (size_t)(j & 3) * 16 + (size_t)(k & 15);
}
// Offset record (64-bit BYTE offsets, so the kernels add one scalar to a base
// pointer per layer): [0] start of the (layer, lower temperature) plane of the
// opacity grid, [1+c] start of the (lower, upper) pair plane of CIA table c.
using idx_t = long long;

// kstop[w] = the deepest layer of walker w's column; bit 30 set: that layer is the top of an
// opaque cloud deck (it emits as a surface).  A cloud-top pressure below the bottom layer
// leaves the column as it is: no deck (the bit is per walker: the walkers of a batch may
// carry their own cloud tops, bartrt_step_set_extras).
constexpr int kDeckBit = 1 << 30;
__host__ __device__ inline int kstop_layer(int raw) { return raw & ~kDeckBit; }
__host__ __device__ inline bool kstop_deck(int raw) { return (raw & kDeckBit) != 0; }
__host__ __device__ inline int idx_stride(int C) { return 1 + C; }

// Cooperative copy of two arrays of 8-byte words from global memory into LDS.
// A thread issues up to U independent loads of each array before its first LDS
// store.  (The plain `for (t = tid; t < n; t += nthreads) dst[t] = src[t]` loop
// compiles, for a runtime trip count, to load / wait / store per element: n /
// nthreads serial trips to memory at the head of every workgroup.)
template <int U = 8, typename TA, typename TB>
__device__ __forceinline__ void stage2_to_lds(TA *dstA, const TA *srcA, int nA, TB *dstB, const TB *srcB,
                                              int nB, int tid, int nthreads) {
  static_assert(sizeof(TA) == 8 && sizeof(TB) == 8, "8-byte words");
  const int span = U * nthreads;
  for (int base = 0; base < nA || base < nB; base += span) {
    TA a[U];
    TB b[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int t = base + j * nthreads + tid;
      a[j] = t < nA ? srcA[t] : TA(0);
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int t = base + j * nthreads + tid;
      b[j] = t < nB ? srcB[t] : TB(0);
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int t = base + j * nthreads + tid;
      if (t < nA) dstA[t] = a[j];
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int t = base + j * nthreads + tid;
      if (t < nB) dstB[t] = b[j];
    }
  }
}

// Three arrays in one batch (see stage2_to_lds).
template <int U = 8, typename TA, typename TB, typename TC>
__device__ __forceinline__ void stage3_to_lds(TA *dstA, const TA *srcA, int nA, TB *dstB, const TB *srcB,
                                              int nB, TC *dstC, const TC *srcC, int nC, int tid,
                                              int nthreads) {
  static_assert(sizeof(TA) == 8 && sizeof(TB) == 8 && sizeof(TC) == 8, "8-byte words");
  const int span = U * nthreads;
  for (int base = 0; base < nA || base < nB || base < nC; base += span) {
    TA a[U];
    TB b[U];
    TC c[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int t = base + j * nthreads + tid;
      a[j] = t < nA ? srcA[t] : TA(0);
      b[j] = t < nB ? srcB[t] : TB(0);
      c[j] = t < nC ? srcC[t] : TC(0);
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int t = base + j * nthreads + tid;
      if (t < nA) dstA[t] = a[j];
      if (t < nB) dstB[t] = b[j];
      if (t < nC) dstC[t] = c[j];
    }
  }
}

struct PrepArgs {
  int L, S, M, Nt, C, W, nwalkers;
  const double *prof;      // [nw][(S+1)][L]
  // per-engine constants in one block, in the order prep_profiles lays them out in
  // LDS: press[L] (barye, atm order, 0 = bottom), dlnp[L] (log(p[i]/p[i+1]), last
  // 0), mass[S] (amu), tgrid[Nt], 1/(tgrid[j+1]-tgrid[j]) [Nt], cia_temp[ncia_temps]
  // (concatenated), its reciprocal spacings [ncia_temps]
  const double *consts;
  int opmol[kMaxMol];      // [M] species index
  int cia_s1[kMaxCia], cia_s2[kMaxCia], cia_nt[kMaxCia], cia_toff[kMaxCia];
  int cia_poff[kMaxCia];   // first pair plane of table c in the CIA buffer
  int cia_kind[kMaxCia];   // 0: the table's values; 1: their second derivatives in T (cfg `cia_interp spline`, the entry
                           // after its values: weights (a^3 - a) h^2 / 6, (b^3 - b) h^2 / 6 instead of a, b)
  int ncia_temps;          // length of the concatenated CIA temperature grids
  // hydrostatic reference (code/makeatm.py:183-263)
  int ref_idx;             // layer closest to refpress
  int ref_exact;           // press[ref_idx] == refpress
  int ref_ib;              // bracket [ib, ib+1] holding refpress in log10 p
  double ref_f;            // interpolation fraction inside the bracket
  double ref_lnp;          // log(refpress / press[ref_idx])
  double gsurf, refradius;
  // scattering / cloud
  int scat_flag, iH2, iHe, has_cloud;
  double scat_value, cloudtop;
  // radius-ramp cloud (cfg cloudrad / cloudfct / cloudext): grey extinction 0 above
  // cloud_rup, rising linearly to cloud_ext at cloud_rdown, cloud_ext below (cm, cm-1)
  double cloud_rup, cloud_rdown, cloud_ext;
  // optional per-walker overrides [nw][3]: reference radius (cm), cloud-top pressure
  // (barye), scattering value; NaN = the engine's setting (prep_body)
  const double *over;
  // outputs
  double *coef;            // [nw][L][coef_stride]
  idx_t *idx;              // [nw][L][idx_stride]
  int *kstop;              // [nw] deepest layer index k to integrate to (| kDeckBit: see kstop_layer)
  unsigned char *ok;       // [nw]
  double *rad_out;         // optional [nw][L] hydrostatic radii, cm, atm layer order
  // transit geometry only (null otherwise): radii top -> bottom and the chord
  // segments DS(k, j) = s_{j-1} - s_j, s_j = sqrt(r_j^2 - r_k^2), j = 1..k (0
  // elsewhere), laid out as the B operands of v_mfma_f64_16x16x4: for row tile
  // kt (chords 16 kt .. 16 kt + 15) and step s (layers 4 s .. 4 s + 3), lane l
  // holds DS(16 kt + l % 16, 4 s + l / 16); see chord_table_index
  double *rtop;            // [nw][L]
  double *ds;              // [nw][chord_table_size(L)]
};

struct RtArgs {
  int L, M, Nt, C, A, W, nwalkers, ntiles;
  int nsel;                // host side only: the walker count the kernel VARIANT is chosen for (0: nwalkers).  The chain
                           // service names its registered clients here, so that a round some worker missed runs the
                           // kernel of the full batch and gives the same bits (svc_core.hpp, Backend::run)
  int Wfull;               // samples of the whole grid (W: this process's block of it): the kernel variant is chosen by
                           // the batch on the WHOLE grid, so that a sharded run adds the same numbers in the same order
                           // as the unsharded one (blocks concatenate bit for bit)
  const double *kappa;     // [L][Nt][W][M]: a wavenumber's molecules contiguous (TableLoader)
  const double *cia;       // [pair planes][W][2]
  unsigned long long kappa_bytes, cia_bytes;  // extents of the two tables
  int window;              // table of 4 GB or more: row-per-layer kernels address it through a moving window
  const double *ext;       // optional line-by-line extinction [nw][L][W] (atm layer order)
  const double *wn;        // [W]
  const double *coef;
  const idx_t *idx;
  const int *kstop;        // per walker: last layer to integrate to, | kDeckBit when it is a cloud deck
  int cloud_on;            // some walker of the launch may carry a deck (wave-uniform hint; the deck itself is per walker)
  int integ;               // integration rule of the eclipse geometry (integ.hpp: 0 / 1 / 2)
  int cut_slant;           // `toomuch` acts on each ray's slant depth tau / mu (cfg `cut slant`): generic kernel
  double toomuch;
  double invmu[kMaxAngles];
  double wgt[kMaxAngles];  // pi (sin^2 hi - sin^2 lo)
  double wq[kMaxAngles];   // wgt[a] * invmu[a]: the angle quadrature of rules 1 / 2 taken before the layer sum
  // `cut slant` (specialised kernels): mu[a] = cos(theta_a) and thr[a] = the largest vertical optical depth a
  // ray of that angle survives, i.e. the largest double t with t * invmu[a] <= toomuch (slant_thresholds)
  double mu[kMaxAngles];
  double thr[kMaxAngles];
  double thrb[kMaxAngles]; // 2^600 * the next double above thr[a] (alive_flags, rt_eclipse_s1s.hpp)
  int drank[kMaxAngles];   // rank of ray a in the order of dying (0 = the smallest thr goes first)
  void *slog;              // `cut slant` event log of the single-wave kernels: slant_log_bytes(...) bytes
  double *spec;            // [nw][W]
  double *tau_out;         // optional [W][L] (single walker), may be null
  int *last_out;           // optional [W]
  double *intens_out;      // optional [A][W] intensities per ray angle (single walker)
  int *walked_out;         // optional diagnostics: layers walked per (walker, column of the launched kernel)
  // transit geometry
  const double *rtop, *ds;
  double inv_starrad2;
  int transparent;         // transit geometry: no opaque core below the last chord (cfg `transparent`)
  // host side only (BARTRT_RT_LAUNCH): events the dispatch of the RT kernel itself stamps with its
  // start and end, or null -- timing without marker packets in the stream (bartrt_timing_*)
  hipEvent_t ev_start, ev_stop;
  // Prefetched preparation (bartrt_prefetch_profiles_dev): the single-wave kernels run
  // prep_profiles' body for the NEXT batch in nprep extra workgroups at the head of this launch
  // (the grid is prep_slots(nprep) + RT workgroups: a multiple of eight keeps the XCD map), into
  // the other set of record buffers -- the next call then starts on its RT kernel directly.
  // nprep < 0 (launch_rt_folded): THIS launch's walkers are prepared by the kernel itself -- every workgroup of the
  // layer-parallel kernels builds its walker's records in LDS from prep_next (= this batch's PrepArgs)
  int nprep;
  PrepArgs prep_next;
};
__host__ __device__ inline int prep_slots(int nprep) { return (nprep + 7) / 8 * 8; }

// thr[a] of RtArgs for a value of `toomuch`: tau > thr[a]  <=>  tau * invmu[a] > toomuch, to the bit (the
// product is monotone in tau, so the set of surviving depths is an interval that ends on one double)
inline void slant_thresholds(RtArgs &r) {
  for (int a = 0; a < r.A; a++) {
    const double im = r.invmu[a];
    double t = r.toomuch / im;
    if (!(t == t) || t > 1.7e308 || t < 0.0) { r.thr[a] = t < 0.0 ? -1.0 : 1.7976931348623157e308; continue; }
    for (int it = 0; it < 64 && t * im > r.toomuch; it++) t = __builtin_nextafter(t, 0.0);
    for (int it = 0; it < 64; it++) {
      const double up = __builtin_nextafter(t, 1.7976931348623157e308);
      if (up == t || up * im > r.toomuch) break;
      t = up;
    }
    r.thr[a] = t;
  }
  for (int a = 0; a < r.A; a++)
    r.thrb[a] = __builtin_nextafter(r.thr[a], __builtin_inf()) * 4.149515568880993e180;   // 2^600; +inf past 4e127
  for (int a = 0; a < r.A; a++) {
    int rk = 0;
    for (int b = 0; b < r.A; b++) rk += (r.thr[b] < r.thr[a] || (r.thr[b] == r.thr[a] && b < a)) ? 1 : 0;
    r.drank[a] = rk;
  }
}
// bytes of RtArgs::slog for a launch: per (walker, tile) workgroup, per lane, A slots of (tau, interval, index)
inline size_t slant_log_bytes(int nwalkers, int ntiles, int block, int A) {
  return (size_t)nwalkers * (size_t)ntiles * (size_t)block * (size_t)A * 20u;
}

// What launch_rt launched (diagnostics; the byte model of bench.py)
struct RtLaunchInfo {
  const char *kernel = "";
  int wn_per_column = 64;   // granularity of RtArgs::walked_out
  int ncolumns = 0;         // entries of walked_out per walker
  bool prep_fused = false;  // the launch carried RtArgs::nprep workgroups of the next batch's preparation
  bool prep_folded = false; // the kernel prepared its own walkers' layer records (RtArgs::nprep < 0): no prep_profiles launch
  bool rtc = false;         // the kernel was instantiated at run time (rtc.hpp), not taken from the ahead-of-time set
};

// ---------------------------------------------------------------------------
// exp(x) for -708 <= x <= 709 (callers clamp) without the special-case selects
// of the library routine: the integer part n of x / ln2 comes out of one FMA
// with the 1.5 * 2^52 shifter (its low dword is n), Cody-Waite reduction by ln2,
// degree-11 interpolant on [-ln2/2, ln2/2] (Chebyshev nodes, max relative error
// 1.7e-17 before rounding), and 2^n is applied by adding n to the exponent
// field (the result stays normal on that range).  16 VALU operations.
constexpr double kExpShift = 6755399441055744.0;  // 1.5 * 2^52
constexpr double kExpMin = -708.0;

__device__ __forceinline__ double exp_scale(double p, double shifted) {
  const int hi = __double2hiint(p) + (__double2loint(shifted) << 20);
  return __hiloint2double(hi, __double2loint(p));
}

__device__ __forceinline__ double exp_core(double x) {
  const double t = fma(x, 1.4426950408889634074, kExpShift);
  const double n = t - kExpShift;
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 2.5110037605963777e-08;
  p = fma(p, r, 2.763263963904103e-07);
  p = fma(p, r, 2.755724091857897e-06);
  p = fma(p, r, 2.4801485482328494e-05);
  p = fma(p, r, 0.00019841269890047113);
  p = fma(p, r, 0.0013888888952314775);
  p = fma(p, r, 0.008333333333319601);
  p = fma(p, r, 0.0416666666664881);
  p = fma(p, r, 0.1666666666666668);
  p = fma(p, r, 0.5000000000000019);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return exp_scale(p, t);
}

// 1/d for normal, positive d: hardware estimate + two Newton steps.  Measured on gfx950
// over 4e6 arguments between 1e-300 and 1e300: v_rcp_f64 alone 4.6e-8, one step 2.2e-15,
// two steps 1.1e-16 relative.
__device__ __forceinline__ double rcp_core(double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);
  return fma(y, e, y);
}
// One Newton step (2.2e-15): the Planck function and the Voigt approximations, whose
// own errors are 1e-13 and up.
__device__ __forceinline__ double rcp_n1(double d) {
  const double y = __builtin_amdgcn_rcp(d);
  return fma(y, fma(-d, y, 1.0), y);
}

// exp() of the RT kernels (Planck exponent and slant transmittances): same
// shifter / exponent-field scheme as exp_core with ONE reduction FMA (n * ln2 is
// exact inside the FMA; ln2 rounded to double is off by 2.3e-17 per unit of n:
// 2e-14 at |x| = 700, nothing at n = 0) and a degree-9 interpolant on
// [-ln2/2, ln2/2] whose constant and linear coefficients are exactly 1: it is
// 1 + r + r^2 g(r) with g through the Chebyshev nodes (tools/gen_exp_coef.py;
// relative error 7e-14 in the value and 1.4e-11 in the DERIVATIVE).  The
// derivative is what the intensity sums see -- they add differences of
// transmittances of neighbouring layers -- and the exact first-order term keeps
// optically thin columns (all arguments near 0) at full precision; an
// unconstrained degree-8 fit (1e-12 in the value) was off by 1e-8 there.
// 13 VALU operations instead of 16.  The spectra's tolerance is 1e-6
// (BASELINE.json north_star); the parity tests hold the kernels to 1e-10.
// N independent arguments are evaluated with their Horner steps interleaved:
// one wave alone on a SIMD then overlaps the N dependent FMA chains instead of
// paying the fp64 pipeline latency N x 9 times in a row.
template <int N>
__device__ __forceinline__ void exp_rt_n(const double (&x)[N], double (&out)[N]) {
  double t[N], r[N], q[N];
#pragma unroll
  for (int a = 0; a < N; a++) t[a] = fma(x[a], 1.4426950408889634074, kExpShift);
#pragma unroll
  for (int a = 0; a < N; a++) r[a] = fma(t[a] - kExpShift, -6.93147180559945286227e-01, x[a]);
  constexpr double cf[9] = {2.4867870179687727e-05, 0.00019841224599656011, 0.0013888839110572009,
                            0.0083333333442029804,  0.041666666786265731,   0.16666666666662586,
                            0.49999999999955108,    1.0,                    1.0};
#pragma unroll
  for (int a = 0; a < N; a++) q[a] = 2.7617564785876086e-06;
#pragma unroll
  for (int j = 0; j < 9; j++) {
#pragma unroll
    for (int a = 0; a < N; a++) q[a] = fma(q[a], r[a], cf[j]);
  }
#pragma unroll
  for (int a = 0; a < N; a++) out[a] = exp_scale(q[a], t[a]);
}

__device__ __forceinline__ double exp_rt(double x) {
  const double xs[1] = {x};
  double o[1];
  exp_rt_n<1>(xs, o);
  return o[0];
}

// Largest optical depth whose slant transmittances exp(-tau / mu_a) all stay in
// exp_core's range; beyond it they are floored at e^-708 .. e^-(708 mu_min/mu_a)
// instead of running on to zero (differences below 1e-50 of the top layers' terms).
__device__ __forceinline__ double tau_cap(const RtArgs &p, int A) {
  double m = p.invmu[0];
  for (int a = 1; a < A; a++) m = p.invmu[a] > m ? p.invmu[a] : m;
  return -kExpMin / m;
}

// ---------------------------------------------------------------------------
// Table layout in HBM (the opacity FILE keeps transit's order o[L][Nt][M][W],
// io.cpp; the engine re-lays it out once at init, grid_transpose):
//   opacity grid  kappa[L][Nt][W][M]   a wavenumber's M molecules contiguous: a lane
//                                       reads its 8 M bytes of a plane with 16-byte
//                                       loads (two molecules each), a wave 512 M
//                                       contiguous bytes per plane;
//   CIA           cia[pair plane][W][2] (alpha_j, alpha_{j+1}) of a wavenumber side
//                                       by side: one 16-byte load per CIA table.
// 2 M + 2 C values per layer arrive in M + C loads (M even) instead of 2 M + 2 C
// 8-byte loads (measured on the bench grid: -5 % at 10 walkers, -10 % at 256).
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

// this lane's M values of the two temperature planes of a layer: r[2 m + pl]
// (pl = 0 lower, 1 upper plane).  vo: byte offset of the lane's first molecule in
// the lower plane, relative to the descriptor; so: scalar byte offset added to it.
template <int M, typename RS>
__device__ __forceinline__ void load_table_lane(RS rs, unsigned vo, int so, unsigned planeB, double *r) {
#pragma unroll
  for (int pl = 0; pl < 2; pl++) {
    const int sp = so + (pl ? (int)planeB : 0);
#pragma unroll
    for (int m = 0; m + 1 < M; m += 2) {
      const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo + m * 8), sp, 0);
      r[2 * m + pl] = __builtin_bit_cast(double, (v2u_t){v.x, v.y});
      r[2 * (m + 1) + pl] = __builtin_bit_cast(double, (v2u_t){v.z, v.w});
    }
    if (M & 1)
      r[2 * (M - 1) + pl] = __builtin_bit_cast(
          double, (v2u_t)__builtin_amdgcn_raw_buffer_load_b64(rs, (int)(vo + (M - 1) * 8), sp, 0));
  }
}

// (alpha_lo, alpha_hi) of one CIA table for this lane: vo = byte offset of the
// lane's pair inside the pair plane, so = the pair plane's byte offset
template <typename RS>
__device__ __forceinline__ void load_cia_lane(RS rs, unsigned vo, int so, double *r2) {
  const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, so, 0);
  r2[0] = __builtin_bit_cast(double, (v2u_t){v.x, v.y});
  r2[1] = __builtin_bit_cast(double, (v2u_t){v.z, v.w});
}

template <int M, int C>
struct TableLoader {
  static constexpr int NLD = 2 * M + 2 * C, NR = NLD > 0 ? NLD : 1, NI = 1 + C;
  unsigned vk, vc;      // this lane's byte offset inside a plane of the grid / a CIA pair plane
  unsigned planeB;
  const char *kappa, *cia;
  int span_k, span_c;   // bytes one layer reads from: two temperature planes / one pair plane
  const idx_t *sI;

  // ii: this lane's wavenumber index; sI: the walker's offset records in LDS
  __device__ __forceinline__ TableLoader(const RtArgs &p, unsigned ii, const idx_t *sI_)
      : vk(ii * 8u * M), vc(ii * 16u), planeB((unsigned)M * (unsigned)p.W * 8u),
        kappa(reinterpret_cast<const char *>(p.kappa)), cia(reinterpret_cast<const char *>(p.cia)), sI(sI_) {
    span_k = (int)(2 * planeB);
    span_c = (int)((unsigned)p.W * 16u);
  }

  // wave-uniform 64-bit byte offset out of a record read from LDS
  static __device__ __forceinline__ long long uniform64(idx_t v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
  }

  // issue the loads of layer k into r (no wait).  The descriptor is rebuilt per
  // layer around the layer's own pair of planes (a 64-bit scalar add), so the
  // 32-bit offsets of the buffer instructions never limit the table size.
  __device__ __forceinline__ void load(int k, double *r) const {   // r[NR]
    const idx_t *ix = sI + k * NI;
    if (M > 0) {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(kappa + uniform64(ix[0])), 0, span_k, 0x00020000);
      load_table_lane<M>(rs, vk, 0, planeB, r);
    }
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(cia + uniform64(ix[1 + cc])), 0, span_c, 0x00020000);
      load_cia_lane(rs, vc, 0, r + 2 * M + 2 * cc);
    }
  }
};

// Kernels whose lane rows work on different layers (ROWS rows of 64 / ROWS lanes)
// address the opacity grid with per-lane 32-bit offsets.  For a grid of 4 GB or
// more the descriptor is rebuilt per step around the smallest plane offset among
// the rows' layers (the layers of a step are adjacent, so the window spans a few
// layer slabs; launch_* checks that it stays below 4 GB) and the lane offsets are
// taken relative to it.  `mine`: the plane offset of this lane's layer.
template <int ROWS>
__device__ __forceinline__ long long row_window_base(idx_t mine) {
  const int lo = (int)(unsigned)mine, hi = (int)(unsigned)((unsigned long long)mine >> 32);
  long long best = 0;
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, r * (64 / ROWS));
    const unsigned h = (unsigned)__builtin_amdgcn_readlane(hi, r * (64 / ROWS));
    const long long v = (long long)(((unsigned long long)h << 32) | l);
    best = (r == 0 || v < best) ? v : best;
  }
  return best;
}

// can the rows of a step (ROWS adjacent layers) be reached from one window?
inline bool window_fits(const RtArgs &a, int rows) {
  const unsigned long long slab = (unsigned long long)a.Nt * a.M * a.W * 8ull;
  return (rows + 1) * slab < (1ull << 32);
}

}  // namespace bartrt
