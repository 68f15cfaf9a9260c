// Single-wave eclipse kernel of integration rule 1 with the `toomuch` cut on each ray's SLANT
// depth (cfg `cut slant`, DESIGN.md C19: SURVEY.md App. A-4 read literally -- "slant path ds = dr /
// cos(theta); tau accumulated from the top; the loop stops where tau > toomuch").  Same walk as
// rt_eclipse_simpson (rt_eclipse_s1.hpp): one lane per (walker, wavenumber), buffer loads two
// layers ahead, layer records from LDS, the optical depth by the Simpson radius table.  What the
// per-angle cut changes:
//
//  * every ray angle a ends on its own layer k_a (the first with tau_k / mu_a > toomuch, at most the
//    vertical cut's) and so on its own PARITY: the hybrid rule starts with the trapezoid of (0, 1)
//    when the angle's point count is even, with a Simpson panel when it is odd.  The angle
//    quadrature can therefore not be taken before the layer sum: each angle keeps the two running
//    sums of the panels that end on even / odd points (P0_a, P1_a), and which of them an angle ends
//    on is known only when it dies.
//  * the panels in WEIGHT form: the Simpson weights (w0, w1, w2) of the tau panel (k-2, k-1, k)
//    depend on the column's tau grid alone, so they are formed once per layer (one reciprocal, as
//    in rt_eclipse_simpson) and an angle's panel is three multiply-adds on its last three
//    integrands y_a = B exp(-tau / mu_a).
//  * an angle is alive at layer k iff the running maximum of tau over the layers above stays at or
//    below its threshold thr_a (RtArgs::thr: the largest tau with tau / mu_a <= toomuch), so the
//    five alive flags are functions of ONE running value -- no per-angle state, sticky by
//    construction.  A dead angle's sums are frozen by multiplying its panel with the flag.
//  * the padded point (integrand 0, one unit of SLANT depth = mu_a of vertical depth past the
//    angle's last point) closes a panel (k_a - 1, k_a, pad) whose weights depend on the angle.
//    Evaluating it in line would cost every layer what it costs the one layer where it counts, and
//    carrying each angle's last two depths along costs fifteen multiply-adds per layer and the
//    registers that decide between two waves per SIMD and spilling.  Instead a lane LOGS its death
//    events: a layer on which its count of living rays dropped writes (tau, the interval it closed,
//    its index) to the lane's slot "rays dead before" of a small per-wave log in global memory
//    (RtArgs::slog; 100 bytes per lane) -- a buffer store whose offset is pushed out of the
//    descriptor's range on lanes without an event, so that the hardware drops it: no branch, no
//    dummy traffic, at most five stores per lane and column.  After the walk each ray that died
//    finds its event (the last one logged at or below its rank in the order of dying), and its pad
//    panel is evaluated once per column (four exponentials: the Planck terms and transmittances of
//    its last two points, from the layer records still in LDS).
//  * zero-width tau panels (two adjacent layers of exactly zero extinction) make a reciprocal
//    infinite and the lane's sums non-finite, which is sticky; a wave that ends with a non-finite
//    flux recomputes its columns ray by ray with SlantRay (integ.hpp), case analysis and all.
#pragma once
#include "integ.hpp"
#include "kernels.hpp"
#include "rt_eclipse_s1.hpp"

#ifndef __HIPCC_RTC__
#include <type_traits>
#endif

namespace bartrt {

// The rays' alive flags as doubles, m[a] = (tm <= thr[a]) ? 1 : 0, and their sum.  A compare and a select per ray
// cost four instructions (the select is two dwords); here tm is scaled once by -2^600 and each flag is ONE
// addition with the result clamped to [0, 1]: RtArgs::thrb[a] = 2^600 * nextafter(thr[a], +inf), so the sum
// is >= 2^600 ulp(thr) >> 1 while tm <= thr[a] and <= 0 from the next double on -- exactly the comparison, for
// every threshold above 2^-500 and every optical depth below 2^400.  (A threshold that overflows the scaling is +inf: alive.)
#ifndef BARTRT_FLAG_CLAMP
#define BARTRT_FLAG_CLAMP 1
#endif
template <int A>
__device__ __forceinline__ double alive_flags(const RtArgs &p, double tm, double (&m)[A]) {
  double n = 0.0;
#if BARTRT_FLAG_CLAMP
  const double tb = tm * -4.149515568880993e180;   // -2^600
#pragma unroll
  for (int a = 0; a < A; a++) {
    asm("v_add_f64 %0, %1, %2 clamp" : "=v"(m[a]) : "v"(tb), "s"(p.thrb[a]));
    n += m[a];
  }
#else
#pragma unroll
  for (int a = 0; a < A; a++) {
    m[a] = tm <= p.thr[a] ? 1.0 : 0.0;
    n += m[a];
  }
#endif
  return n;
}

// MIG (rt_eclipse_simpson_slant<..., MIG = true>, launched when the columns do not divide evenly over the SIMDs -- ten
// walkers on 1e4 samples are 1 570 waves on 1 024 SIMDs, six or seven per compute unit: the SIMDs that hold two finish at
// twice the time of those that hold one, which then idle): columns MIGRATE inside a compute unit.  A wave that ends its
// column as the last one at work on its SIMD, while other SIMDs of its CU still work, does not exit: it marks a slot of
// the CU's record "a wave waits" and polls it (slant_take); a wave that shares its SIMD with another one at work looks,
// every mig_cb blocks of six layers, at the CU's word and, if somebody waits, hands its column over at that block
// boundary -- the lane state of the walk (8 + 4 A doubles per lane) goes through global memory, written through (sc1)
// and drained before the slot is filled, read with sc1 loads (nothing handed over is read through a cache that may hold
// an older copy), the event log likewise.  The taker stages the column's walker, restarts the loads at the hand-over
// layer and walks on with the same instructions: the spectrum's bits do not depend on who walked what.  Which CU and
// SIMD a wave is on is read from the hardware (HW_ID, XCC_ID); a waiting wave leaves when nothing is at work on its CU
// any more, so nobody waits on work that cannot come.  Every word of the protocol belongs to one CU (kernels.hpp).

// (debug builds, -DBARTRT_MIG_DEBUG: counts of the protocol's events in the header words 8..15 of RtArgs::mig_ctl)
#ifdef BARTRT_MIG_DEBUG
#define BARTRT_MIG_COUNT(i) do { if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(ctl + (i), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define BARTRT_MIG_COUNT(i) do { } while (0)
#endif

// this wave's compute unit as a key below kMigCuKeys (XCD, shader engine / array / CU) and its SIMD in the CU
__device__ __forceinline__ void mig_where(unsigned &cu, unsigned &simd) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  cu = ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);
  simd = (hw >> 4) & 3u;
}
__device__ __forceinline__ unsigned *mig_cu_word(unsigned *ctl, unsigned cu) { return ctl + kMigCu + (size_t)cu * kMigCuStride; }
__device__ __forceinline__ unsigned long long *mig_cu_slots(unsigned *ctl, unsigned cu) {
  return reinterpret_cast<unsigned long long *>(ctl + kMigCu + (size_t)cu * kMigCuStride + kMigSlotWord);
}
// waves at work on the CU (the sum of the four SIMD counts of its word)
__device__ __forceinline__ int mig_at_work(unsigned wv) { return (int)((wv & 15u) + ((wv >> 4) & 15u) + ((wv >> 8) & 15u) + ((wv >> 12) & 15u)); }
// the first lane adds, every lane has the value from before
__device__ __forceinline__ unsigned mig_add_ret(unsigned *p, unsigned v) {
  unsigned old = 0;
  if (threadIdx.x == 0) old = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)old);
}
__device__ __forceinline__ void mig_add(unsigned *p, unsigned v) {
  if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned mig_load(const unsigned *p) {   // (the first lane loads)
  unsigned v = 0;
  if (threadIdx.x == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// a slot of this launch: [tag : state]; compare-and-swap by the first lane, every lane learns whether it took
__device__ __forceinline__ bool mig_slot_cas(unsigned long long *slot, unsigned long long expect, unsigned long long want) {
  int ok = 0;
  if (threadIdx.x == 0)
    ok = __hip_atomic_compare_exchange_strong(slot, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
  return __builtin_amdgcn_readfirstlane(ok) != 0;
}

// a walker's layer records and the Simpson weights of its radius grid into LDS (the single-wave kernels' layout)
template <int NC, int NI>
__device__ __forceinline__ void slant_stage(const RtArgs &p, double *smem, int w) {
  const int L = p.L;
  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  double *sWw = smem + (size_t)L * NC + (size_t)L * NI;
  stage2_to_lds(sC, p.coef + (size_t)w * L * NC, L * NC, sI, p.idx + (size_t)w * L * NI, L * NI, threadIdx.x,
                blockDim.x);
  __syncthreads();
  simpson_radius_table(sWw, sC, NC, L, kstop_layer(p.kstop[w]), threadIdx.x, blockDim.x);
  __syncthreads();
}

// OUT: also writes the optical depths tau_out[W][L] / last_out[W] (the deepest layer a ray of the sample reaches;
// deeper layers repeat its depth) and the per-ray intensities intens_out[A][W] of a single walker -- what `tau.dat`
// and `outintens` hold (code/cf.py:46-94 reads them back) -- from the same walk, instead of a second launch of the
// generic kernel.
// (the OUT build is a once-per-run diagnostic launch of one walker: it takes the registers of a whole SIMD -- one
// wave per SIMD -- instead of spilling the output bookkeeping)
//
// slant_column: ONE column (tile, walker w) of the walker whose records are staged in LDS, from the top (RESUME =
// false) or -- MIG -- from a hand-over at the block that begins with layer kres (RESUME = true).  Returns false when the
// column was handed over instead of finished (MIG).
template <int AT, int MT, int CT, bool SQ, int SCHED, bool EXT, bool OUT, bool MIG, bool RESUME>
__device__ __forceinline__ bool slant_column(const RtArgs &p, double *smem, const int tile, const int w, [[maybe_unused]] const int kres) {
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 4 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C;
  constexpr int NR = NLD + (EXT ? 1 : 0) > 0 ? NLD + (EXT ? 1 : 0) : 1;
  constexpr int AE = SQ ? A - 1 : A;  // transmittances that need an exponential
  constexpr int kBlk = 6;             // layers per straight-line block (even: a layer's parity is its position)
  static_assert(kBlk - 2 <= kSimpsonPad, "the radius table's overrun entries");
  static_assert(!MIG || (!OUT && !EXT), "columns migrate in the plain table kernel only");
  static_assert(MIG || !RESUME, "only a migrated column resumes");
  constexpr int NS = 8 + 4 * A;                  // MIG: doubles of lane state a hand-over carries
#ifdef BARTRT_MIG_PLAINLOG   // (A/B builds, tools/ab_build.py)
  constexpr int kLogSt = 0, kLogLd = 1;
#else
  constexpr int kLogSt = MIG ? 16 : 0;           // the event log's stores / loads: sc1 when another XCD may read them
  constexpr int kLogLd = MIG ? 16 : 1;
#endif
  const int L = p.L, W = p.W;
  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  const double *sW = smem + (size_t)L * NC + (size_t)L * NI;
  // MIG: two words of LDS per lane behind the Simpson weights, where the two counters a donor looks at arrive (below)
  [[maybe_unused]] unsigned *const mbox = reinterpret_cast<unsigned *>(smem + (size_t)L * NC + (size_t)L * NI + 4 * (size_t)(L + kSimpsonPad));
  [[maybe_unused]] unsigned *const ctl = p.mig_ctl;
  const int kraw = p.kstop[w], kend = kstop_layer(kraw);
  const bool deck_on = kstop_deck(kraw);

  const int i = tile * blockDim.x + threadIdx.x;
  const bool valid = i < W;
  const unsigned ii = valid ? (unsigned)i : (unsigned)(W - 1);
  const double nu = p.wn[ii];
  const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
  const double nu4 = (nu * nu) * (nu * nu);
  const TableLoader<M, C> tab(p, ii, sI);
  const double *extw = EXT ? p.ext + (size_t)w * L * W + ii : nullptr;
  auto load_layer = [&](int k, double (&r)[NR]) {
    tab.load(k, r);
    if constexpr (EXT) r[NLD] = extw[(size_t)(L - 1 - k) * W];
  };
  const double tcap = tau_cap(p, A);
  // `toomuch` ends a ray only where a deeper layer exists (nothing follows the bottom layer)
  const int kcut = kend < L - 2 ? kend : L - 2;
  double thr_max = p.thr[0];
#pragma unroll
  for (int a = 1; a < A; a++) thr_max = p.thr[a] > thr_max ? p.thr[a] : thr_max;

  // optical depth: tau of the last even layer, the last two extinctions
  double s_even = 0.0, eprev = 0.0, e2 = 0.0;
  // the tau grid of the intensity integrals: abscissa of the previous point, the previous
  // interval and its reciprocal; tm = the largest tau so far (layers <= kcut)
  double x1 = 0.0, h0 = 0.0, r0 = 1.0, tm = 0.0;
  [[maybe_unused]] int out_last = 0;          // OUT: the deepest layer with a living ray, its optical depth
  [[maybe_unused]] double out_tau = 0.0;
  // per ray angle: the last two integrands, the sums of the panels that end on even / odd points
  double y1[A], y2[A], P0[A], P1[A];
#pragma unroll
  for (int a = 0; a < A; a++) { y1[a] = y2[a] = P0[a] = P1[a] = 0.0; }
  // the wave's event log: [slot 0 .. A-1][lane] (tau, interval) pairs, then [slot][lane] layer indices
  // (-1: no event in that slot), addressed through a descriptor of exactly its size
  const unsigned nth = blockDim.x;                      // 64, 128 or 256
  const unsigned lsh = (unsigned)__builtin_ctz(nth) + 4u;    // log2 of a slot's bytes in the pair part of the log
  const unsigned log_bytes = nth * (unsigned)A * 20u;
  const auto rs_log = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char *>(p.slog) + ((size_t)w * p.ntiles + tile) * log_bytes, 0, (int)log_bytes, 0x00020000);
  const unsigned log_k0 = nth * (unsigned)A * 16u + threadIdx.x * 4u;   // this lane's index of slot 0
  if constexpr (!RESUME) {
#pragma unroll
    for (int sl = 0; sl < A; sl++) __builtin_amdgcn_raw_buffer_store_b32(-1, rs_log, (int)(log_k0 + sl * nth * 4u), 0, kLogSt);
  }
  double nprev = (double)A;  // rays alive when the previous layer began (a sum of the 0 / 1 flags)
  const unsigned tid16 = threadIdx.x * 16u;
  // logs the event "the rays alive dropped from nprev to nnow on layer kev" (tau and interval of that layer
  // are x1 and h0 by now); lanes without one store out of range, which the hardware drops
  auto log_event = [&](double nnow, int kev) {
    const bool ev = nnow < nprev;
    const unsigned slot = (unsigned)((double)A - nprev);
    const unsigned off = ev ? (slot << lsh) + tid16 : 0x7ffffff0u;
    const unsigned offk = ev ? log_k0 + (slot << (lsh - 2u)) : 0x7ffffff0u;
    v4u_t v;
    v.x = (unsigned)__double2loint(x1); v.y = (unsigned)__double2hiint(x1);
    v.z = (unsigned)__double2loint(h0); v.w = (unsigned)__double2hiint(h0);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs_log, (int)off, 0, kLogSt);
    __builtin_amdgcn_raw_buffer_store_b32(kev, rs_log, (int)offk, 0, kLogSt);
    nprev = nnow;
  };

  // one layer.  J = position in the six-layer block (the parity of k), FIRST = the block of
  // k0 = 0, MASKED = the column's last block: layers past kend are walked with clamped inputs
  // and masked; the blocks above it lie inside the column (k0 + 5 <= kcut) and carry no range
  // logic at all.
  auto layer = [&](auto Jc, auto Fc, auto Mc, int k0, const double (&r)[NR], const double (&cf)[NC],
                   double (&cfn)[NC]) {
    constexpr int J = decltype(Jc)::value;
    constexpr bool FIRST = decltype(Fc)::value, MASKED = decltype(Mc)::value;
    const int k = k0 + J;
    auto read_rec = [&](int kk, double (&c_)[NC]) {
      const double *c = sC + ((MASKED || J == kBlk - 1) ? (kk < kend ? kk : kend) : kk) * NC;
#pragma unroll
      for (int j = 0; j < NC; j++) c_[j] = c[j];
    };
    if constexpr (SCHED != 0) read_rec(k + 1, cfn);
    else read_rec(k, const_cast<double (&)[NC]>(cf));
    double e = fma(cf[2 + 2 * M + 2 * C], nu4, cf[3 + 2 * M + 2 * C]);   // Rayleigh + grey cloud
#pragma unroll
    for (int j = 0; j < NLD; j++) e = fma(cf[2 + j], r[j], e);
    if constexpr (EXT) e += r[NLD];
    const double *wk = sW + 4 * k;
    double tau;
    if constexpr ((J & 1) != 0) {
      tau = fma(eprev + e, wk[3], s_even);
    } else if constexpr (FIRST && J == 0) {
      tau = 0.0;
    } else {
      s_even = fma(wk[0], e2, fma(wk[1], eprev, fma(wk[2], e, s_even)));
      tau = s_even;
    }
    e2 = eprev;
    eprev = e;
    // alive flags from the running maximum of tau over the layers above (1.0 / 0.0)
    bool inr = true;                      // wave-uniform: the layer lies inside the column
    if constexpr (MASKED) inr = k <= kend;
    // (one multiplication and, per ray, one clamped addition: alive_flags)
    double m[A];
    const double nnow = alive_flags<A>(p, tm, m);
    if constexpr (OUT) {
      if (p.tau_out && valid && inr && tm <= thr_max) {   // some ray of the sample is alive on this layer
        p.tau_out[(size_t)i * L + k] = tau;
        out_last = k;
        out_tau = tau;
      }
    }
    if constexpr (MASKED) {
#pragma unroll
      for (int a = 0; a < A; a++) m[a] = inr ? m[a] : 0.0;
    }
    if constexpr (!(FIRST && J == 0)) log_event(nnow, k - 1);
    // Planck exponent and the slant-path exponents in one interleaved batch
    const double tcl = fmin(tau, tcap);
    double xs[AE + 1], ex[AE + 1], y[A];
    xs[AE] = fmin(cf[1] * nu, 700.0);
#pragma unroll
    for (int a = 0; a < AE; a++) xs[a] = -tcl * p.invmu[a];
    exp_rt_n<AE + 1>(xs, ex);
    const double B = bnum * rcp_n1(ex[AE] - 1.0);
#pragma unroll
    for (int a = 0; a < AE; a++) y[a] = B * ex[a];
    if (SQ) y[A - 1] = (B * ex[0]) * ex[0];
    if constexpr (FIRST && J == 0) {
#pragma unroll
      for (int a = 0; a < A; a++) y1[a] = y[a];
    } else {
      // the interval this layer closes (a unit one on masked overrun layers: tau stands still there)
      double h1 = tau - x1;
      if constexpr (MASKED) h1 = inr ? h1 : 1.0;
      double w0, w1, w2, r1 = 1.0;
      if constexpr (FIRST && J == 1) {
        w0 = 0.0; w1 = w2 = 0.5 * h1;      // the first interval: a trapezoid
      } else {
        r1 = rcp_n1(h1);
        const double hs = h0 + h1, s6 = hs * (1.0 / 6.0);
        w0 = s6 * fma(-h1, r0, 2.0);
        w2 = s6 * fma(-h0, r1, 2.0);
        w1 = (hs - w0) - w2;              // the three weights add up to the panel's width
      }
#pragma unroll
      for (int a = 0; a < A; a++) {
        const double c = fma(w0, y2[a], fma(w1, y1[a], w2 * y[a]));
        if constexpr ((J & 1) != 0) P1[a] = fma(c, m[a], P1[a]);
        else P0[a] = fma(c, m[a], P0[a]);
        y2[a] = y1[a];
        y1[a] = y[a];
      }
      if constexpr (FIRST && J == 1) r1 = rcp_n1(h1);
      h0 = h1; r0 = r1;
    }
    x1 = tau;
    if constexpr (MASKED) tm = (k <= kcut) ? fmax(tm, tau) : tm;
    else tm = fmax(tm, tau);
  };
  auto clampk = [&](int k) { return k < kend ? k : kend; };
  using std::integral_constant;
  using std::true_type;
  using std::false_type;

  // THREE load slots in rotation over straight-line blocks of SIX layers: a slot is reloaded for the layer
  // three below right after the layer that used it, so every load is issued two layers of arithmetic before
  // its use -- the shortest distance of rt_eclipse_simpson's two-pairs-of-slots scheme -- with a quarter fewer
  // registers in flight (with two CIA slots the four-slot form spilled, and a spill's reload queues behind the
  // table loads in flight: 119 against 88 us per ten-walker launch)
  double s0[NR], s1[NR], s2[NR];
  double cfE[NC], cfO[NC];
  auto block6 = [&](auto Fc, auto Mc, int k0) {
    constexpr bool MASKED = decltype(Mc)::value;
    auto inblk = [&](int k) { return MASKED ? clampk(k) : k; };
    layer(integral_constant<int, 0>{}, Fc, Mc, k0, s0, cfE, cfO);
    load_layer(inblk(k0 + 3), s0);
    layer(integral_constant<int, 1>{}, Fc, Mc, k0, s1, cfO, cfE);
    load_layer(inblk(k0 + 4), s1);
    layer(integral_constant<int, 2>{}, Fc, Mc, k0, s2, cfE, cfO);
    load_layer(inblk(k0 + 5), s2);
    layer(integral_constant<int, 3>{}, Fc, Mc, k0, s0, cfO, cfE);
    load_layer(clampk(k0 + 6), s0);
    layer(integral_constant<int, 4>{}, Fc, Mc, k0, s1, cfE, cfO);
    load_layer(clampk(k0 + 7), s1);
    layer(integral_constant<int, 5>{}, Fc, Mc, k0, s2, cfO, cfE);
    load_layer(clampk(k0 + 8), s2);
  };
  // MIG: the lane state of the walk to / from the column's slot of RtArgs::mig_state ([column][NS][64 lanes])
  [[maybe_unused]] auto mig_slot = [&]() { return p.mig_state + ((size_t)w * p.ntiles + tile) * NS * 64 + threadIdx.x; };
  const int kfirst = RESUME ? kres : 0;
  if constexpr (RESUME) {
    const double *const mst = mig_slot();
    auto get = [&](int s) { return __hip_atomic_load(mst + s * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    s_even = get(0); eprev = get(1); e2 = get(2); x1 = get(3); h0 = get(4); r0 = get(5); tm = get(6); nprev = get(7);
#pragma unroll
    for (int a = 0; a < A; a++) { y1[a] = get(8 + a); y2[a] = get(8 + A + a); P0[a] = get(8 + 2 * A + a); P1[a] = get(8 + 3 * A + a); }
  }
  load_layer(clampk(kfirst), s0);
  load_layer(clampk(kfirst + 1), s1);
  load_layer(clampk(kfirst + 2), s2);
  if constexpr (SCHED != 0) {
#pragma unroll
    for (int j = 0; j < NC; j++) cfE[j] = sC[kfirst * NC + j];
  }
  // some ray of this lane is still alive (the one with the largest threshold goes last)
  auto any_active = [&]() {
    unsigned long long mk = __ballot(tm <= thr_max);
    asm volatile("" : "+s"(mk));
    return mk != 0ull;
  };
  int kw = kBlk;   // layers walked (whole blocks)
  if constexpr (MIG) {
    // hands the column over at the block that begins with layer k0, if a wave of this CU waits for one and this SIMD keeps
    // a wave at work
    auto try_donate = [&](int k0) -> bool {
      unsigned cu, simd;
      mig_where(cu, simd);
      unsigned *const word = mig_cu_word(ctl, cu);
      unsigned long long *const slots = mig_cu_slots(ctl, cu);
      const unsigned long long tag = (unsigned long long)p.mig_epoch << 32;
      // a slot somebody waits on becomes this donor's (one 128-byte line holds the CU's slots: a lane each)
      unsigned long long sv = 0;
      if (threadIdx.x < kMigSlots) sv = __hip_atomic_load(slots + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long waits = __ballot(sv == (tag | kMigWaiting));
      BARTRT_MIG_COUNT(10);
      if (!waits) { BARTRT_MIG_COUNT(11); return false; }
      const int si = __builtin_ctzll(waits);
      // this SIMD keeps a wave at work (the count is taken down first: of two waves of a SIMD only one may go)
      const unsigned old = mig_add_ret(word, 0u - (1u << (4 * simd)));
      if ((((old >> (4 * simd)) & 15u) < 2u && !p.mig_force) || !mig_slot_cas(slots + si, tag | kMigWaiting, tag | kMigReserved)) {
        if (((old >> (4 * simd)) & 15u) < 2u) BARTRT_MIG_COUNT(12); else BARTRT_MIG_COUNT(13);
        mig_add(word, 1u << (4 * simd));
        return false;
      }
      double *const mst = mig_slot();
      auto put = [&](int s, double v) { __hip_atomic_store(mst + s * 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      put(0, s_even); put(1, eprev); put(2, e2); put(3, x1); put(4, h0); put(5, r0); put(6, tm); put(7, nprev);
#pragma unroll
      for (int a = 0; a < A; a++) { put(8 + a, y1[a]); put(8 + A + a, y2[a]); put(8 + 2 * A + a, P0[a]); put(8 + 3 * A + a, P1[a]); }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // state and log are written through: visible before the slot says so
      if (threadIdx.x == 0) {
        __hip_atomic_store(slots + si, tag | kMigColumn | (unsigned)((k0 << 20) | (w * p.ntiles + tile)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_fetch_add(ctl + kMigMoves, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return true;
    };
    // The word a walking wave looks at -- its CU's counts of waves at work and of waves that wait -- is asked for one look
    // ahead and lands in LDS by itself (global_load_lds, sc1): no register lives across the blocks for it, and the look
    // itself reads LDS.  (A hint: try_donate decides with atomics.)
    // (Written as assembly: the compiler would guard every later LDS read -- the layer records -- against a load into
    // LDS it knows of with a wait for ALL loads in flight, the table rows fetched ahead among them.  The order is kept by
    // hand: a look comes at least one block of layers after its question, and that block's own waits on younger
    // loads have retired the older ones.)
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    const unsigned mbox_lds = (unsigned)(size_t)(lds_u32 *)mbox;
    auto ask = [&]() {
      unsigned cu, simd;
      mig_where(cu, simd);
      const unsigned *const g0 = mig_cu_word(ctl, cu);
      unsigned m0_was;   // (M0 holds the LDS address of such a load; the compiler's own use of it is restored)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %3, %1 sc1\n\ts_mov_b32 m0, %0"
                   : "=&s"(m0_was) : "s"(g0), "s"(mbox_lds), "v"(0u) : "memory");
    };
    if (kcut >= kBlk - 1) {
      int k0 = kfirst;
      bool alive = true;     // (a column is handed over alive, with a whole block inside it ahead)
      if constexpr (!RESUME) {
        block6(true_type{}, false_type{}, 0);
        k0 = kBlk;
        alive = any_active();
      }
      mbox[threadIdx.x] = 0u;
      ask();
      // (the walk between two looks is the loop of the kernel that does not migrate, bounded by one more scalar: what a
      // look needs lives outside it)
      for (int klook = k0 + p.mig_cb * kBlk;; klook += p.mig_cb * kBlk) {
        while (alive & (k0 + kBlk - 1 <= kcut) & (k0 < klook)) {
          block6(false_type{}, false_type{}, k0);
          k0 += kBlk;
          alive = any_active();
        }
        if (!(alive & (k0 + kBlk - 1 <= kcut))) break;
#ifdef BARTRT_MIG_NOLOOK
        continue;
#endif
        unsigned cu, simd;
        mig_where(cu, simd);
        const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)mbox[0]);
        const bool want = ((wv >> 16) & 255u) > 0u && (((wv >> (4 * simd)) & 15u) >= 2u || p.mig_force);
        if (((wv >> 16) & 255u) > 0u) { BARTRT_MIG_COUNT(8); BARTRT_MIG_COUNT(16 + (int)(((wv >> (4 * simd)) & 15u) < 3u ? ((wv >> (4 * simd)) & 15u) : 3u)); BARTRT_MIG_COUNT(20 + (k0 / 12 < 9 ? k0 / 12 : 9)); }
        if (want) BARTRT_MIG_COUNT(9);
        ask();
        if (want && try_donate(k0)) return false;
      }
      kw = k0;
      if (alive & (k0 <= kend)) {
        block6(false_type{}, true_type{}, k0);
        kw = k0 + kBlk;
      }
    } else {
      if constexpr (!RESUME) block6(true_type{}, true_type{}, 0);
    }
  } else
  if (kcut >= kBlk - 1) {
    block6(true_type{}, false_type{}, 0);
    int k0 = kBlk;
    bool alive = any_active();
    if (alive & (k0 + kBlk - 1 <= kcut)) {
      do {
        block6(false_type{}, false_type{}, k0);
        k0 += kBlk;
        alive = any_active();
      } while (alive & (k0 + kBlk - 1 <= kcut));
    }
    kw = k0;
    if (alive & (k0 <= kend)) {   // the column's last, partial block (at most kBlk layers are left: kend <= kcut + 1)
      block6(false_type{}, true_type{}, k0);
      kw = k0 + kBlk;
    }
  } else {
    block6(true_type{}, true_type{}, 0);
  }

  // ---- after the walk: per ray angle the sum of its parity, its padded panel, the deck's surface term
  double F = 0.0;
  {
    // an event on the last layer walked
    double mfin[A];
    const double nnow = alive_flags<A>(p, tm, mfin);
    log_event(nnow, kw - 1);
    const double tauend = x1;   // tau(kend) when the wave reached the column's end (the table's overrun entries)
    double Bend = 0.0;
    if (deck_on) Bend = bnum * rcp_n1(exp_rt(fmin(sC[kend * NC + 1] * nu, 700.0)) - 1.0);
    const bool anydied = __any(nnow < (double)A);
    int kev[A];
#pragma unroll
    for (int sl = 0; sl < A; sl++) kev[sl] = -1;
    if (anydied) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the log's stores have reached L2
#pragma unroll
      for (int sl = 0; sl < A; sl++)
        kev[sl] = __builtin_amdgcn_raw_buffer_load_b32(rs_log, (int)(log_k0 + sl * nth * 4u), 0, kLogLd /* glc; MIG: sc1 */);
    }
#pragma unroll
    for (int a = 0; a < A; a++) {
      const bool died = !(tm <= p.thr[a]);
      // the ray's event: the last one logged in a slot at or below its rank in the order of dying
      int kd = kend, slot = 0;
      const int rank = p.drank[a];
#pragma unroll
      for (int sl = 0; sl < A; sl++) {
        const bool take = sl <= rank && kev[sl] >= 0;
        kd = take ? kev[sl] : kd;
        slot = take ? sl : slot;
      }
      // the last point's index: the padded one (kd + 1) for a ray that died on layer kd, else kend
      const bool odd_end = died ? ((kd + 1) & 1) != 0 : (kend & 1) != 0;
      double S = odd_end ? P1[a] : P0[a];
      if (__any(died)) {
        // panel (kd - 1, kd, pad): h0 = tau(kd) - tau(kd - 1), h1 = mu_a (one unit of slant depth), y(pad) = 0
        const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_log, (int)((unsigned)slot * nth * 16u + threadIdx.x * 16u), 0, kLogLd);
        const double taud = __builtin_bit_cast(double, (v2u_t){v.x, v.y});
        const double hd = __builtin_bit_cast(double, (v2u_t){v.z, v.w});
        const int kdc = died ? kd : 1, kp = kdc > 0 ? kdc - 1 : 0;
        const double mu = p.mu[a];
        double xe[4], ee[4];
        xe[0] = -fmin(taud, tcap) * p.invmu[a];
        xe[1] = -fmin(taud - hd, tcap) * p.invmu[a];
        xe[2] = fmin(sC[kdc * NC + 1] * nu, 700.0);
        xe[3] = fmin(sC[kp * NC + 1] * nu, 700.0);
        exp_rt_n<4>(xe, ee);
        const double yd = bnum * rcp_n1(ee[2] - 1.0) * ee[0];
        const double yp = bnum * rcp_n1(ee[3] - 1.0) * ee[1];
        const double rd = rcp_n1(hd), hs = hd + mu, s6 = hs * (1.0 / 6.0);
        const double wp = s6 * fma(-mu, rd, 2.0);
        const double wd = s6 * (hs * hs) * (rd * p.invmu[a]);
        const double pad = fma(wp, yp, wd * yd);
        S += died ? pad : 0.0;
      }
      F = fma(p.wq[a], S, F);
      double Idk = 0.0;
      if (deck_on) {
        // an opaque deck this ray reached below its cut emits as a surface: B(kend) exp(-tau(kend) / mu_a)
        const double xd = tauend * p.invmu[a];
        const bool deck = !died && !(xd > p.toomuch);
        const double Ed = exp_rt(fmax(-fmin(tauend, tcap) * p.invmu[a], kExpMin));
        Idk = deck ? Bend * Ed : 0.0;
        F = fma(p.wgt[a], Idk, F);
      }
      if constexpr (OUT) {
        if (p.intens_out && valid) p.intens_out[(size_t)a * W + i] = fma(p.invmu[a], S, Idk);
      }
    }
  }
  if (__any(!(fabs(F) < __builtin_huge_val()))) {
    // a zero-width panel somewhere in this wave (or an overflow): ray by ray, one layer at a time.  Only the lanes
    // whose own result is not finite TAKE the slow walk's numbers: a finite lane keeps the bits of the fast walk, so
    // a wavenumber's result does not depend on which other wavenumbers share its wave (blocks of a sharded grid
    // start their tiles elsewhere: the concatenated blocks stay the unsharded spectrum bit for bit)
    const double F_fast = F;
    const bool fast_ok = fabs(F) < __builtin_huge_val();
    double se = 0.0, ep = 0.0, ep2 = 0.0, tau = 0.0;
    SlantRay<kIntegSimpson> ray[A];
    bool alive_a[A];
#pragma unroll
    for (int a = 0; a < A; a++) alive_a[a] = true;
    bool act = true;
    for (int k = 0; k <= kend; k++) {
      double r[NR];
      load_layer(k, r);
      const double *c = sC + k * NC;
      double e = fma(c[2 + 2 * M + 2 * C], nu4, c[3 + 2 * M + 2 * C]);
#pragma unroll
      for (int j = 0; j < NLD; j++) e = fma(c[2 + j], r[j], e);
      if constexpr (EXT) e += r[NLD];
      const double *wk = sW + 4 * k;
      double t = tau;
      if (k & 1) t = fma(ep + e, wk[3], se);
      else if (k >= 2) t = fma(wk[0], ep2, fma(wk[1], ep, fma(wk[2], e, se)));
      if (act) {   // frozen once no ray is alive, as TauColumn<kIntegSimpson> keeps it
        tau = t;
        if (!(k & 1) && k >= 2) se = t;
      }
      ep2 = ep;
      ep = e;
      const double B = bnum * rcp_n1(exp_rt(fmin(c[1] * nu, 700.0)) - 1.0);
      bool any = false;
#pragma unroll
      for (int a = 0; a < A; a++) {
        const double x = tau * p.invmu[a];
        ray[a].point(alive_a[a], x, B, exp_rt(fmax(-fmin(tau, tcap) * p.invmu[a], kExpMin)));
        alive_a[a] = alive_a[a] && !(x > p.toomuch);
        any = any || alive_a[a];
      }
      act = any;
      if (!__any(act)) break;
    }
    F = 0.0;
#pragma unroll
    for (int a = 0; a < A; a++) {
      const double Ia = ray[a].result(deck_on && alive_a[a], L);
      F += p.wgt[a] * Ia;
      if constexpr (OUT) {
        if (p.intens_out && valid && !fast_ok) p.intens_out[(size_t)a * W + i] = Ia;
      }
    }
    F = fast_ok ? F_fast : F;
  }
  if constexpr (OUT) {
    if (p.tau_out && valid) {
      for (int kk = out_last + 1; kk < L; kk++) p.tau_out[(size_t)i * L + kk] = out_tau;
      p.last_out[i] = out_last;
    }
  }
  if (valid) p.spec[(size_t)w * W + i] = F;
  if (p.walked_out && threadIdx.x == 0)  // diagnostics: layers this wave walked (bench.py's byte model)
    p.walked_out[(size_t)w * p.ntiles + tile] = (kw < kend + 1 ? kw : kend + 1);
  return true;
}

// MIG: what a wave does after it has finished a column.  While another wave of its SIMD is at work, or workgroups are
// still to come, it leaves; else it takes a ticket and waits for a column handed over by a SIMD that walks two, stages
// that column's walker and walks on from the hand-over layer -- until every column of the launch is finished.  (Its own
// function, not inlined: the registers of the walk from the top are allotted as if this were not there.  It reads the
// kernel's argument block where the kernel does -- the kernarg segment, scalar loads -- instead of taking a reference to
// it: a reference would force a copy in scratch, and every value read from that copy would count as divergent.)
#ifdef BARTRT_MIG_INLINE
template <int AT, int MT, int CT, bool SQ, int SCHED>
__device__ __forceinline__ void slant_take(const RtArgs &p, double *smem, int w, const int bid) {
  constexpr int NC = 4 + 2 * MT + 2 * CT, NI = 1 + CT;
#else
template <int AT, int MT, int CT, bool SQ, int SCHED>
__device__ __attribute__((noinline)) void slant_take(double *smem, int w, const int bid) {
  constexpr int NC = 4 + 2 * MT + 2 * CT, NI = 1 + CT;
  typedef __attribute__((address_space(4))) const RtArgs karg_t;
  // (a function that is not a kernel is not handed the kernarg segment's address, but the address of the implicit
  // arguments that follow the kernel's one explicit argument in it, eight-byte aligned)
  const char __attribute__((address_space(4))) *const ia = (const char __attribute__((address_space(4))) *)__builtin_amdgcn_implicitarg_ptr();
  RtArgs p;   // a copy the optimiser keeps in scalar registers, like the kernel's own view of its arguments
  __builtin_memcpy(&p, (const RtArgs *)(karg_t *)(ia - ((sizeof(RtArgs) + 7) & ~(size_t)7)), sizeof(RtArgs));
#endif
  unsigned *const ctl = p.mig_ctl;
  const unsigned long long tag = (unsigned long long)p.mig_epoch << 32;
  for (;;) {
    unsigned cu, simd;
    mig_where(cu, simd);
    unsigned *const word = mig_cu_word(ctl, cu);
    unsigned long long *const slots = mig_cu_slots(ctl, cu);
    const unsigned old = mig_add_ret(word, 0u - (1u << (4 * simd)));   // this wave is not at work any more
    if (((old >> (4 * simd)) & 15u) != 1u) return;                    // another wave of this SIMD is
    if (mig_at_work(old) <= 1) return;                                // nothing else is at work on this CU: nobody to take from
    // a free slot of the CU becomes "a wave waits" (free: another launch's tag, or this launch's free mark)
    int mine = -1;
    for (int tries = 0; tries < 4 && mine < 0; tries++) {
      unsigned long long sv = tag | kMigReserved;
      if (threadIdx.x < kMigSlots) sv = __hip_atomic_load(slots + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long free_ = __ballot((sv >> 32) != p.mig_epoch || (unsigned)sv == kMigFree);
      if (!free_) return;
      const int si = __builtin_ctzll(free_);
      const unsigned long long cur = __shfl(sv, si);
      if (mig_slot_cas(slots + si, cur, tag | kMigWaiting)) mine = si;
    }
    if (mine < 0) return;
    BARTRT_MIG_COUNT(14);
    mig_add(word, 1u << 16);
    // the first lane polls the slot (and, now and then, whether anything is still at work on this CU)
    unsigned got = kMigFree;
    for (int spin = 0;; spin++) {
      unsigned lo = 0;
      if (threadIdx.x == 0) lo = (unsigned)__hip_atomic_load(slots + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lo = (unsigned)__builtin_amdgcn_readfirstlane((int)lo);
      if (lo & kMigColumn) { got = lo; break; }
      const bool lost = spin > kMigSpinMax;
      if (lo == kMigWaiting && (lost || ((spin & 3) == 3 && mig_at_work(mig_load(word)) == 0))) {
        // nobody is left to hand anything over: the slot goes back -- unless a donor has just taken it
        if (mig_slot_cas(slots + mine, tag | kMigWaiting, tag | kMigFree)) {
          if (lost && threadIdx.x == 0) __hip_atomic_store(ctl + kMigError, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __builtin_amdgcn_s_sleep(24);
    }
    mig_add(word, 0u - (1u << 16));
    if (!(got & kMigColumn)) return;
    if (threadIdx.x == 0) __hip_atomic_store(slots + mine, tag | kMigFree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mig_add(word, 1u << (4 * simd));
    const int u = (int)(got & 0xfffffu), kres = (int)((got >> 20) & 0x7ffu), tile = u % p.ntiles;
    if (u / p.ntiles != w) {
      w = u / p.ntiles;
      __syncthreads();
      slant_stage<NC, NI>(p, smem, w);
    }
    if (!slant_column<AT, MT, CT, SQ, SCHED, false, false, true, true>(p, smem, tile, w, kres)) return;   // handed over again
  }
}

template <int AT, int MT, int CT, bool SQ, int SCHED = 1, bool EXT = false, bool OUT = false, bool MIG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OUT ? 1 : 2, BARTRT_WPE)))
void rt_eclipse_simpson_slant(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int NC = 4 + 2 * MT + 2 * CT, NI = 1 + CT;
  int bid = blockIdx.x;
  if (p.nprep > 0) {   // the head of the grid prepares the NEXT batch's layer records (RtArgs::nprep)
    if (bid < p.nprep) { prep_block(p.prep_next, bid, smem); return; }
    bid -= prep_slots(p.nprep);
    if (bid < 0) return;
  }
  int tile, w;
  block_to_work(bid, p.nwalkers, tile, w);
  if constexpr (!MIG) {
    if (tile >= p.ntiles) return;
    slant_stage<NC, NI>(p, smem, w);
    slant_column<AT, MT, CT, SQ, SCHED, EXT, OUT, false, false>(p, smem, tile, w, -1);
  } else {
    unsigned *const ctl = p.mig_ctl;
    if (tile < p.ntiles) {
      unsigned cu, simd;
      mig_where(cu, simd);
      mig_add(mig_cu_word(ctl, cu), 1u << (4 * simd));
      slant_stage<NC, NI>(p, smem, w);
      const bool finished = slant_column<AT, MT, CT, SQ, SCHED, false, false, true, false>(p, smem, tile, w, -1);
#if defined(BARTRT_MIG_NOEND)
      (void)finished;
#elif defined(BARTRT_MIG_INLINE)
      if (finished) slant_take<AT, MT, CT, SQ, SCHED>(p, smem, w, bid);
#else
      if (finished) slant_take<AT, MT, CT, SQ, SCHED>(smem, w, bid);
#endif
    }
  }
}

// the builds (rt_eclipse_i1s_ilp.hip): ray grids of five angles and of the other sizes, the line-by-line hand-off
// (the launchers are declared with the other specialised kernels' in rt_eclipse.hpp)

}  // namespace bartrt
