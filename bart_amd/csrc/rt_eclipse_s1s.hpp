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

// OUT: also writes the optical depths tau_out[W][L] / last_out[W] (the deepest layer a ray of the sample reaches;
// deeper layers repeat its depth) and the per-ray intensities intens_out[A][W] of a single walker -- what `tau.dat`
// and `outintens` hold (code/cf.py:46-94 reads them back) -- from the same walk, instead of a second launch of the
// generic kernel.
// (the OUT build is a once-per-run diagnostic launch of one walker: it takes the registers of a whole SIMD -- one
// wave per SIMD -- instead of spilling the output bookkeeping)
template <int AT, int MT, int CT, bool SQ, int SCHED = 1, bool EXT = false, bool OUT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OUT ? 1 : 2, BARTRT_WPE)))
void rt_eclipse_simpson_slant(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 4 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C;
  constexpr int NR = NLD + (EXT ? 1 : 0) > 0 ? NLD + (EXT ? 1 : 0) : 1;
  constexpr int AE = SQ ? A - 1 : A;  // transmittances that need an exponential
  constexpr int kBlk = 6;             // layers per straight-line block (even: a layer's parity is its position)
  static_assert(kBlk - 2 <= kSimpsonPad, "the radius table's overrun entries");
  const int L = p.L, W = p.W;
  int bid = blockIdx.x;
  if (p.nprep > 0) {   // the head of the grid prepares the NEXT batch's layer records (RtArgs::nprep)
    if (bid < p.nprep) { prep_block(p.prep_next, bid, smem); return; }
    bid -= prep_slots(p.nprep);
    if (bid < 0) return;
  }
  int tile, w;
  block_to_work(bid, p.nwalkers, tile, w);
  if (tile >= p.ntiles) return;

  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  double *sWw = smem + (size_t)L * NC + (size_t)L * NI;
  const double *sW = sWw;
  stage2_to_lds(sC, p.coef + (size_t)w * L * NC, L * NC, sI, p.idx + (size_t)w * L * NI, L * NI, threadIdx.x,
                blockDim.x);
  const int kraw = p.kstop[w], kend = kstop_layer(kraw);
  const bool deck_on = kstop_deck(kraw);
  __syncthreads();
  simpson_radius_table(sWw, sC, NC, L, kend, threadIdx.x, blockDim.x);
  __syncthreads();

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
#pragma unroll
  for (int sl = 0; sl < A; sl++) __builtin_amdgcn_raw_buffer_store_b32(-1, rs_log, (int)(log_k0 + sl * nth * 4u), 0, 0);
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
    __builtin_amdgcn_raw_buffer_store_b128(v, rs_log, (int)off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(kev, rs_log, (int)offk, 0, 0);
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
  load_layer(clampk(0), s0);
  load_layer(clampk(1), s1);
  load_layer(clampk(2), s2);
  if constexpr (SCHED != 0) {
#pragma unroll
    for (int j = 0; j < NC; j++) cfE[j] = sC[j];
  }
  // some ray of this lane is still alive (the one with the largest threshold goes last)
  auto any_active = [&]() {
    unsigned long long mk = __ballot(tm <= thr_max);
    asm volatile("" : "+s"(mk));
    return mk != 0ull;
  };
  int kw = kBlk;   // layers walked (whole blocks)
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
        kev[sl] = __builtin_amdgcn_raw_buffer_load_b32(rs_log, (int)(log_k0 + sl * nth * 4u), 0, 1 /* glc */);
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
        const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_log, (int)((unsigned)slot * nth * 16u + threadIdx.x * 16u), 0, 1);
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
}

// the builds (rt_eclipse_i1s_ilp.hip): ray grids of five angles and of the other sizes, the line-by-line hand-off
// (the launchers are declared with the other specialised kernels' in rt_eclipse.hpp)

}  // namespace bartrt
