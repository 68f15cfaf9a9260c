// Single-wave eclipse kernel of integration rule 1 (SURVEY.md App. A-4 as recalled: the
// Simpson / trapezoid hybrid with the zero-padded integrand; integ.hpp states the rule, the
// parity tests hold it to its scalar CPU restatement): the same walk as rt_eclipse_fast (rt_eclipse.hpp) -- one lane per
// (walker, wavenumber), buffer loads two layers ahead, layer records from LDS -- with the
// rule's accumulation arranged so that a layer costs 15 fp64 operations more than rule 0's
// instead of 43:
//
//  * angle quadrature first.  I_a = (1/mu_a) sum_k W_k(tau) B_k E_{a,k} with Simpson weights
//    W_k that depend on the tau grid only, so F = sum_a w_a I_a = sum_k W_k Y_k with ONE
//    integrand Y_k = B_k sum_a (w_a / mu_a) E_{a,k} per layer (RtArgs::wq) instead of A.
//  * the panel (k-2, k-1, k) in differences: with h0 = x_{k-1} - x_{k-2}, h1 = x_k - x_{k-1},
//    d_k = Y_k - Y_{k-1}:
//        6 c_k = (h0 + h1) [ 6 Y_{k-1} + (h1/h0 - 2) d_{k-1} + (2 - h0/h1) d_k ]
//    -- ONE reciprocal per layer (1/h1; it is the next layer's 1/h0), 11 operations.
//  * the parity of a layer index is a compile-time constant of the unrolled four-layer
//    block: panels that end on even / odd points go to P0 / P1 without selects; the first
//    block (whose second point closes a trapezoid instead of a panel) is peeled.
//  * the padded point (integrand 0, one unit of tau past the last point) is the FIRST DEAD
//    LAYER of a lane: there h1 is forced to 1 and Y to 0 and the panel is counted once
//    more; a lane that died on the last layer of a block gets it in the epilogue.
//  * no per-layer handling of zero-width panels (h0 == 0 or h1 == 0; the restatement takes
//    the two trapezoids): a zero width makes the reciprocal infinite and the lane's sum
//    non-finite, which is sticky; a wave that ends with a non-finite flux recomputes its
//    columns with the plain accumulator (ColumnFlux<kIntegSimpson>, integ.hpp), case
//    analysis and all.  Zero-width tau intervals need two adjacent layers of exactly zero
//    extinction: the known-answer tests have them, opacity tables do not.
//  * optical depth: sW[4 k + (0,1,2)] = Simpson weights of the radius panel that ends on
//    an even layer k, sW[4 k + 3] = half the path length of (k-1, k) for odd k, computed
//    once per workgroup with two reciprocals per panel; zero past the column's last layer
//    kend, except that the entry after an odd kend re-adds kend's trapezoid, so that the
//    block's masked overrun layers leave tau at tau(kend) (the cloud deck's surface term
//    reads it after the loop).  tau is not frozen on dead lanes (no select): their terms
//    are masked where they are added.
#pragma once
#include "integ.hpp"
#include "kernels.hpp"

#ifndef __HIPCC_RTC__
#include <type_traits>
#endif

namespace bartrt {

__host__ __device__ inline size_t simpson4_lds_doubles(int L) { return 4 * (size_t)(L + kSimpsonPad); }

__device__ __forceinline__ void simpson_radius_table(double *sW, const double *sC, int NC, int L, int kend, int tid,
                                                     int nthreads) {
  for (int k = tid; k < L + kSimpsonPad; k += nthreads) {
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, hd = 0.0;
    if (k <= kend && k >= 1) {
      const double h1 = sC[k * NC];
      if (k & 1) {
        hd = 0.5 * h1;
      } else {
        const double h0 = sC[(k - 1) * NC], hs = h0 + h1;
        if (h0 == 0.0 || h1 == 0.0) {
          w0 = 0.5 * h0; w1 = 0.5 * hs; w2 = 0.5 * h1;
        } else {
          const double r0 = rcp_core(h0), r1 = rcp_core(h1), s6 = hs * (1.0 / 6.0);
          w0 = s6 * (2.0 - h1 * r0);
          w1 = s6 * (hs * hs * (r0 * r1));
          w2 = s6 * (2.0 - h0 * r1);
        }
      }
    } else if (k == kend + 1 && (kend & 1) && k < L + kSimpsonPad) {
      // even overrun layer after an odd last layer: s_even + hd (e_{kend-1} + e_kend) = tau(kend)
      w0 = w1 = 0.5 * sC[kend * NC];
    }
    sW[4 * k] = w0; sW[4 * k + 1] = w1; sW[4 * k + 2] = w2; sW[4 * k + 3] = hd;
  }
}

template <int AT, int MT, int CT, bool SQ, int SCHED = 0, bool EXT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SCHED ? 2 : 3, BARTRT_WPE)))
void rt_eclipse_simpson(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 4 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C;
  constexpr int NR = NLD + (EXT ? 1 : 0) > 0 ? NLD + (EXT ? 1 : 0) : 1;
  constexpr int AE = SQ ? A - 1 : A;  // transmittances that need an exponential
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
  // `toomuch` ends a column only where a deeper layer exists (nothing follows the bottom
  // layer: a lane that passes the cut there ends like one that did not)
  const int kcut = kend < L - 2 ? kend : L - 2;

  // optical depth: tau of the last even layer, the last two extinctions
  double s_even = 0.0, eprev = 0.0, e2 = 0.0;
  // intensity integral: abscissa, integrand and interval of the previous point, P0 / P1 = six
  // times the sums of the panels that end on even / odd points
  double x1 = 0.0, y1 = 0.0, dprev = 0.0, h0 = 1.0, r0 = 1.0, P0 = 0.0, P1 = 0.0;
  bool active = true;       // the lane has not passed toomuch
  bool live_prev = false;   // the previous layer counted for this lane
  bool oddf = false;        // the lane's last point has an odd index
  double mprev = 1.0;       // 1 if the previous layer counted (unmasked blocks)

  // one layer.  J = position in the four-layer block (the parity of k), FIRST = the block of
  // k0 = 0, MASKED = the column's last block: layers past kend are walked with clamped inputs
  // and masked; the blocks above it lie inside the column (k0 + 3 <= kcut) and carry no range
  // logic at all: a layer counts for a lane iff the lane was active when the PREVIOUS layer
  // began (it is still active, or this is its first dead layer: the padded point).
  auto layer = [&](auto Jc, auto Fc, auto Mc, int k0, const double (&r)[NR], const double (&cf)[NC],
                   double (&cfn)[NC]) {
    constexpr int J = decltype(Jc)::value;
    constexpr bool FIRST = decltype(Fc)::value, MASKED = decltype(Mc)::value;
    const int k = k0 + J;
    auto read_rec = [&](int kk, double (&c_)[NC]) {
      const double *c = sC + ((MASKED || J == 3) ? (kk < kend ? kk : kend) : kk) * NC;
#pragma unroll
      for (int j = 0; j < NC; j++) c_[j] = c[j];
    };
    bool live = active;
    if constexpr (MASKED) live = active & (k <= kend);
    // the ILP build (one or two waves per SIMD) reads the next layer's record one layer ahead;
    // the occupancy build reads its own at the point of use: 28 registers fewer, and with them
    // three resident waves per SIMD instead of two
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
    // Planck exponent and the slant-path exponents in one interleaved batch
    const double tcl = fmin(tau, tcap);
    double xs[AE + 1], ex[AE + 1], es[A];
    xs[AE] = fmin(cf[1] * nu, 700.0);
#pragma unroll
    for (int a = 0; a < AE; a++) xs[a] = -tcl * p.invmu[a];
    exp_rt_n<AE + 1>(xs, ex);
#pragma unroll
    for (int a = 0; a < AE; a++) es[a] = ex[a];
    if (SQ) es[A - 1] = ex[0] * ex[0];
    const double B = bnum * rcp_n1(ex[AE] - 1.0);
    const double m = live ? 1.0 : 0.0;
    const double y = (B * m) * angle_sum_q<A>(p, es);   // 0 on the padded point
    if constexpr (FIRST && J == 0) {
      y1 = y;
    } else {
      // counted: a live layer, or the padded point = the first dead layer of a lane that
      // passed toomuch
      double mc = mprev;
      if constexpr (MASKED) mc = (live | (live_prev & !active)) ? 1.0 : 0.0;
      const double h1 = live ? tau - x1 : 1.0;
      const double r1 = rcp_n1(h1);
      const double d = y - y1;
      double u;
      if constexpr (FIRST && J == 1) {
        u = 3.0 * h1 * (y1 + y);   // the first interval: a trapezoid
      } else {
        double t = fma(h1, r0, -2.0) * dprev;
        t = fma(fma(-h0, r1, 2.0), d, t);
        t = fma(6.0, y1, t);
        u = (h0 + h1) * t;
      }
      if constexpr ((J & 1) != 0) P1 = fma(u, mc, P1);
      else P0 = fma(u, mc, P0);
      h0 = h1; r0 = r1; dprev = d; y1 = y;
    }
    x1 = tau;
    live_prev = live;
    mprev = m;
    bool over = active & (tau > p.toomuch);
    if constexpr (MASKED) over = over & (k <= kcut);
    if constexpr ((J & 1) == 0) oddf = oddf | over;   // its last point is the padded one, k + 1
    active = active & !over;
  };
  auto clampk = [&](int k) { return k < kend ? k : kend; };
  using std::integral_constant;
  using std::true_type;
  using std::false_type;

  // two pairs of slots; each pair is reloaded two layers before it is used and the loads
  // that cross the loop's back edge were issued two layers earlier
  double a0[NR], a1[NR], b0[NR], b1[NR];
  double cfE[NC], cfO[NC];
  auto block4 = [&](auto Fc, auto Mc, int k0) {
    constexpr bool MASKED = decltype(Mc)::value;
    load_layer(MASKED ? clampk(k0 + 2) : k0 + 2, b0);
    load_layer(MASKED ? clampk(k0 + 3) : k0 + 3, b1);
    layer(integral_constant<int, 0>{}, Fc, Mc, k0, a0, cfE, cfO);
    layer(integral_constant<int, 1>{}, Fc, Mc, k0, a1, cfO, cfE);
    load_layer(clampk(k0 + 4), a0);
    load_layer(clampk(k0 + 5), a1);
    layer(integral_constant<int, 2>{}, Fc, Mc, k0, b0, cfE, cfO);
    layer(integral_constant<int, 3>{}, Fc, Mc, k0, b1, cfO, cfE);
  };
  load_layer(clampk(0), a0);
  load_layer(clampk(1), a1);
  if constexpr (SCHED != 0) {
#pragma unroll
    for (int j = 0; j < NC; j++) cfE[j] = sC[j];
  }
  // (the ballot goes through an opaque move: left visible, the compiler threads the exit test
  // into the middle of the block -- "no lane survived layer k0 + 1" -- and the split block
  // loses the interleaving of the next layers' loads with this layer's arithmetic)
  auto any_active = [&]() {
    unsigned long long mk = __ballot(active);
    asm volatile("" : "+s"(mk));
    return mk != 0ull;
  };
  int kw = 4;   // layers walked (whole blocks)
  if (kcut >= 3) {
    block4(true_type{}, false_type{}, 0);
    // one exit test per block (k0 and the ballot combined): two exits make the compiler
    // evaluate the break early and split the block
    int k0 = 4;
    bool alive = any_active();
    if (alive & (k0 + 3 <= kcut)) {
      do {
        block4(false_type{}, false_type{}, k0);
        k0 += 4;
        alive = any_active();
      } while (alive & (k0 + 3 <= kcut));
    }
    kw = k0;
    if (alive & (k0 <= kend)) {   // the column's last, partial block
      block4(false_type{}, true_type{}, k0);
      kw = k0 + 4;
    }
  } else {
    block4(true_type{}, true_type{}, 0);
  }

  // a lane that passed toomuch on the last layer walked: its padded point is point k + 1 (even)
  const bool padm = live_prev && !active;
  if (__any(padm)) {
    double t = (r0 - 2.0) * dprev;
    t = fma(h0 - 2.0, y1, t);
    t = fma(6.0, y1, t);
    P0 += padm ? (h0 + 1.0) * t : 0.0;
  }
  // lanes that never passed the cut end on the column's last layer
  oddf = oddf || (active && (kend & 1));
  double F = (oddf ? P1 : P0) * (1.0 / 6.0);
  if (deck_on) {
    // an opaque deck reached below toomuch emits as a surface: B(kend) sum_a w_a E_a(tau(kend));
    // x1 is tau(kend) here (see the table's overrun entry)
    const bool deck = active && !(x1 > p.toomuch);
    if (__any(deck)) {
      const double tcl = fmin(x1, tcap);
      double xs[A + 1], ex[A + 1], es[A];
      xs[A] = fmin(sC[kend * NC + 1] * nu, 700.0);
#pragma unroll
      for (int a = 0; a < A; a++) xs[a] = -tcl * p.invmu[a];
      exp_rt_n<A + 1>(xs, ex);
#pragma unroll
      for (int a = 0; a < A; a++) es[a] = ex[a];
      const double yd = bnum * rcp_n1(ex[A] - 1.0) * angle_sum<A>(p, es);
      F += deck ? yd : 0.0;
    }
  }
  if (__any(!(fabs(F) < __builtin_huge_val()))) {
    // a zero-width panel somewhere in this wave (or an overflow): the plain accumulator,
    // one layer at a time; the lanes whose own result is finite keep it (rt_eclipse_s1s.hpp)
    const double F_fast = F;
    const bool fast_ok = fabs(F) < __builtin_huge_val();
    double se = 0.0, ep = 0.0, ep2 = 0.0, tau = 0.0;
    ColumnFlux<kIntegSimpson, A> cf(p);
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
      if (act) {   // frozen on dead lanes, as TauColumn<kIntegSimpson> keeps it
        tau = t;
        if (!(k & 1) && k >= 2) se = t;
      }
      ep2 = ep;
      ep = e;
      double es[A];
#pragma unroll
      for (int a = 0; a < A; a++) es[a] = exp_rt(fmax(-fmin(tau, tcap) * p.invmu[a], kExpMin));
      const double B = bnum * rcp_n1(exp_rt(fmin(c[1] * nu, 700.0)) - 1.0);
      cf.layer(p, A, act, 0.0, tau, 0.0, B, es);
      act = act && !(tau > p.toomuch);
      if (!__any(act)) break;
    }
    F = cf.flux(p, A, deck_on && act, 0.0, L);
    F = fast_ok ? F_fast : F;
  }
  if (valid) p.spec[(size_t)w * W + i] = F;
  if (p.walked_out && threadIdx.x == 0)  // diagnostics: layers this wave walked (bench.py's byte model)
    p.walked_out[(size_t)w * p.ntiles + tile] = (kw < kend + 1 ? kw : kend + 1);
}

// the ILP-scheduled build (rt_eclipse_i1_ilp.hip), and the line-by-line hand-off
#ifndef __HIPCC_RTC__
bool launch_rt_simpson_ilp(const RtArgs &b, bool sq, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err);
bool launch_rt_simpson_ext(const RtArgs &b, bool sq, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err);
#endif

}  // namespace bartrt
