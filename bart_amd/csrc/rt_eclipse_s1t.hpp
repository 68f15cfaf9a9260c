// `cut slant`, rule 1, five ray angles: a TEAM of three waves per 64-wavenumber column.
//
// The single-wave kernel (rt_eclipse_s1s.hpp) walks a column's layers with ~200 instructions per
// layer in ONE wave.  Ten walkers at W = 1e4 are 1 570 such waves on 1 024 SIMDs: SIMDs with two
// waves take twice as long as SIMDs with one and the launch ends with them (max / mean 1.31), and
// a lone walker's 157 waves take the full serial 58 us.  Here the layer's work is cut three ways
// along its natural seams, each part a wave of the same workgroup:
//
//   wave 0  producer   layer records, table loads (three slots in rotation), extinction, optical
//                      depth by the Simpson radius table -> tau into an LDS ring (two halves of six
//                      layers); the running maximum of tau decides when every ray of every lane is dead;
//   wave 1  rays {0, 1, 4}   (4 = the squared partner of 0 when the grid has one: 0 / 60 degrees)
//   wave 2  rays {2, 3}      + the lane's death-event log (rt_eclipse_s1s.hpp)
//                      each: Planck term, its rays' transmittances, the tau panel's weights, its rays'
//                      even / odd Simpson sums.
//
// One raw workgroup barrier per six layers hands a half of the ring over (the producer fills half b + 1
// while the consumers read half b); the exit decision travels with the half.  The Planck term is
// evaluated by both consumers (17 operations twice) so that the ring carries ONE double per (lane,
// layer): 6 kB, which keeps six workgroups = 18 waves resident per CU.  4 710 shorter waves instead of
// 1 570: max / mean 1.09 at ten walkers, and a lone column's serial path is 85 instead of 206
// instructions per layer.  After the walk the consumers evaluate their rays' padded panels from the
// log and the two partial fluxes meet in LDS.
#pragma once
#include "integ.hpp"
#include "kernels.hpp"
#include "rt_eclipse_s1.hpp"

#ifndef __HIPCC_RTC__
#include <type_traits>
#endif

namespace bartrt {

constexpr int kTeamBlk = 6;   // layers per half of the ring (even: a layer's parity is its position in the half)
__host__ __device__ inline size_t team_lds_doubles() { return 2 * kTeamBlk * 64 + 2 + 64 + 64; }

template <int MT, int CT, bool SQ>
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(4, 8)))
void rt_eclipse_slant_team(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int A = 5, M = MT, C = CT, NB = kTeamBlk;
  constexpr int NC = 4 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C;
  constexpr int NR = NLD > 0 ? NLD : 1;
  static_assert(NB - 1 <= kSimpsonPad && (NB & 1) == 0, "ring half");
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
  double *sT = sWw + simpson4_lds_doubles(L);            // [2 halves][NB][64] optical depths
  int *sFlag = reinterpret_cast<int *>(sT + 2 * NB * 64);   // [2] the producer saw every ray of every lane dead
  double *sEnd = sT + 2 * NB * 64 + 2;                   // [64] tau of the last layer walked
  double *sF = sEnd + 64;                                // [64] wave 2's part of the flux
  stage2_to_lds(sC, p.coef + (size_t)w * L * NC, L * NC, sI, p.idx + (size_t)w * L * NI, L * NI, threadIdx.x, 192);
  if (threadIdx.x < 2) sFlag[threadIdx.x] = 0;
  const int kraw = p.kstop[w], kend = kstop_layer(kraw);
  const bool deck_on = kstop_deck(kraw);
  __syncthreads();
  simpson_radius_table(sWw, sC, NC, L, kend, threadIdx.x, 192);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
  const int i = tile * 64 + lane;
  const bool valid = i < W;
  const unsigned ii = valid ? (unsigned)i : (unsigned)(W - 1);
  const double nu = p.wn[ii];
  const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
  const double tcap = tau_cap(p, A);
  const int kcut = kend < L - 2 ? kend : L - 2;   // `toomuch` ends a ray only where a deeper layer exists
  const int nblk = kend / NB + 1;                 // halves that hold a layer of the column
  auto clampk = [&](int k) { return k < kend ? k : kend; };
  // LDS writes of a half complete, then meet the other waves (raw barrier: a __syncthreads() fence would also
  // drain the table loads in flight)
  auto handoff = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // the lane's event log (rt_eclipse_s1s.hpp): wave 2 writes it, both consumers read it after the walk
  const unsigned nth = 64;
  const unsigned log_bytes = nth * (unsigned)A * 20u;
  const auto rs_log = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char *>(p.slog) + ((size_t)w * p.ntiles + tile) * log_bytes, 0, (int)log_bytes, 0x00020000);
  const unsigned log_k0 = nth * (unsigned)A * 16u + (unsigned)lane * 4u;

  if (role == 0) {
    // ---------------- producer: extinction and optical depth ----------------
    const double nu4 = (nu * nu) * (nu * nu);
    const TableLoader<M, C> tab(p, ii, sI);
    auto load_layer = [&](int k, double (&r)[NR]) { tab.load(k, r); };
    double thr_max = p.thr[0];
#pragma unroll
    for (int a = 1; a < A; a++) thr_max = p.thr[a] > thr_max ? p.thr[a] : thr_max;
    double s_even = 0.0, eprev = 0.0, e2 = 0.0, tm = 0.0, tau = 0.0;
    // (registers decide how many teams a CU holds -- 128 for four waves per SIMD, five workgroups per CU: two load
    // slots reloaded one layer ahead and the layer's record read where it is used; the other waves of the SIMD
    // cover the latency the wave's own prefetch does not)
    auto layer = [&](auto Jc, int k0, const double (&r)[NR]) {
      constexpr int J = decltype(Jc)::value;
      const int k = k0 + J;
      const double *cf = sC + clampk(k) * NC;
      double e = fma(cf[2 + 2 * M + 2 * C], nu4, cf[3 + 2 * M + 2 * C]);   // Rayleigh + grey cloud
#pragma unroll
      for (int j = 0; j < NLD; j++) e = fma(cf[2 + j], r[j], e);
      const double *wk = sW + 4 * k;   // (zero weights on layer 0 and past kend: tau starts at 0 and stands still at tau(kend))
      if constexpr ((J & 1) != 0) {
        tau = fma(eprev + e, wk[3], s_even);
      } else {
        s_even = fma(wk[0], e2, fma(wk[1], eprev, fma(wk[2], e, s_even)));
        tau = s_even;
      }
      e2 = eprev;
      eprev = e;
      sT[((k0 / NB) & 1) * (NB * 64) + J * 64 + lane] = tau;
      tm = k <= kcut ? fmax(tm, tau) : tm;
    };
    using std::integral_constant;
    double s0[NR], s1[NR];
    load_layer(clampk(0), s0);
    load_layer(clampk(1), s1);
    int blk = 0;
    for (; blk < nblk; blk++) {
      const int k0 = blk * NB;
      layer(integral_constant<int, 0>{}, k0, s0);
      load_layer(clampk(k0 + 2), s0);
      layer(integral_constant<int, 1>{}, k0, s1);
      load_layer(clampk(k0 + 3), s1);
      layer(integral_constant<int, 2>{}, k0, s0);
      load_layer(clampk(k0 + 4), s0);
      layer(integral_constant<int, 3>{}, k0, s1);
      load_layer(clampk(k0 + 5), s1);
      layer(integral_constant<int, 4>{}, k0, s0);
      load_layer(clampk(k0 + 6), s0);
      layer(integral_constant<int, 5>{}, k0, s1);
      load_layer(clampk(k0 + 7), s1);
      // the exit decision travels with the half, so all three waves leave after the same barrier
      const bool stop = !__any(tm <= thr_max);
      if (lane == 0) sFlag[blk & 1] = stop ? 1 : 0;
      handoff();
      if (stop) { blk++; break; }
    }
    sEnd[lane] = tau;    // tau(kend) when the column was walked to its end
    handoff();           // ... the consumers' epilogues
    handoff();           // ... the two partial fluxes
    if (p.walked_out && lane == 0)
      p.walked_out[(size_t)w * p.ntiles + tile] = (blk * NB < kend + 1 ? blk * NB : kend + 1);
    return;
  }

  // ---------------- consumers: Planck term, transmittances, Simpson sums of their rays ----------------
  // wave 1: rays 0, 1 and A - 1 (the squared partner of ray 0 under SQ); wave 2: rays 2, 3
  const bool first = role == 1;
  constexpr int NRAY = 3;                       // register slots; wave 2 leaves the third idle
  int ra[NRAY];
  ra[0] = first ? 0 : 2; ra[1] = first ? 1 : 3; ra[2] = first ? A - 1 : 3;
  const int nray = first ? 3 : 2;
  double invmu_r[NRAY], thr_r[NRAY];
#pragma unroll
  for (int j = 0; j < NRAY; j++) { invmu_r[j] = p.invmu[ra[j]]; thr_r[j] = p.thr[ra[j]]; }
  double x1 = 0.0, h0 = 0.0, r0 = 1.0, tm = 0.0;
  double y1[NRAY], y2[NRAY], P0[NRAY], P1[NRAY];
#pragma unroll
  for (int j = 0; j < NRAY; j++) { y1[j] = y2[j] = P0[j] = P1[j] = 0.0; }
  int nprev = A;
  auto log_event = [&](int nnow, int kev) {   // (wave 2; see rt_eclipse_s1s.hpp)
    const bool ev = nnow < nprev;
    const unsigned slot = (unsigned)(A - nprev);
    const unsigned off = ev ? slot * nth * 16u + (unsigned)lane * 16u : 0x7ffffff0u;
    const unsigned offk = ev ? log_k0 + slot * nth * 4u : 0x7ffffff0u;
    v4u_t v;
    v.x = (unsigned)__double2loint(x1); v.y = (unsigned)__double2hiint(x1);
    v.z = (unsigned)__double2loint(h0); v.w = (unsigned)__double2hiint(h0);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs_log, (int)off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(kev, rs_log, (int)offk, 0, 0);
    nprev = nnow;
  };
  auto count_alive = [&]() {
    int n = 0;
#pragma unroll
    for (int a = 0; a < A; a++) n += (tm <= p.thr[a]) ? 1 : 0;
    return n;
  };
  if (!first) {
#pragma unroll
    for (int sl = 0; sl < A; sl++) __builtin_amdgcn_raw_buffer_store_b32(-1, rs_log, (int)(log_k0 + sl * nth * 4u), 0, 0);
  }
  // one layer of the half `hb` (u = its position, the parity of k)
  auto clayer = [&](auto Uc, auto Fc, int k0, const double *half) {
    constexpr int U = decltype(Uc)::value;
    constexpr bool FIRST = decltype(Fc)::value;
    const int k = k0 + U;
    const double tau = half[U * 64 + lane];
    const bool inr = k <= kend;                       // wave-uniform
    const double c2t = sC[clampk(k) * NC + 1];
    double m[NRAY];
#pragma unroll
    for (int j = 0; j < NRAY; j++) m[j] = (inr && tm <= thr_r[j]) ? 1.0 : 0.0;
    if (!first) {
      if constexpr (!(FIRST && U == 0)) log_event(count_alive(), k - 1);
    }
    const double tcl = fmin(tau, tcap);
    double y[NRAY];
    if (first) {
      if constexpr (SQ) {
        double xs[3] = {-tcl * invmu_r[0], -tcl * invmu_r[1], fmin(c2t * nu, 700.0)}, ex[3];
        exp_rt_n<3>(xs, ex);
        const double B = bnum * rcp_n1(ex[2] - 1.0);
        y[0] = B * ex[0]; y[1] = B * ex[1]; y[2] = y[0] * ex[0];
      } else {
        double xs[4] = {-tcl * invmu_r[0], -tcl * invmu_r[1], -tcl * invmu_r[2], fmin(c2t * nu, 700.0)}, ex[4];
        exp_rt_n<4>(xs, ex);
        const double B = bnum * rcp_n1(ex[3] - 1.0);
        y[0] = B * ex[0]; y[1] = B * ex[1]; y[2] = B * ex[2];
      }
    } else {
      double xs[3] = {-tcl * invmu_r[0], -tcl * invmu_r[1], fmin(c2t * nu, 700.0)}, ex[3];
      exp_rt_n<3>(xs, ex);
      const double B = bnum * rcp_n1(ex[2] - 1.0);
      y[0] = B * ex[0]; y[1] = B * ex[1]; y[2] = 0.0;
    }
    if constexpr (FIRST && U == 0) {
#pragma unroll
      for (int j = 0; j < NRAY; j++) y1[j] = y[j];
    } else {
      const double h1 = inr ? tau - x1 : 1.0;      // (a unit interval on overrun layers: tau stands still there)
      double w0, w1, w2, r1;
      if constexpr (FIRST && U == 1) {
        w0 = 0.0; w1 = w2 = 0.5 * h1;               // the first interval: a trapezoid
        r1 = rcp_n1(h1);
      } else {
        r1 = rcp_n1(h1);
        const double hs = h0 + h1, s6 = hs * (1.0 / 6.0);
        w0 = s6 * fma(-h1, r0, 2.0);
        w2 = s6 * fma(-h0, r1, 2.0);
        w1 = (hs - w0) - w2;
      }
#pragma unroll
      for (int j = 0; j < NRAY; j++) {
        const double c = fma(w0, y2[j], fma(w1, y1[j], w2 * y[j]));
        if constexpr ((U & 1) != 0) P1[j] = fma(c, m[j], P1[j]);
        else P0[j] = fma(c, m[j], P0[j]);
        y2[j] = y1[j];
        y1[j] = y[j];
      }
      h0 = h1; r0 = r1;
    }
    x1 = tau;
    tm = k <= kcut ? fmax(tm, tau) : tm;
  };
  using std::integral_constant;
  using std::true_type;
  using std::false_type;
  auto chalf = [&](auto Fc, int blk) {
    const double *half = sT + (blk & 1) * (NB * 64);
    const int k0 = blk * NB;
    clayer(integral_constant<int, 0>{}, Fc, k0, half);
    clayer(integral_constant<int, 1>{}, Fc, k0, half);
    clayer(integral_constant<int, 2>{}, Fc, k0, half);
    clayer(integral_constant<int, 3>{}, Fc, k0, half);
    clayer(integral_constant<int, 4>{}, Fc, k0, half);
    clayer(integral_constant<int, 5>{}, Fc, k0, half);
  };
  int blk = 0;
  {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int stop = sFlag[0];
    chalf(true_type{}, 0);
    blk = 1;
    if (!__builtin_amdgcn_readfirstlane(stop)) {
      for (; blk < nblk;) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int stp = sFlag[blk & 1];
        chalf(false_type{}, blk);
        blk++;
        if (__builtin_amdgcn_readfirstlane(stp)) break;
      }
    }
  }
  if (!first) {
    log_event(count_alive(), blk * NB - 1);   // an event on the last layer walked
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the log's stores have reached L2
  }
  handoff();   // the log is complete, the producer's tau(kend) is in sEnd

  // ---- per ray of this wave: the sum of its parity, its padded panel, the deck's surface term
  double F = 0.0;
  {
    const double tauend = sEnd[lane];
    double Bend = 0.0;
    if (deck_on) Bend = bnum * rcp_n1(exp_rt(fmin(sC[kend * NC + 1] * nu, 700.0)) - 1.0);
    const bool anydied = __any(count_alive() < A);
    int kev[A];
#pragma unroll
    for (int sl = 0; sl < A; sl++) kev[sl] = -1;
    if (anydied) {
#pragma unroll
      for (int sl = 0; sl < A; sl++)
        kev[sl] = __builtin_amdgcn_raw_buffer_load_b32(rs_log, (int)(log_k0 + sl * nth * 4u), 0, 1 /* glc */);
    }
#pragma unroll
    for (int j = 0; j < NRAY; j++) {
      if (j >= nray) break;
      const int a = ra[j];
      const bool died = !(tm <= thr_r[j]);
      int kd = kend, slot = 0;
      const int rank = p.drank[a];
#pragma unroll
      for (int sl = 0; sl < A; sl++) {
        const bool take = sl <= rank && kev[sl] >= 0;
        kd = take ? kev[sl] : kd;
        slot = take ? sl : slot;
      }
      const bool odd_end = died ? ((kd + 1) & 1) != 0 : (kend & 1) != 0;
      double S = odd_end ? P1[j] : P0[j];
      if (__any(died)) {
        const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_log, (int)((unsigned)slot * nth * 16u + (unsigned)lane * 16u), 0, 1);
        const double taud = __builtin_bit_cast(double, (v2u_t){v.x, v.y});
        const double hd = __builtin_bit_cast(double, (v2u_t){v.z, v.w});
        const int kdc = died ? kd : 1, kp = kdc > 0 ? kdc - 1 : 0;
        const double mu = p.mu[a];
        double xe[4], ee[4];
        xe[0] = -fmin(taud, tcap) * invmu_r[j];
        xe[1] = -fmin(taud - hd, tcap) * invmu_r[j];
        xe[2] = fmin(sC[kdc * NC + 1] * nu, 700.0);
        xe[3] = fmin(sC[kp * NC + 1] * nu, 700.0);
        exp_rt_n<4>(xe, ee);
        const double yd = bnum * rcp_n1(ee[2] - 1.0) * ee[0];
        const double yp = bnum * rcp_n1(ee[3] - 1.0) * ee[1];
        const double rd = rcp_n1(hd), hs = hd + mu, s6 = hs * (1.0 / 6.0);
        const double wp = s6 * fma(-mu, rd, 2.0);
        const double wd = s6 * (hs * hs) * (rd * invmu_r[j]);
        const double pad = fma(wp, yp, wd * yd);
        S += died ? pad : 0.0;
      }
      F = fma(p.wq[a], S, F);
      if (deck_on) {
        const double xd = tauend * invmu_r[j];
        const bool deck = !died && !(xd > p.toomuch);
        const double Ed = exp_rt(fmax(-fmin(tauend, tcap) * invmu_r[j], kExpMin));
        F += deck ? p.wgt[a] * Bend * Ed : 0.0;
      }
    }
  }
  if (!first) sF[lane] = F;
  handoff();
  if (!first) return;
  F += sF[lane];
  if (__any(!(fabs(F) < __builtin_huge_val()))) {
    // a zero-width panel somewhere in this column (or an overflow): ray by ray, each with its own walk of the
    // layers (the rare path keeps to few registers: they decide how many teams a CU holds)
    const double nu4 = (nu * nu) * (nu * nu);
    const TableLoader<M, C> tab(p, ii, sI);
    F = 0.0;
    for (int a = 0; a < A; a++) {
      const double im = p.invmu[a];
      double se = 0.0, ep = 0.0, ep2 = 0.0, tau = 0.0;
      SlantRay<kIntegSimpson> ray;
      bool act = true;
      for (int k = 0; k <= kend; k++) {
        double r[NR];
        tab.load(k, r);
        const double *c = sC + k * NC;
        double e = fma(c[2 + 2 * M + 2 * C], nu4, c[3 + 2 * M + 2 * C]);
#pragma unroll
        for (int j = 0; j < NLD; j++) e = fma(c[2 + j], r[j], e);
        const double *wk = sW + 4 * k;
        double t = tau;
        if (k & 1) t = fma(ep + e, wk[3], se);
        else if (k >= 2) t = fma(wk[0], ep2, fma(wk[1], ep, fma(wk[2], e, se)));
        if (act) {
          tau = t;
          if (!(k & 1) && k >= 2) se = t;
        }
        ep2 = ep;
        ep = e;
        const double B = bnum * rcp_n1(exp_rt(fmin(c[1] * nu, 700.0)) - 1.0);
        const double x = tau * im;
        ray.point(act, x, B, exp_rt(fmax(-fmin(tau, tcap) * im, kExpMin)));
        act = act && !(x > p.toomuch);
        if (!__any(act)) break;
      }
      F += p.wgt[a] * ray.result(deck_on && act, L);
    }
  }
  if (valid) p.spec[(size_t)w * W + i] = F;
}

#ifndef __HIPCC_RTC__
bool launch_rt_slant_team(const RtArgs &b, bool sq, int nblocks, size_t sh, hipStream_t st, hipError_t &err);
#endif

}  // namespace bartrt
