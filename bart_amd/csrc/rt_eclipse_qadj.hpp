// rt_eclipse_qadj: the layer-parallel walk of rule 1 under `cut slant` with all rays per lane
// (rt_eclipse_quad<..., ALLR>, rt_eclipse.hpp) on a lane order that keeps a wavenumber's rows on ADJACENT lanes,
// so that everything that crosses the rows travels by DPP row shifts instead of ds_bpermute.
//
// rt_eclipse_quad puts row q of column m on lane q * WN + m: the R layers of a step read WN consecutive
// wavenumbers each, and a value from the row below comes through the LDS crossbar (ds_bpermute_b32, two per
// double, ~52 per step, a dozen of them in one dependent chain: the two prefix scans).  On the launches this
// kernel serves -- one to four walkers, a wave or two per SIMD -- nothing hides that latency.
//
// Here column m's rows are lanes m * R .. m * R + R - 1 (R = 8 or 16: a column lies inside one 16-lane DPP row),
// and the rows' ORDER alternates from step to step: on even steps row q sits on lane m R + q, on odd steps on lane
// m R + R - 1 - q.  The lane that holds the LAST row of a step therefore holds row 0 of the next one, and the lane
// of row R - 2 holds row 1: every carry of the walk (the extinction, optical depth and integrands of the two layers
// above the step, the running sums and maxima) stays in its lane -- no transport at all -- and "the row below" is
// one `row_shr:1` on even steps, one `row_shl:1` on odd ones.  Prefix sums and maxima run over row_shr / row_shl
// by 1, 2, 4 (8).  The arithmetic is rt_eclipse_quad<ALLR>'s (same panels, same weights, same alive / pad rules);
// sums that crossed the rows as "carry + prefix" there are taken as a prefix over (carry + first term) here, so
// spectra agree with it to rounding, not bit for bit.  Table reads: a lane's layer and wavenumber as before --
// the same addresses per wave, issued from other lanes.
#pragma once
#include "integ.hpp"
#include "kernels.hpp"
#include "prep.hpp"

namespace bartrt {

// a double moved across lanes by a DPP control word (lanes without a source read 0)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int kDppRowShl = 0x100, kDppRowShr = 0x110;   // + n (1 .. 15)

// value of the lane D rows below (row q - D); ODD: this step's rows run downwards in the lane order
template <int D, bool ODD>
__device__ __forceinline__ double rows_below(double x) {
  if constexpr (ODD) return dpp_move<kDppRowShl + D>(x);
  else return dpp_move<kDppRowShr + D>(x);
}

template <int R, bool ODD>
__device__ __forceinline__ double rows_prefix_sum(double v, int q) {
  { const double t = rows_below<1, ODD>(v); v += q >= 1 ? t : 0.0; }
  { const double t = rows_below<2, ODD>(v); v += q >= 2 ? t : 0.0; }
  { const double t = rows_below<4, ODD>(v); v += q >= 4 ? t : 0.0; }
  if constexpr (R > 8) { const double t = rows_below<8, ODD>(v); v += q >= 8 ? t : 0.0; }
  return v;
}
template <int R, bool ODD>
__device__ __forceinline__ double rows_prefix_max(double v, int q) {   // (values >= 0)
  { const double t = rows_below<1, ODD>(v); v = fmax(v, q >= 1 ? t : 0.0); }
  { const double t = rows_below<2, ODD>(v); v = fmax(v, q >= 2 ? t : 0.0); }
  { const double t = rows_below<4, ODD>(v); v = fmax(v, q >= 4 ? t : 0.0); }
  if constexpr (R > 8) { const double t = rows_below<8, ODD>(v); v = fmax(v, q >= 8 ? t : 0.0); }
  return v;
}

// (BARTRT_QADJ_WPE, A/B builds: held to three waves per SIMD -- 168 registers instead of 178-211 -- the kernel spills
// 3-60 registers and slows down: one walker at W = 1e4 34.8 -> 59.6 us, demo shape three walkers 23.2 -> 25.5; round 5)
#ifndef BARTRT_QADJ_WPE
#define BARTRT_QADJ_WPE 1
#endif
template <int AT, int MT, int CT, bool SQ, int R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BARTRT_QADJ_WPE))) void rt_eclipse_qadj(RtArgs p) {
  static_assert(R == 8 || R == 16, "a column's rows lie inside one 16-lane DPP row");
  extern __shared__ double smem[];
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 4 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C, NR = NLD > 0 ? NLD : 1;
  constexpr int AE = SQ ? A - 1 : A;     // transmittances that need an exponential
  constexpr int WN = 64 / R;             // wavenumbers per wave
  const int L = p.L, W = p.W;
  int bid = blockIdx.x;
  if (p.nprep > 0) {   // the head of the grid prepares the NEXT batch's layer records (RtArgs::nprep)
    if (bid < p.nprep) { prep_block(p.prep_next, bid, smem); return; }
    bid -= prep_slots(p.nprep);
    if (bid < 0) return;
  }
  // XCD-aware block -> (tile, walker) map (block_to_work, rt_eclipse.hpp)
  const int xcd = bid & 7, jb = bid >> 3;
  const int w = jb % p.nwalkers, tile = (jb / p.nwalkers) * 8 + xcd;
  if (tile >= p.ntiles) return;

  // (the rows of a wave read R different layer records at once: one double of padding per record spreads an even
  // record stride over all LDS banks)
  constexpr int NCS = (R >= 16 && NC % 2 == 0) ? NC + 1 : NC;
  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NCS);
  const double *sW = smem + (size_t)L * NCS + (size_t)L * NI;
  // nprep < 0: the workgroup prepares its walker's layer records itself (as rt_eclipse_quad does; launch_rt_spec, `fold`)
  __shared__ int sKstopFold;
  const bool fold = p.nprep < 0;
  if (fold) {
    PrepFold pf;
    pf.coef = sC; pf.stride = NCS; pf.idx = sI; pf.kstop = &sKstopFold; pf.global = tile == 0;
    prep_block(p.prep_next, w, smem + (size_t)L * NCS + (size_t)L * NI + 4 * (size_t)(L + kSimpsonPad), pf);
  } else if constexpr (NCS == NC) {
    stage2_to_lds(sC, p.coef + (size_t)w * L * NC, L * NC, sI, p.idx + (size_t)w * L * NI, L * NI, threadIdx.x, 256);
  } else {
    const double *src = p.coef + (size_t)w * L * NC;
    const idx_t *srci = p.idx + (size_t)w * L * NI;
    for (int k = threadIdx.x; k < L * NC; k += 256) sC[(k / NC) * NCS + k % NC] = src[k];
    for (int k = threadIdx.x; k < L * NI; k += 256) sI[k] = srci[k];
  }
  __syncthreads();
  simpson_radius_weights(const_cast<double *>(sW), sC, NCS, L, threadIdx.x, 256);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int m = lane / R, qe = lane % R, qo = R - 1 - qe;   // this lane's row on even / odd steps
  const int i0 = (tile * 4 + (threadIdx.x >> 6)) * WN;      // this wave's first wavenumber
  if (i0 >= W) return;
  const unsigned ii = (i0 + m < W) ? (unsigned)(i0 + m) : (unsigned)(W - 1);
  const double nu = p.wn[ii];
  const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
  const double nu4 = (nu * nu) * (nu * nu);
  const int kraw = fold ? sKstopFold : p.kstop[w], kend = kstop_layer(kraw);
  const bool deck_on = kstop_deck(kraw);
  const double tcap = tau_cap(p, A);

  const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.cia), 0, (int)p.cia_bytes, 0x00020000);
  const unsigned vk = ii * 8u * (unsigned)M, vc = ii * 16u, planeB = (unsigned)M * (unsigned)W * 8u;
  auto load_layer = [&](int j, double (&r)[NR]) {
    const idx_t *ix = sI + j * NI;
    if (M > 0) {
      const idx_t mine = ix[0];
      long long base = 0ll;
      if (p.window) {
        // the smallest plane offset among the step's layers (lanes 0 .. R - 1 hold all R of them)
        const int lo = (int)(unsigned)mine, hi = (int)(unsigned)((unsigned long long)mine >> 32);
#pragma unroll
        for (int r2 = 0; r2 < R; r2++) {
          const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, r2), h = (unsigned)__builtin_amdgcn_readlane(hi, r2);
          const long long v = (long long)(((unsigned long long)h << 32) | l);
          base = (r2 == 0 || v < base) ? v : base;
        }
      }
      const unsigned long long left = p.kappa_bytes - (unsigned long long)base;
      const auto rs_k = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.kappa) + base), 0,
          (int)(unsigned)(left < 0xffffffffull ? left : 0xffffffffull), 0x00020000);
      load_table_lane<M>(rs_k, (unsigned)(mine - base) + vk, 0, planeB, r);
    }
#pragma unroll
    for (int cc = 0; cc < C; cc++) load_cia_lane(rs_c, (unsigned)ix[1 + cc] + vc, 0, r + 2 * M + 2 * cc);
  };
  auto clampk = [&](int k) { return k < kend ? k : kend; };

  double Fs = 0.0;   // surface term of a cloud deck (one lane per wavenumber sets it)
  // The carries: values of the two layers above the step, held by the lane that is row 0 (the first of each
  // pair) / used by row 0 only.  Every lane keeps its own latest values; the lane that was the last row IS row 0.
  double c_e = 0.0, c2_e = 0.0, c_tau = 0.0, c2_tau = 0.0, c_S = 0.0, c_tm = 0.0, c_tm1 = 0.0, thr_max = p.thr[0];
  // a lane's layers j = R s + q alternate parity with the step (q = qe / R - 1 - qe, R even): the terms of its
  // even and of its odd steps are kept apart -- at the end the sum whose rows have the parity of the ray's last point counts
  double Ia[2][A], ca_y[A], ca2_y[A];
  int Npad[A];
  const int kcut = kend < L - 2 ? kend : L - 2;
#pragma unroll
  for (int a = 0; a < A; a++) { Ia[0][a] = Ia[1][a] = ca_y[a] = ca2_y[a] = 0.0; Npad[a] = -1; thr_max = p.thr[a] > thr_max ? p.thr[a] : thr_max; }
  bool active = i0 + m < W;   // some ray of the column was alive when the step before began its last row

  auto step = [&](int s, const double (&rv)[NR], auto odd_tag) {
    constexpr bool ODD = decltype(odd_tag)::value;
    const int q = ODD ? qo : qe;
    const int j = R * s + q, jc = clampk(j);
    const bool inrange = j <= kend;
    const double *c = sC + jc * NCS;
    double cf[NC];
#pragma unroll
    for (int x = 0; x < NC; x++) cf[x] = c[x];
    double e = fma(cf[2 + 2 * M + 2 * C], nu4, cf[3 + 2 * M + 2 * C]);   // Rayleigh + grey cloud
#pragma unroll
    for (int x = 0; x < NLD; x++) e = fma(cf[2 + x], rv[x], e);
    // extinction of the two layers above
    const double e_b = rows_below<1, ODD>(e);
    const double eprev = q == 0 ? c_e : e_b;
    const double ep_b = rows_below<1, ODD>(eprev);
    const double e2 = q == 0 ? c2_e : ep_b;
    c_e = e;
    c2_e = eprev;
    // optical depth: even rows contribute the radius panel that ends on them (row 0 carries the sum so far), odd
    // rows add the trapezoid of their last interval on top of the even sum below them
    const double *wS = sW + 3 * jc;
    double v = ((q & 1) == 0 && j >= 2 && inrange && active) ? fma(wS[0], e2, fma(wS[1], eprev, wS[2] * e)) : 0.0;
    v += q == 0 ? c_S : 0.0;
    const double S = rows_prefix_sum<R, ODD>(v, q);
    c_S = S;
    const double tau = (q & 1) ? fma((eprev + e) * cf[0], 0.5, S) : S;
    // the tau grid of this lane's panel (j - 2, j - 1, j)
    const double tau_b = rows_below<1, ODD>(tau);
    const double tau1 = q == 0 ? c_tau : tau_b;
    const double t1_b = rows_below<1, ODD>(tau1);
    const double tau2 = q == 0 ? c2_tau : t1_b;
    c_tau = tau;
    c2_tau = tau1;
    // largest optical depth of the layers above this one / above the one above (layers past kcut do not count:
    // a ray ends only where a deeper layer exists)
    double pm = j <= kcut ? tau : 0.0;
    pm = q == 0 ? fmax(pm, c_tm) : pm;
    const double incl = rows_prefix_max<R, ODD>(pm, q);
    const double incl_b = rows_below<1, ODD>(incl);
    const double excl = q == 0 ? c_tm : incl_b;            // over the layers < j
    const double excl_b = rows_below<1, ODD>(excl);
    const double excl1 = q == 0 ? c_tm1 : excl_b;          // over the layers < j - 1
    c_tm = incl;
    c_tm1 = excl;
    // Planck term and transmittances
    const double tcl = fmin(tau, tcap);
    double xs[AE + 1], ex[AE + 1], y[A];
    xs[AE] = fmin(cf[1] * nu, 700.0);
#pragma unroll
    for (int a = 0; a < AE; a++) xs[a] = -tcl * p.invmu[a];
    exp_rt_n<AE + 1>(xs, ex);
    const double B = bnum * rcp_n1(ex[AE] - 1.0);
#pragma unroll
    for (int a = 0; a < AE; a++) y[a] = B * ex[a];
    if (SQ) y[A - 1] = y[0] * ex[0];
    // Simpson weights of the panel (as rt_eclipse_quad<ALLR>)
    const double h0 = tau1 - tau2, h1 = tau - tau1, hs = h0 + h1;
    const bool deg0 = h0 == 0.0, deg = deg0 || h1 == 0.0;
    const double r0 = rcp_core(deg0 ? 1.0 : h0), r1 = rcp_core(h1 == 0.0 ? 1.0 : h1);
    const double s6 = hs * (1.0 / 6.0);
    double w0 = deg ? 0.5 * h0 : s6 * (2.0 - h1 * r0);
    double w1 = deg ? 0.5 * hs : s6 * (hs * hs * (r0 * r1));
    double w2 = deg ? 0.5 * h1 : s6 * (2.0 - h0 * r1);
    if (j == 1) { w0 = 0.0; w1 = 0.5 * (tau - tau1); w2 = w1; }   // the first interval: a trapezoid
#pragma unroll
    for (int a = 0; a < A; a++) {
      const double yb = rows_below<1, ODD>(y[a]);
      const double y1 = q == 0 ? ca_y[a] : yb;
      const double y1b = rows_below<1, ODD>(y1);
      const double y2 = q == 0 ? ca2_y[a] : y1b;
      ca_y[a] = y[a];
      ca2_y[a] = y1;
      const bool alive = inrange && excl <= p.thr[a];
      const bool pad = j >= 1 && j < L && !(excl <= p.thr[a]) && excl1 <= p.thr[a];   // the ray died on layer j - 1
      double cterm = fma(w0, y2, fma(w1, y1, w2 * y[a]));
      if (__any(pad)) {
        // one unit of slant depth (mu_a of vertical depth) past the last point, integrand 0 there
        const double hp = h0 + p.mu[a], p6 = hp * (1.0 / 6.0);
        const double v0 = deg0 ? 0.5 * h0 : p6 * (2.0 - p.mu[a] * r0);
        const double v1 = deg0 ? 0.5 * hp : p6 * (hp * hp * (r0 * p.invmu[a]));
        cterm = pad ? fma(v0, y2, v1 * y1) : cterm;
        Npad[a] = pad ? j : Npad[a];
      }
      Ia[ODD ? 1 : 0][a] += ((alive || pad) && j >= 1) ? cterm : 0.0;
      if (deck_on && j == kend && alive && !(tau > p.thr[a])) Fs = fma(p.wgt[a], y[a], Fs);   // deck reached below the ray's cut
    }
    // some ray was alive when the step's last row began (its pad or more layers may follow): the verdict of the lane
    // that holds the last row, read by the column's other lanes from a ballot
    const unsigned long long last_ok = __ballot(q == R - 1 && excl <= thr_max);
    const int holder = ODD ? (lane - qe) : (lane - qe + R - 1);   // the lane that holds row R - 1 on this step
    active = ((last_ok >> holder) & 1ull) != 0ull;
  };

  // rule 1 may need the row after the column's last layer for the padded point
  const int klast = kend + 1 < L ? kend + 1 : L - 1;
  // Loads in flight ahead of the step that uses them: the next step's.  (Two steps ahead -- a third register slot, 256
  // registers with a handful of spills at four molecules -- was measured in round 5: no gain on either grid, 35.2 -> 36.6 us
  // for one walker at W = 1e4.  Nor does the launch react to the max-ILP schedule or to a table that fits the Infinity
  // Cache: MEASUREMENTS_ARCHIVE.md, round 5.)
  double rbuf[2][NR];
  int slast = 0;   // the last step walked
  load_layer(clampk(qe), rbuf[0]);
  for (int s = 0; R * s <= klast; s += 2) {
    load_layer(clampk(R * (s + 1) + qo), rbuf[1]);
    step(s, rbuf[0], std::false_type{});
    slast = s;
    if (!__any(active) || R * (s + 1) > klast) break;
    load_layer(clampk(R * (s + 2) + qe), rbuf[0]);
    step(s + 1, rbuf[1], std::true_type{});
    slast = s + 1;
    if (!__any(active)) break;
  }
  // the rows of a wavenumber hold its layers' terms.  Each ray's last point is its padded row (known to the lane that
  // evaluated it) or the column's last layer; the terms of the rows of that parity are added up, one lane writes.
  double F = Fs;
#pragma unroll
  for (int a = 0; a < A; a++) {
    int np = Npad[a];
    for (int o = 1; o < R; o <<= 1) { const int t = __shfl_xor(np, o); np = t > np ? t : np; }
    const int nlast = np >= 0 ? np : kend;
    const double mine = ((qe & 1) == (nlast & 1)) ? Ia[0][a] : Ia[1][a];   // (qe on even steps, R - 1 - qe on odd ones)
    F = fma(p.wq[a], mine, F);
  }
  for (int o = 1; o < R; o <<= 1) F += __shfl_xor(F, o);
  if (qe == 0 && i0 + m < W) p.spec[(size_t)w * W + i0 + m] = F;
  if (p.walked_out && lane == 0) {
    const int layers = R * (slast + 1) < kend + 1 ? R * (slast + 1) : kend + 1;
    p.walked_out[(size_t)w * (4 * p.ntiles) + tile * 4 + (threadIdx.x >> 6)] = layers;
  }
}

}  // namespace bartrt
