// Integration rules of the eclipse geometry (DESIGN.md conventions C6 / C8; the
// parity tests hold each rule to its scalar CPU restatement).  The reference's engine
// is absent (empty submodule), so which rule it applies cannot be checked: the
// product carries the three candidates as a switch (cfg key `integ`, environment
// BARTRT_INTEG, bartrt_set_integ) and every eclipse kernel is built for each.
//
//   0  transmittance  tau by trapezoid over radius; I(mu) = int B d(exp(-tau/mu)),
//                     trapezoid in the transmittance (the default).
//   1  simpson        SURVEY.md App. A-4 as recalled: tau[k] by the Simpson /
//                     trapezoid hybrid over the layers k .. top (panels (0,1,2),
//                     (2,3,4), ... from the top, the trapezoid of (k-1, k) for odd
//                     k); I(mu) = (1/mu) int B exp(-tau/mu) dtau by the same hybrid
//                     over the points 0 .. last from the top (an even count starts
//                     with the trapezoid of (0, 1)), padded by one point of zero
//                     integrand one unit of tau past `last` when a layer exists
//                     there.
//   2  trapz_tau      tau as in 0; I(mu) = (1/mu) int B exp(-tau/mu) dtau by plain
//                     trapezoid over 0 .. last.
// A cloud deck reached below `toomuch` adds its surface term B exp(-tau/mu) in
// every rule (and ends rule 1's point list without the padded point).
#pragma once
#include "kernels.hpp"

namespace bartrt {

enum { kIntegTransmittance = 0, kIntegSimpson = 1, kIntegTrapzTau = 2, kIntegCount = 3 };

// Simpson weights of the panel (k-2, k-1, k) over radius, per layer, in LDS:
// sW[3 k + (0,1,2)] for even k >= 2 (zero elsewhere).  dr(k) = r_{k-1} - r_k is
// word 0 of the layer's coefficient record.  Called by every thread of the
// workgroup after the records are staged; the caller synchronises.  The table has
// kSimpsonPad extra (zero) layers: the unrolled layer blocks index a few layers
// past the column's end before masking them.
constexpr int kSimpsonPad = 8;
__host__ __device__ inline size_t simpson_lds_doubles(int L) { return 3 * (size_t)(L + kSimpsonPad); }
__device__ __forceinline__ void simpson_radius_weights(double *sW, const double *sC, int NC, int L, int tid,
                                                       int nthreads) {
  for (int k = tid; k < L + kSimpsonPad; k += nthreads) {
    double w0 = 0.0, w1 = 0.0, w2 = 0.0;
    if (k >= 2 && (k & 1) == 0 && k < L) {
      const double h0 = sC[(k - 1) * NC], h1 = sC[k * NC], hs = h0 + h1;
      if (h0 == 0.0 || h1 == 0.0) {
        w0 = 0.5 * h0; w1 = 0.5 * hs; w2 = 0.5 * h1;
      } else {
        w0 = hs / 6.0 * (2.0 - h1 / h0);
        w1 = hs / 6.0 * (hs * hs / (h0 * h1));
        w2 = hs / 6.0 * (2.0 - h0 / h1);
      }
    }
    sW[3 * k] = w0; sW[3 * k + 1] = w1; sW[3 * k + 2] = w2;
  }
}

// Optical depth of a column walked from the top, one layer per call (lane = one
// wavenumber).  `live`: the layer counts (the column has not passed toomuch and
// the layer is in range); lv = live ? 0.5 : 0.
template <int INTEG>
struct TauColumn {
  double tau = 0.0, eprev = 0.0;
  __device__ __forceinline__ void layer(int, bool, double lv, double e, double dr, const double *) {
    tau += (eprev + e) * dr * lv;
    eprev = e;
  }
};

template <>
struct TauColumn<kIntegSimpson> {
  double tau = 0.0, eprev = 0.0, e2 = 0.0, s_even = 0.0;
  // sW: the workgroup's Simpson weights (simpson_radius_weights); k is wave-uniform
  __device__ __forceinline__ void layer(int k, bool live, double lv, double e, double dr, const double *sW) {
    if (k & 1) {
      const double t = fma((eprev + e) * dr, lv, s_even);
      tau = live ? t : tau;
    } else if (k >= 2) {
      const double s = fma(sW[3 * k], e2, fma(sW[3 * k + 1], eprev, fma(sW[3 * k + 2], e, s_even)));
      s_even = live ? s : s_even;
      tau = live ? s : tau;
    }
    e2 = eprev;
    eprev = e;
  }
};

// Emergent intensity per ray angle of a column walked from the top.
//   layer(A, live, lv, tau, Bprev, B, E): after the layer's optical depth, Planck
//     term B (Bprev: the layer above) and transmittances E[a] = exp(-tau / mu_a);
//   flux(p, A, deck, Bprev, L, out): after the walk; deck = the column reached a
//     cloud deck below toomuch.  Returns sum_a wgt[a] I_a; out (optional): I_a.
template <int INTEG, int AMAX>
struct ColumnIntens;

template <int AMAX>
struct ColumnIntens<kIntegTransmittance, AMAX> {
  double I[AMAX], fprev[AMAX];
  __device__ __forceinline__ ColumnIntens() {
#pragma unroll
    for (int a = 0; a < AMAX; a++) { I[a] = 0.0; fprev[a] = 1.0; }
  }
  __device__ __forceinline__ void layer(int A, bool, double lv, double, double Bprev, double B,
                                        const double (&E)[AMAX]) {
    const double hb = (Bprev + B) * lv;
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      if (a >= A) break;
      I[a] = fma(hb, fprev[a] - E[a], I[a]);
      fprev[a] = E[a];
    }
  }
  __device__ __forceinline__ double flux(const RtArgs &p, int A, bool deck, double Bprev, int, double *out) {
    double F = 0.0;
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      if (a >= A) break;
      const double Ia = I[a] + (deck ? Bprev * fprev[a] : 0.0);
      F += p.wgt[a] * Ia;
      if (out) out[a] = Ia;
    }
    return F;
  }
};

template <int AMAX>
struct ColumnIntens<kIntegTrapzTau, AMAX> {
  double I[AMAX], y1[AMAX], x1 = 0.0;
  __device__ __forceinline__ ColumnIntens() {
#pragma unroll
    for (int a = 0; a < AMAX; a++) { I[a] = 0.0; y1[a] = 0.0; }
  }
  __device__ __forceinline__ void layer(int A, bool live, double lv, double tau, double, double B,
                                        const double (&E)[AMAX]) {
    const double h = (tau - x1) * lv;
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      if (a >= A) break;
      const double y = B * E[a];
      I[a] = fma(y1[a] + y, h, I[a]);
      y1[a] = live ? y : y1[a];
    }
    x1 = live ? tau : x1;
  }
  __device__ __forceinline__ double flux(const RtArgs &p, int A, bool deck, double, int, double *out) {
    double F = 0.0;
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      if (a >= A) break;
      const double Ia = fma(I[a], p.invmu[a], deck ? y1[a] : 0.0);
      F += p.wgt[a] * Ia;
      if (out) out[a] = Ia;
    }
    return F;
  }
};

// Flux-only accumulators of the kernels that do not return per-angle intensities
// (single-wave, producer/consumer, quad-layer).  Every rule is linear in the
// transmittances E_{a,k} = exp(-tau_k / mu_a) with weights that do not depend on the
// ray angle, so the angle quadrature is taken BEFORE the layer sum:
//   rule 0   F = sum_a w_a sum_k hb_k (E_{a,k-1} - E_{a,k}) = sum_k hb_k (G_{k-1} - G_k),
//            G_k = sum_a w_a E_{a,k};
//   rules 1 / 2   I_a = (1/mu_a) sum_k W_k(tau) B_k E_{a,k}  (W_k: the Simpson-hybrid or
//            trapezoid weights of the tau grid)  =>  F = sum_k W_k Y_k,
//            Y_k = B_k sum_a (w_a / mu_a) E_{a,k}   (RtArgs::wq = w_a / mu_a)
// -- one running sum and one previous value instead of A of each (16 VGPRs and a
// few operations less per layer; in the quad-layer kernel one value crosses the lane
// rows instead of A).
template <int INTEG, int AMAX>
struct ColumnFlux;

// G = sum_a w_a E_a
template <int AMAX>
__device__ __forceinline__ double angle_sum(const RtArgs &p, const double (&E)[AMAX]) {
  double g = p.wgt[0] * E[0];
#pragma unroll
  for (int a = 1; a < AMAX; a++) g = fma(p.wgt[a], E[a], g);
  return g;
}

template <int AMAX>
struct ColumnFlux<kIntegTransmittance, AMAX> {
  double F = 0.0, Gprev;
  __device__ __forceinline__ explicit ColumnFlux(const RtArgs &p) {
    Gprev = p.wgt[0];                       // transmittances are 1 above the top layer
#pragma unroll
    for (int a = 1; a < AMAX; a++) Gprev += p.wgt[a];
  }
  __device__ __forceinline__ void layer(const RtArgs &p, int, bool, double lv, double, double Bprev, double B,
                                        const double (&E)[AMAX]) {
    const double G = angle_sum<AMAX>(p, E);
    F = fma((Bprev + B) * lv, Gprev - G, F);
    Gprev = G;
  }
  __device__ __forceinline__ double flux(const RtArgs &, int, bool deck, double Bprev, int) {
    return deck ? fma(Bprev, Gprev, F) : F;
  }
};

// Simpson weights of the panel over (x - h0 - h1, x - h1, x); zero-width halves
// fall back to the two trapezoids.
__device__ __forceinline__ void simpson_tau_weights(double h0, double h1, double &w0, double &w1, double &w2) {
  const bool deg = h0 == 0.0 || h1 == 0.0;
  const double hs = h0 + h1;
  const double r0 = rcp_core(deg ? 1.0 : h0), r1 = rcp_core(deg ? 1.0 : h1);
  const double s6 = hs * (1.0 / 6.0);
  w0 = deg ? 0.5 * h0 : s6 * (2.0 - h1 * r0);
  w1 = deg ? 0.5 * hs : s6 * (hs * hs * (r0 * r1));
  w2 = deg ? 0.5 * h1 : s6 * (2.0 - h0 * r1);
}

template <int AMAX>
struct ColumnIntens<kIntegSimpson, AMAX> {
  // points so far (n), the last two abscissae and integrands, the running sums
  // of the panels that end on an even / odd point index (P0 / P1; P1 starts
  // with the trapezoid of the first interval)
  double P0[AMAX], P1[AMAX], y1[AMAX], y2[AMAX], x1 = 0.0, x2 = 0.0;
  int n = 0;
  __device__ __forceinline__ ColumnIntens() {
#pragma unroll
    for (int a = 0; a < AMAX; a++) { P0[a] = 0.0; P1[a] = 0.0; y1[a] = 0.0; y2[a] = 0.0; }
  }
  // appends the point (x, y[]) for the lanes with `live`
  __device__ __forceinline__ void point(int A, bool live, double x, const double (&y)[AMAX]) {
    const double h0 = x1 - x2, h1 = x - x1;
    double w0, w1, w2;
    simpson_tau_weights(h0, h1, w0, w1, w2);
    const bool second = n == 1, odd = (n & 1) != 0;
    const bool add0 = live && n >= 2 && !odd, add1 = live && n >= 1 && odd;
    // the second point closes the first interval: a trapezoid into P1
    w0 = second ? 0.0 : w0;
    w1 = second ? 0.5 * h1 : w1;
    w2 = second ? 0.5 * h1 : w2;
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      if (a >= A) break;
      const double c = fma(w0, y2[a], fma(w1, y1[a], w2 * y[a]));
      P0[a] += add0 ? c : 0.0;
      P1[a] += add1 ? c : 0.0;
      y2[a] = live ? y1[a] : y2[a];
      y1[a] = live ? y[a] : y1[a];
    }
    x2 = live ? x1 : x2;
    x1 = live ? x : x1;
    n += live ? 1 : 0;
  }
  __device__ __forceinline__ void layer(int A, bool live, double, double tau, double, double B,
                                        const double (&E)[AMAX]) {
    double y[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; a++) y[a] = a < A ? B * E[a] : 0.0;
    point(A, live, tau, y);
  }
  __device__ __forceinline__ double flux(const RtArgs &p, int A, bool deck, double, int L, double *out) {
    double ysurf[AMAX], zero[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; a++) { ysurf[a] = deck ? y1[a] : 0.0; zero[a] = 0.0; }
    // one padded point (integrand 0, one unit of tau further) while a layer exists below
    point(A, !deck && n < L, x1 + 1.0, zero);
    const bool odd_end = ((n - 1) & 1) != 0;   // parity of the last point's index
    double F = 0.0;
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      if (a >= A) break;
      const double Ia = fma(odd_end ? P1[a] : P0[a], p.invmu[a], ysurf[a]);
      F += p.wgt[a] * Ia;
      if (out) out[a] = Ia;
    }
    return F;
  }
};

// One ray angle's emergent intensity with the `toomuch` cut on ITS slant depth (cfg `cut slant`,
// DESIGN.md C19; generic kernel only): the angle has its own last layer and, under rule 1, its own
// padded point one unit of SLANT depth further.  Works in the slant depth x = tau / mu throughout,
// so the result is I_a itself.  point(live, x, B, E): layer by layer from the top.
template <int INTEG>
struct SlantRay {
  double I = 0.0, P1 = 0.0, y1 = 0.0, y2 = 0.0, x1 = 0.0, x2 = 0.0, Bprev = 0.0, Eprev = 1.0;
  int n = 0;
  __device__ __forceinline__ void simpson_point(bool live, double x, double y) {
    // x arrives as the product tau / mu of the caller: fused into these differences it would be taken
    // unrounded against the ROUNDED x1 of the layer above, and two layers of equal optical depth would
    // be a rounding error apart instead of a zero-width panel (1e15-fold garbage through 1 / h; the
    // zero-band case of tests/test_gpu_simpson.py caught it) -- no contraction here
#pragma clang fp contract(off)
    const double h0 = x1 - x2, h1 = x - x1;
    double w0, w1, w2;
    simpson_tau_weights(h0, h1, w0, w1, w2);
    const bool second = n == 1, odd = (n & 1) != 0;
    w0 = second ? 0.0 : w0;
    w1 = second ? 0.5 * h1 : w1;
    w2 = second ? 0.5 * h1 : w2;
    const double c = fma(w0, y2, fma(w1, y1, w2 * y));
    I += (live && n >= 2 && !odd) ? c : 0.0;      // panels that end on even points
    P1 += (live && n >= 1 && odd) ? c : 0.0;      // ... on odd points (starts with the first interval's trapezoid)
    y2 = live ? y1 : y2; y1 = live ? y : y1;
    x2 = live ? x1 : x2; x1 = live ? x : x1;
    n += live ? 1 : 0;
  }
  __device__ __forceinline__ void point(bool live, double x, double B, double E) {
    if (INTEG == kIntegTransmittance) {
      I += live ? 0.5 * (Bprev + B) * (Eprev - E) : 0.0;
    } else if (INTEG == kIntegTrapzTau) {
#pragma clang fp contract(off)
      I += live ? 0.5 * (y1 + B * E) * (x - x1) : 0.0;
      y1 = live ? B * E : y1; x1 = live ? x : x1;
    } else {
      simpson_point(live, x, B * E);
    }
    Bprev = live ? B : Bprev;
    Eprev = live ? E : Eprev;
  }
  // deck: the ray reached the cloud deck below toomuch (its surface term B E of the last layer)
  __device__ __forceinline__ double result(bool deck, int L) {
    double r = I;
    if (INTEG == kIntegSimpson) {
      simpson_point(!deck && n < L, x1 + 1.0, 0.0);
      r = ((n - 1) & 1) ? P1 : I;
    }
    return deck ? r + Bprev * Eprev : r;
  }
};

// sum_a (w_a / mu_a) E_a
template <int AMAX>
__device__ __forceinline__ double angle_sum_q(const RtArgs &p, const double (&E)[AMAX]) {
  double g = p.wq[0] * E[0];
#pragma unroll
  for (int a = 1; a < AMAX; a++) g = fma(p.wq[a], E[a], g);
  return g;
}

// Rule 2, angle quadrature first: plain trapezoid of Y in tau.
template <int AMAX>
struct ColumnFlux<kIntegTrapzTau, AMAX> {
  double F = 0.0, y1 = 0.0, x1 = 0.0, ydeck = 0.0;
  __device__ __forceinline__ explicit ColumnFlux(const RtArgs &) {}
  __device__ __forceinline__ void layer(const RtArgs &p, int, bool live, double lv, double tau, double, double B,
                                        const double (&E)[AMAX]) {
    const double y = B * angle_sum_q<AMAX>(p, E);
    F = fma(y1 + y, (tau - x1) * lv, F);
    y1 = live ? y : y1;
    x1 = live ? tau : x1;
    if (p.cloud_on) {   // wave-uniform: the surface term wants sum_a w_a B E_a of the last layer
      const double yd = B * angle_sum<AMAX>(p, E);
      ydeck = live ? yd : ydeck;
    }
  }
  __device__ __forceinline__ double flux(const RtArgs &, int, bool deck, double, int) {
    return deck ? F + ydeck : F;
  }
};

// `cut slant` in the single-wave kernel of rules 0 / 2 (rt_eclipse_fast<..., SLANT = true>): every ray ends on the
// first layer whose slant depth passes toomuch, i.e. ray a counts on layer k iff the largest optical depth of the
// layers above, tm, is at or below its threshold RtArgs::thr[a] (the largest tau with tau / mu_a <= toomuch).  The
// rays' terms are masked one by one (their weights selected against tm), so the quadrature is a masked dot product
// per layer instead of one value taken before the layer sum.  Neither rule pads a point, and a ray that passes the
// cut on the bottom layer ends there like one that does not -- except for a cloud deck's surface term, which a ray
// gets iff it is alive after the column's last layer.  (Rule 1: rt_eclipse_s1s.hpp.)
template <int INTEG, int AMAX>
struct ColumnFluxSlant;

template <int AMAX>
struct ColumnFluxSlant<kIntegTransmittance, AMAX> {
  double F = 0.0, tm = 0.0, Eprev[AMAX];
  __device__ __forceinline__ explicit ColumnFluxSlant(const RtArgs &) {
#pragma unroll
    for (int a = 0; a < AMAX; a++) Eprev[a] = 1.0;      // transmittances are 1 above the top layer
  }
  __device__ __forceinline__ void layer(const RtArgs &p, int, bool live, double lv, double tau, double Bprev, double B,
                                        const double (&E)[AMAX]) {
    double g = 0.0;
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      const double wm = tm <= p.thr[a] ? p.wgt[a] : 0.0;
      g = fma(wm, Eprev[a] - E[a], g);
      Eprev[a] = E[a];
    }
    F = fma((Bprev + B) * lv, g, F);
    tm = live ? fmax(tm, tau) : tm;
  }
  __device__ __forceinline__ double flux(const RtArgs &p, int, bool deck_on, double Bprev, int) {
    if (!deck_on) return F;
    double g = 0.0;
#pragma unroll
    for (int a = 0; a < AMAX; a++) g += tm <= p.thr[a] ? p.wgt[a] * Eprev[a] : 0.0;
    return fma(Bprev, g, F);
  }
};

template <int AMAX>
struct ColumnFluxSlant<kIntegTrapzTau, AMAX> {
  double F = 0.0, tm = 0.0, x1 = 0.0, Bl = 0.0, y1[AMAX], El[AMAX];
  __device__ __forceinline__ explicit ColumnFluxSlant(const RtArgs &) {
#pragma unroll
    for (int a = 0; a < AMAX; a++) { y1[a] = 0.0; El[a] = 1.0; }
  }
  __device__ __forceinline__ void layer(const RtArgs &p, int, bool live, double lv, double tau, double, double B,
                                        const double (&E)[AMAX]) {
    double g = 0.0;
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      const double wm = tm <= p.thr[a] ? p.wq[a] : 0.0;
      const double y = B * E[a];
      g = fma(wm, y1[a] + y, g);
      y1[a] = y;
      El[a] = live ? E[a] : El[a];
    }
    F = fma((tau - x1) * lv, g, F);
    x1 = live ? tau : x1;
    Bl = live ? B : Bl;
    tm = live ? fmax(tm, tau) : tm;
  }
  __device__ __forceinline__ double flux(const RtArgs &p, int, bool deck_on, double, int) {
    if (!deck_on) return F;
    double g = 0.0;
#pragma unroll
    for (int a = 0; a < AMAX; a++) g += tm <= p.thr[a] ? p.wgt[a] * El[a] : 0.0;
    return fma(Bl, g, F);
  }
};

// Rule 1, angle quadrature first, with the scalar restatement's case analysis (zero-width
// panels, padded point) spelled out per point: the accumulator of the producer / consumer
// kernel and of the careful path rt_eclipse_simpson falls back to.  The tuned walk of the
// single-wave kernel is in rt_eclipse_s1.hpp.
template <int AMAX>
struct ColumnFlux<kIntegSimpson, AMAX> {
  // points so far (n), the last two abscissae and integrands, the running sums of the
  // panels that end on an even / odd point index (P1 starts with the trapezoid of the
  // first interval)
  double P0 = 0.0, P1 = 0.0, y1 = 0.0, y2 = 0.0, x1 = 0.0, x2 = 0.0, ydeck = 0.0;
  int n = 0;
  __device__ __forceinline__ explicit ColumnFlux(const RtArgs &) {}
  __device__ __forceinline__ void point(bool live, double x, double y) {
    const double h0 = x1 - x2, h1 = x - x1;
    double w0, w1, w2;
    simpson_tau_weights(h0, h1, w0, w1, w2);
    const bool second = n == 1, odd = (n & 1) != 0;
    w0 = second ? 0.0 : w0;          // the second point closes the first interval: a trapezoid into P1
    w1 = second ? 0.5 * h1 : w1;
    w2 = second ? 0.5 * h1 : w2;
    const double c = fma(w0, y2, fma(w1, y1, w2 * y));
    P0 += (live && n >= 2 && !odd) ? c : 0.0;
    P1 += (live && n >= 1 && odd) ? c : 0.0;
    y2 = live ? y1 : y2;
    y1 = live ? y : y1;
    x2 = live ? x1 : x2;
    x1 = live ? x : x1;
    n += live ? 1 : 0;
  }
  __device__ __forceinline__ void layer(const RtArgs &p, int, bool live, double, double tau, double, double B,
                                        const double (&E)[AMAX]) {
    point(live, tau, B * angle_sum_q<AMAX>(p, E));
    if (p.cloud_on) {
      const double yd = B * angle_sum<AMAX>(p, E);
      ydeck = live ? yd : ydeck;
    }
  }
  __device__ __forceinline__ double flux(const RtArgs &, int, bool deck, double, int L) {
    // one padded point (integrand 0, one unit of tau further) while a layer exists below
    point(!deck && n < L, x1 + 1.0, 0.0);
    const double F = ((n - 1) & 1) ? P1 : P0;   // parity of the last point's index
    return deck ? F + ydeck : F;
  }
};

}  // namespace bartrt
