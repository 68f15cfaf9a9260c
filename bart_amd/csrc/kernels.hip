// CDNA4 (gfx950) kernels of the forward radiative-transfer path, eclipse geometry.
//
//  prep_profiles     one workgroup per walker: mean molecular mass, hydrostatic
//                    radii (reference law code/makeatm.py:183-263), densities and
//                    the per-layer interpolation weights -> coefficient records.
//  rt_eclipse        generic: one lane per (walker, wavenumber) streams the
//                    opacity grid layer by layer from the top (coalesced along
//                    the wavenumber axis, the grid's fastest index), accumulates
//                    the optical depth and the emergent intensity per ray angle,
//                    stops the wave once every lane passed `toomuch`.  Runtime
//                    angle / molecule / CIA counts; also carries the line-by-line
//                    extinction and the tau / per-angle outputs.
//  rt_eclipse_fast   the same walk specialised at compile time, buffer loads with
//                    scalar plane offsets, two pairs of register slots in flight.
//  rt_eclipse_split  5-8 walkers: producer / consumer wave pair per column.
//  rt_eclipse_quad   1-4 walkers: 16 wavenumbers x 4 layers per wave and step.
//
// The walker's coefficient records are staged in LDS once per workgroup and
// read back as wave-uniform broadcasts.  Interpolation + reduction, fp64 VALU;
// the transit geometry (transit_geom.hip) is where MFMA fits.
#include "kernels.hpp"
#include "prep.hpp"

#include <cmath>
#include <cstdlib>
#include <string>
#include <utility>

// upper bound on resident waves per SIMD the specialised kernels are compiled
// for: lets the compiler spend registers on loads in flight (measured best: 3-4)
#ifndef BARTRT_WPE
#define BARTRT_WPE 4
#endif

// Kernel choice by 64-wavenumber columns per launch (measured at W = 1e4, L = 100:
// 157 columns per walker; microseconds per launch, quad-layer / split / single-wave):
//   1 walker 26 / 39 / 54     2 walkers 33 / 38 / 48     4 walkers 45 / 47 / 50
//   5 walkers 54 / 52 / 52    6 walkers 58 / 53 / 55     8 walkers 73 / 68 / 75
//   9 walkers 81 / 80 / 80    10 walkers 90 / 85 / 84  (single-wave from here on)
constexpr long kQuadMaxColumns = 640;
constexpr long kOctoMaxColumns = 400;  // eight layers per step (R = 8) below this: 1 walker 23 us, 2 walkers 30 us
constexpr long kSplitMaxColumns = 1300;

namespace bartrt {

__global__ __launch_bounds__(128) void prep_profiles(PrepArgs p) {
  extern __shared__ double sm[];
  const int L = p.L, S = p.S, w = blockIdx.x;
  // Everything this kernel reads from HBM (the walker's profile and the block of
  // per-engine constants) is pulled into LDS in ONE batch of independent loads;
  // the phases of prep_body then run out of LDS.  The kernel is pure latency:
  // each dependent trip to memory it avoids is worth most of a microsecond.
  stage2_to_lds(prep_lds_profile(sm, L), p.prof + (size_t)w * (S + 1) * L, (S + 1) * L,
                prep_lds_consts(sm, L, S), p.consts, 2 * L + S + 2 * p.Nt + 2 * p.ncia_temps,
                threadIdx.x, blockDim.x);
  prep_body(p, w, sm, p.over ? p.over + (size_t)3 * w : nullptr);
}

// ---------------------------------------------------------------------------
// XCD-aware block -> (tile, walker) map.  Blocks b and b+8 share an XCD (and
// its L2), so all walkers of one wavenumber tile are placed on one XCD, walker
// index fastest: they stream the same grid rows at about the same time and
// the XCD's L2 serves the repeats.
__device__ inline void block_to_work(int b, int nwalkers, int &tile, int &walker) {
  const int xcd = b & 7, j = b >> 3;
  walker = j % nwalkers;
  tile = (j / nwalkers) * 8 + xcd;
}

template <int AT, int MT, int CT>
__global__ __launch_bounds__(256) void rt_eclipse(RtArgs p) {
  extern __shared__ double smem[];
  const int A = AT > 0 ? AT : p.A;
  const int M = MT >= 0 ? MT : p.M;
  const int C = CT >= 0 ? CT : p.C;
  const int L = p.L, W = p.W;
  const int NC = coef_stride(M, C), NI = idx_stride(C);
  int tile, w;
  block_to_work(blockIdx.x, p.nwalkers, tile, w);
  if (tile >= p.ntiles) return;

  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  {
    const double *gC = p.coef + (size_t)w * L * NC;
    const idx_t *gI = p.idx + (size_t)w * L * NI;
    stage2_to_lds(sC, gC, L * NC, sI, gI, L * NI, threadIdx.x, blockDim.x);
  }
  __syncthreads();

  const int i = tile * blockDim.x + threadIdx.x;
  const bool valid = i < W;
  const int ii = valid ? i : W - 1;  // keep every lane's loads in range
  const double nu = p.wn[ii];
  const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
  const double nu4 = (nu * nu) * (nu * nu);
  const size_t MW = (size_t)M * W;

  constexpr int AMAX = AT > 0 ? AT : kMaxAngles;
  double I[AMAX], fprev[AMAX];
#pragma unroll
  for (int a = 0; a < AMAX; a++) { I[a] = 0.0; fprev[a] = 1.0; }  // fprev: E_{a,k-1}

  double tau = 0.0, eprev = 0.0, Bprev = 0.0;
  bool active = true;
  int last = 0;
  const int kend = p.kstop[w];
  for (int k = 0; k <= kend; ++k) {
    const double *c = sC + k * NC;
    const idx_t *ix = sI + k * NI;
    const int l = L - 1 - k;
    double e = c[2 + 2 * M + 2 * C] * nu4;
    if (p.ext) e += p.ext[((size_t)w * L + l) * W + ii];
    const double *kb = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p.kappa) + ix[0]) + ii;
#pragma unroll
    for (int m = 0; m < (MT >= 0 ? MT : kMaxMol); m++) {
      if (MT < 0 && m >= M) break;
      e += c[2 + 2 * m] * kb[(size_t)m * W] + c[3 + 2 * m] * kb[MW + (size_t)m * W];
    }
#pragma unroll
    for (int cc = 0; cc < (CT >= 0 ? CT : kMaxCia); cc++) {
      if (CT < 0 && cc >= C) break;
      const double *ab = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p.cia) + ix[1 + cc]) + ii;
      e += c[2 + 2 * M + 2 * cc] * ab[0] + c[3 + 2 * M + 2 * cc] * ab[W];
    }
    const double dtau = active ? 0.5 * (eprev + e) * c[0] : 0.0;
    tau += dtau;
    // I_a += (B_{k-1} + B_k)/2 * (E_{a,k-1} - E_{a,k}), E = exp(-tau/mu):
    // trapezoid in the transmittance (exact for an isothermal column)
    const double B = bnum * rcp_core(exp_rt(fmin(c[1] * nu, 700.0)) - 1.0);
    const double hb = active ? 0.5 * (Bprev + B) : 0.0;
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      if (AT <= 0 && a >= A) break;
      const double E = exp_rt(fmax(-tau * p.invmu[a], kExpMin));
      I[a] += hb * (fprev[a] - E);
      fprev[a] = E;
    }
    Bprev = B;
    eprev = e;
    if (p.tau_out && valid) p.tau_out[(size_t)i * L + k] = tau;
    if (active) {
      last = k;
      if (tau > p.toomuch) active = false;
    }
    if (!__any(active)) break;
  }
  double F = 0.0;
  const bool surf = p.cloud_on && active;  // reached the deck below toomuch
#pragma unroll
  for (int a = 0; a < AMAX; a++) {
    if (AT <= 0 && a >= A) break;
    F += p.wgt[a] * (I[a] + (surf ? Bprev * fprev[a] : 0.0));
    if (p.intens_out && valid) p.intens_out[(size_t)a * W + i] = I[a] + (surf ? Bprev * fprev[a] : 0.0);
  }
  if (valid) {
    p.spec[(size_t)w * W + i] = F;
    if (p.tau_out) {
      for (int k = last + 1; k < L; k++) p.tau_out[(size_t)i * L + k] = tau;
      p.last_out[i] = last;
    }
  }
}

// Specialised kernel: compile-time angle / molecule / CIA counts, scalar row
// bases (SGPR) + one 32-bit lane offset for every load, and two pairs of
// register slots of 2M+2C loads kept in flight ahead of the arithmetic, so that
// the one or two waves a SIMD holds at small batch sizes cover the HBM latency
// by themselves.
// SQ: the last ray angle has exactly half the cosine of the first (launch_rt
// orders them so; 0 and 60 degrees of the usual raygrid), so its transmittance
// is the first one's square: exp(-2 tau / mu) = exp(-tau / mu)^2 -- one
// multiplication instead of one of the six exponentials of a layer.
template <int AT, int MT, int CT, bool SQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, BARTRT_WPE)))
void rt_eclipse_fast(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 3 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C;
  constexpr int NR = NLD > 0 ? NLD : 1;
  const int L = p.L, W = p.W;
  int tile, w;
  block_to_work(blockIdx.x, p.nwalkers, tile, w);
  if (tile >= p.ntiles) return;

  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  {
    const double *gC = p.coef + (size_t)w * L * NC;
    const idx_t *gI = p.idx + (size_t)w * L * NI;
    stage2_to_lds(sC, gC, L * NC, sI, gI, L * NI, threadIdx.x, blockDim.x);
  }
  __syncthreads();

  const int i = tile * blockDim.x + threadIdx.x;
  const bool valid = i < W;
  const unsigned ii = valid ? (unsigned)i : (unsigned)(W - 1);
  const double nu = p.wn[ii];
  const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
  const double nu4 = (nu * nu) * (nu * nu);
  const TableLoader<M, C> tab(p, ii, sI);
  auto load_layer = [&](int k, double (&r)[NR]) { tab.load(k, r); };

  double I[A], fprev[A];
#pragma unroll
  for (int a = 0; a < A; a++) { I[a] = 0.0; fprev[a] = 1.0; }  // fprev: E_{a,k-1}
  double tau = 0.0, eprev = 0.0, Bprev = 0.0;
  bool active = true;
  const int kend = p.kstop[w];
  const double tcap = tau_cap(p, A);

  // One layer's arithmetic.  Straight-line: layer indices past the end are
  // clamped and masked instead of branched around, so that inside an unrolled
  // block the compiler waits (counted vmcnt) on exactly the loads a layer
  // needs and leaves the younger ones in flight.
  // The record of a layer is read from LDS one layer ahead (cf / cfn alternate
  // between two register sets), so its latency is not waited out at the head of
  // the layer that uses it; masking by multiplication keeps the reads of the
  // path length and of c2/T out of conditional blocks.
  auto read_rec = [&](int k, double (&cf)[NC]) {
    const double *c = sC + (k < kend ? k : kend) * NC;
#pragma unroll
    for (int j = 0; j < NC; j++) cf[j] = c[j];
  };
  auto layer = [&](int k, const double (&r)[NR], const double (&cf)[NC], double (&cfn)[NC]) {
    const bool live = active && k <= kend;
    read_rec(k + 1, cfn);
    const double lv = live ? 0.5 : 0.0;
    double e = cf[2 + 2 * M + 2 * C] * nu4;
#pragma unroll
    for (int j = 0; j < NLD; j++) e = fma(cf[2 + j], r[j], e);
    const double dtau = (eprev + e) * cf[0] * lv;
    tau += dtau;
    // Planck exponent and the A slant-path exponents in one interleaved batch
    const double tc = fmin(tau, tcap);
    constexpr int AE = SQ ? A - 1 : A;  // transmittances that need an exponential
    double xs[AE + 1], ex[AE + 1], es[A];
    xs[AE] = fmin(cf[1] * nu, 700.0);
#pragma unroll
    for (int a = 0; a < AE; a++) xs[a] = -tc * p.invmu[a];
    exp_rt_n<AE + 1>(xs, ex);
#pragma unroll
    for (int a = 0; a < AE; a++) es[a] = ex[a];
    if (SQ) es[A - 1] = ex[0] * ex[0];
    const double B = bnum * rcp_core(ex[AE] - 1.0);
    const double hb = (Bprev + B) * lv;
#pragma unroll
    for (int a = 0; a < A; a++) {
      I[a] = fma(hb, fprev[a] - es[a], I[a]);
      fprev[a] = es[a];
    }
    Bprev = B;
    eprev = e;
    active = active && !(live && tau > p.toomuch);
  };
  auto clampk = [&](int k) { return k < kend ? k : kend; };

  // two pairs of slots; each pair is reloaded two layers before it is used and
  // the loads that cross the loop's back edge were issued two layers earlier
  double a0[NR], a1[NR], b0[NR], b1[NR];
  load_layer(clampk(0), a0);
  load_layer(clampk(1), a1);
  double cfE[NC], cfO[NC];
  read_rec(0, cfE);
  for (int k0 = 0; k0 <= kend; k0 += 4) {
    load_layer(clampk(k0 + 2), b0);
    load_layer(clampk(k0 + 3), b1);
    layer(k0, a0, cfE, cfO);
    layer(k0 + 1, a1, cfO, cfE);
    load_layer(clampk(k0 + 4), a0);
    load_layer(clampk(k0 + 5), a1);
    layer(k0 + 2, b0, cfE, cfO);
    layer(k0 + 3, b1, cfO, cfE);
    if (!__any(active)) break;
  }
  double F = 0.0;
  const bool surf = p.cloud_on && active;
#pragma unroll
  for (int a = 0; a < A; a++) F += p.wgt[a] * (I[a] + (surf ? Bprev * fprev[a] : 0.0));
  if (valid) p.spec[(size_t)w * W + i] = F;
}

// Few-walker variant (5-8 walkers at W = 1e4; below that the quad-layer
// kernel is faster still): the layer loop is split over TWO waves per 64
// wavenumbers.  Wave 0 (producer) streams the tables and advances the optical
// depth and the Planck term; wave 1 (consumer) turns each tau into the A
// transmittances and accumulates the intensities.  The halves are about equal
// in issue slots, so the serial time per layer halves at unchanged total work
// -- it pays while the single-wave columns cannot load the 1 024 SIMDs evenly
// (8 walkers: 68 vs 75 us; from 9 walkers on the single-wave kernel is as fast).
// Hand-off: an LDS ring of two 4-layer halves [tau, (B_{k-1}+B_k)/2 * live]
// per lane and ONE raw workgroup barrier per 4 layers (the consumer reads half
// b while the producer fills half b+1).
template <int AT, int MT, int CT, bool SQ>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, BARTRT_WPE)))
void rt_eclipse_split(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 3 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C;
  constexpr int NR = NLD > 0 ? NLD : 1;
  const int L = p.L, W = p.W;
  int tile, w;
  block_to_work(blockIdx.x, p.nwalkers, tile, w);
  if (tile >= p.ntiles) return;

  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  double *sX = smem + (size_t)L * NC + (size_t)L * NI;  // [2 halves][4 layers][tau, hb][64]
  int *sFlag = reinterpret_cast<int *>(sX + 1024);      // [half] producer saw every lane finished
  double *sEnd = sX + 1024 + 2;                         // [64] B of the last layer (cloud deck term)
  {
    const double *gC = p.coef + (size_t)w * L * NC;
    const idx_t *gI = p.idx + (size_t)w * L * NI;
    stage2_to_lds(sC, gC, L * NC, sI, gI, L * NI, threadIdx.x, 128);
    if (threadIdx.x < 2) sFlag[threadIdx.x] = 0;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform
  const int i = tile * 64 + lane;
  const bool valid = i < W;
  const unsigned ii = valid ? (unsigned)i : (unsigned)(W - 1);
  const int kend = p.kstop[w];
  const int nblk = kend / 4 + 1;  // 4-layer blocks; both waves run the same count

  if (role == 0) {
    // ---------------- producer: extinction, tau, Planck ----------------
    const double nu = p.wn[ii];
    const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
    const double nu4 = (nu * nu) * (nu * nu);
    const TableLoader<M, C> tab(p, ii, sI);
    auto load_layer = [&](int k, double (&r)[NR]) { tab.load(k, r); };
    double tau = 0.0, eprev = 0.0, Bprev = 0.0;
    bool active = true;
    auto layer = [&](int k, const double (&r)[NR]) {
      const int kc = k < kend ? k : kend;
      const bool live = active && k <= kend;
      const double *c = sC + kc * NC;
      double cf[NC];
#pragma unroll
      for (int j = 0; j < NC; j++) cf[j] = c[j];
      const double lv = live ? 0.5 : 0.0;
      double e = cf[2 + 2 * M + 2 * C] * nu4;
#pragma unroll
      for (int j = 0; j < NLD; j++) e = fma(cf[2 + j], r[j], e);
      tau += (eprev + e) * cf[0] * lv;
      const double B = bnum * rcp_core(exp_rt(fmin(cf[1] * nu, 700.0)) - 1.0);
      double *slot = sX + (k & 7) * 128;   // half (k/4)&1, layer k&3
      slot[lane] = tau;
      slot[64 + lane] = (Bprev + B) * lv;
      Bprev = B;
      eprev = e;
      active = active && !(live && tau > p.toomuch);
    };
    // LDS writes of a 4-layer block complete, then meet the consumer (raw
    // barrier: a __syncthreads() fence would also drain the table loads in flight)
    auto handoff = [&]() {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    auto clampk = [&](int k) { return k < kend ? k : kend; };
    double a0[NR], a1[NR], b0[NR], b1[NR];
    load_layer(clampk(0), a0);
    load_layer(clampk(1), a1);
    for (int blk = 0; blk < nblk; blk++) {
      const int k0 = blk * 4;
      load_layer(clampk(k0 + 2), b0);
      load_layer(clampk(k0 + 3), b1);
      layer(k0, a0);
      layer(k0 + 1, a1);
      load_layer(clampk(k0 + 4), a0);
      load_layer(clampk(k0 + 5), a1);
      layer(k0 + 2, b0);
      layer(k0 + 3, b1);
      // the exit decision travels with the block, so both waves leave after
      // the same barrier
      const bool stop = !__any(active);
      if (lane == 0) sFlag[blk & 1] = stop ? 1 : 0;
      handoff();
      if (stop) break;
    }
    sEnd[lane] = (p.cloud_on && active) ? Bprev : 0.0;
    handoff();
  } else {
    // ---------------- consumer: transmittances and intensities ----------------
    double I[A], fprev[A];
#pragma unroll
    for (int a = 0; a < A; a++) { I[a] = 0.0; fprev[a] = 1.0; }
    const double tcap = tau_cap(p, A);
    for (int blk = 0; blk < nblk; blk++) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int stop = sFlag[blk & 1];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double *slot = sX + ((blk & 1) * 4 + u) * 128;
        const double tc = fmin(slot[lane], tcap), hb = slot[64 + lane];
        constexpr int AE = SQ ? A - 1 : A;
        double xs[AE], es[A];
#pragma unroll
        for (int a = 0; a < AE; a++) xs[a] = -tc * p.invmu[a];
        {
          double ex[AE];
          exp_rt_n<AE>(xs, ex);
#pragma unroll
          for (int a = 0; a < AE; a++) es[a] = ex[a];
          if (SQ) es[A - 1] = ex[0] * ex[0];
        }
#pragma unroll
        for (int a = 0; a < A; a++) {
          I[a] = fma(hb, fprev[a] - es[a], I[a]);
          fprev[a] = es[a];
        }
      }
      if (__builtin_amdgcn_readfirstlane(stop)) break;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const double bsurf = sEnd[lane];
    double F = 0.0;
#pragma unroll
    for (int a = 0; a < A; a++) F += p.wgt[a] * (I[a] + bsurf * fprev[a]);
    if (valid) p.spec[(size_t)w * W + i] = F;
  }
}

// ---------------------------------------------------------------------------
// Quad-layer variant: a wave takes 16 wavenumbers and FOUR layers at a time.
// Lane (q = l / 16, m = l % 16) owns wavenumber m in the layers j = 4 s + q: per
// step s the four lane rows load and evaluate four consecutive layers side by
// side, the optical depth is a 4-lane prefix sum on top of the running value of
// the previous step (the wavefront scan of the tau integral), and every lane
// turns its own tau into its layer's Planck term and A transmittances.  The
// previous layer's values a trapezoid step needs come from the lane row below
// (row 3 of the previous step for row 0).  Same arithmetic per (layer,
// wavenumber) as the single-wave kernel, but a column is 25 steps deep instead
// of 100 layers, and a launch is made of four times as many, four times shorter
// waves: ten walkers are 6 250 of them over 1 024 SIMDs instead of 1 570 that
// leave half of the SIMDs with two and half with one.  The `toomuch` exit is
// per 16 wavenumbers and per step of four layers.
// R = lane rows = layers per step (4, or 8 for the smallest launches: 8
// wavenumbers x 8 layers per wave, twice the waves, half the depth).
template <int AT, int MT, int CT, bool SQ, int R>
__global__ __launch_bounds__(256) void rt_eclipse_quad(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 3 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C, NR = NLD > 0 ? NLD : 1;
  constexpr int AE = SQ ? A - 1 : A;  // transmittances that need an exponential
  constexpr int WN = 64 / R;           // wavenumbers per wave
  const int L = p.L, W = p.W;
  int tile, w;
  block_to_work(blockIdx.x, p.nwalkers, tile, w);
  if (tile >= p.ntiles) return;

  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  stage2_to_lds(sC, p.coef + (size_t)w * L * NC, L * NC, sI, p.idx + (size_t)w * L * NI, L * NI,
                threadIdx.x, 256);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int q = lane / WN, m = lane % WN;
  const int i0 = (tile * 4 + (threadIdx.x >> 6)) * WN;  // this wave's first wavenumber
  if (i0 >= W) return;
  const unsigned ii = i0 + m < W ? (unsigned)(i0 + m) : (unsigned)(W - 1);
  const double nu = p.wn[ii];
  const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
  const double nu4 = (nu * nu) * (nu * nu);
  const int kend = p.kstop[w];
  const double tcap = tau_cap(p, A);

  // per-lane table addressing: plane offset of the lane's layer + row + lane
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.cia), 0, (int)p.cia_bytes, 0x00020000);
  const unsigned off = ii * 8u, rowB = (unsigned)W * 8u, planeB = (unsigned)M * rowB;
  auto load_layer = [&](int j, double (&r)[NR]) {
    const idx_t *ix = sI + j * NI;
    if (M > 0) {
      const idx_t mine = ix[0];
      const long long base = p.window ? row_window_base<R>(mine) : 0ll;
      const unsigned long long left = p.kappa_bytes - (unsigned long long)base;
      const auto rs_k = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.kappa) + base), 0,
          (int)(unsigned)(left < 0xffffffffull ? left : 0xffffffffull), 0x00020000);
      const unsigned po = (unsigned)(mine - base) + off;
#pragma unroll
      for (int mm = 0; mm < M; mm++) {
        r[2 * mm] = __builtin_bit_cast(double, (v2u)__builtin_amdgcn_raw_buffer_load_b64(rs_k, (int)(po + mm * rowB), 0, 0));
        r[2 * mm + 1] = __builtin_bit_cast(double, (v2u)__builtin_amdgcn_raw_buffer_load_b64(rs_k, (int)(po + planeB + mm * rowB), 0, 0));
      }
    }
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
      const unsigned po = (unsigned)ix[1 + cc] + off;
      r[2 * M + 2 * cc] = __builtin_bit_cast(double, (v2u)__builtin_amdgcn_raw_buffer_load_b64(rs_c, (int)po, 0, 0));
      r[2 * M + 2 * cc + 1] = __builtin_bit_cast(double, (v2u)__builtin_amdgcn_raw_buffer_load_b64(rs_c, (int)(po + rowB), 0, 0));
    }
  };
  auto clampk = [&](int k) { return k < kend ? k : kend; };

  // the lanes of one wavenumber: bits m, WN + m, 2 WN + m, ... of a ballot
  const unsigned long long col_bits = (R == 4 ? 0x0001000100010001ull : 0x0101010101010101ull) << m;
  const unsigned long long below_bits = col_bits & ((1ull << (WN * q)) - 1ull);
  const int from_below = (lane + 64 - WN) & 63;  // row q - 1 (the last row for row 0)

  double I[A];
#pragma unroll
  for (int a = 0; a < A; a++) I[a] = 0.0;
  // carries of row 0: extinction, Planck term, transmittances of the layer just
  // above this step (row 3 of the previous step), and the optical depth there
  double c_e = 0.0, c_B = 0.0, c_E[AE], c_tau = 0.0;
#pragma unroll
  for (int a = 0; a < AE; a++) c_E[a] = 1.0;
  bool active = true;  // no layer above this step passed `toomuch` (per wavenumber, all rows agree)

  auto step = [&](int s, const double (&rv)[NR]) {
    const int j = R * s + q, jc = clampk(j);
    const bool inrange = j <= kend;
    const double *c = sC + jc * NC;
    double cf[NC];
#pragma unroll
    for (int x = 0; x < NC; x++) cf[x] = c[x];
    double e = cf[2 + 2 * M + 2 * C] * nu4;
#pragma unroll
    for (int x = 0; x < NLD; x++) e = fma(cf[2 + x], rv[x], e);
    // extinction of the layer above
    const double e_below = __shfl(e, from_below);
    const double eprev = q == 0 ? c_e : e_below;
    c_e = e_below;
    // optical depth: R-lane prefix sum of the step's increments + the running value
    double v = (eprev + e) * cf[0] * ((inrange && active) ? 0.5 : 0.0);
#pragma unroll
    for (int d = 1; d < R; d <<= 1) {
      const double t = __shfl(v, (lane + 64 - d * WN) & 63);
      if (q >= d) v += t;
    }
    const double tau = c_tau + v;
    c_tau = __shfl(tau, (R - 1) * WN + m);
    // which layers of this step are still above the cut
    const unsigned long long over = __ballot(inrange && active && tau > p.toomuch);
    const bool live = inrange && active && (over & below_bits) == 0ull;
    active = active && (over & col_bits) == 0ull;
    // Planck term and transmittances of this lane's layer
    const double tc = fmin(tau, tcap);
    double xs[AE + 1], ex[AE + 1];
    xs[AE] = fmin(cf[1] * nu, 700.0);
#pragma unroll
    for (int a = 0; a < AE; a++) xs[a] = -tc * p.invmu[a];
    exp_rt_n<AE + 1>(xs, ex);
    const double B = bnum * rcp_core(ex[AE] - 1.0);
    // the layer above: row q - 1, or the carry for row 0
    const double B_below = __shfl(B, from_below);
    const double Bprev = q == 0 ? c_B : B_below;
    c_B = B_below;
    const double hb = (Bprev + B) * (live ? 0.5 : 0.0);
    double Eprev[A], E[A];
#pragma unroll
    for (int a = 0; a < AE; a++) {
      const double E_below = __shfl(ex[a], from_below);
      Eprev[a] = q == 0 ? c_E[a] : E_below;
      c_E[a] = E_below;
      E[a] = ex[a];
    }
    if (SQ) {
      Eprev[A - 1] = Eprev[0] * Eprev[0];
      E[A - 1] = E[0] * E[0];
    }
#pragma unroll
    for (int a = 0; a < A; a++) I[a] = fma(hb, Eprev[a] - E[a], I[a]);
    if (p.cloud_on && j == kend && live && !(tau > p.toomuch)) {  // deck reached below toomuch
#pragma unroll
      for (int a = 0; a < A; a++) I[a] = fma(B, E[a], I[a]);
    }
  };

  double ra[NR], rb[NR];
  load_layer(clampk(q), ra);
  for (int s = 0; R * s <= kend; s += 2) {
    load_layer(clampk(R * (s + 1) + q), rb);
    step(s, ra);
    if (!__any(active)) break;
    load_layer(clampk(R * (s + 2) + q), ra);
    if (R * (s + 1) <= kend) {
      step(s + 1, rb);
      if (!__any(active)) break;
    }
  }
  // the rows of a wavenumber hold its layers' terms: sum them; row 0 writes
  double F = 0.0;
#pragma unroll
  for (int a = 0; a < A; a++) {
    double t = I[a];
    for (int o = WN; o < 64; o <<= 1) t += __shfl_xor(t, o);
    F = fma(p.wgt[a], t, F);
  }
  if (q == 0 && i0 + m < W) p.spec[(size_t)w * W + i0 + m] = F;
}

// ---------------------------------------------------------------------------
hipError_t launch_prep(const PrepArgs &a, hipStream_t st) {
  if (a.nwalkers <= 0) return hipSuccess;
  const size_t sh = sizeof(double) * prep_lds_doubles(a.L, a.S, a.Nt, a.ncia_temps);
  hipLaunchKernelGGL(prep_profiles, dim3(a.nwalkers), dim3(128), sh, st, a);
  return hipGetLastError();
}

template <int AT, int MT, int CT>
static hipError_t launch_rt_t(const RtArgs &a, int block, int nblocks, size_t sh, hipStream_t st) {
  // layer records above the 64 kB default (deep columns, many molecules): opt in once
  static size_t allowed = 64 * 1024;
  if (sh > allowed) {
    if (sh > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rt_eclipse<AT, MT, CT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return e;
    allowed = sh;
  }
  hipLaunchKernelGGL((rt_eclipse<AT, MT, CT>), dim3(nblocks), dim3(block), sh, st, a);
  return hipGetLastError();
}

// If one ray angle has exactly half the cosine of another (0 and 60 degrees of
// the usual raygrid 0 20 40 60 80), put that pair first and last: the SQ kernels
// take the last transmittance as the square of the first.
static bool order_angles_for_square(RtArgs &r) {
  for (int i = 0; i < r.A; i++)
    for (int j = 0; j < r.A; j++) {
      if (i == j || std::fabs(r.invmu[j] - 2.0 * r.invmu[i]) > 8.9e-16 * r.invmu[j]) continue;
      auto swap_angles = [&](int x, int y) {
        std::swap(r.invmu[x], r.invmu[y]);
        std::swap(r.wgt[x], r.wgt[y]);
      };
      swap_angles(0, i);
      if (j == 0) j = i;  // the doubled angle sat in slot 0 and moved to i
      swap_angles(r.A - 1, j);
      return true;
    }
  return false;
}

// block: threads per workgroup (64 or 256); a.ntiles must be ceil(W/block).
hipError_t launch_rt(const RtArgs &a, int block, hipStream_t st) {
  if (a.nwalkers <= 0 || a.W <= 0) return hipSuccess;
  const int ntiles8 = (a.ntiles + 7) / 8 * 8;
  const int nblocks = ntiles8 * a.nwalkers;
  const size_t sh = sizeof(double) * (size_t)a.L * coef_stride(a.M, a.C) +
                    sizeof(idx_t) * (size_t)a.L * idx_stride(a.C);
  static const std::string kmode = [] {
    const char *e = std::getenv("BARTRT_KERNEL");  // generic | mono | split | lp (A/B runs)
    return std::string(e ? e : "");
  }();
  static const bool force_window = std::getenv("BARTRT_WINDOW") != nullptr;  // windowed addressing on any grid (tests)
  static const bool allow_sq = [] {
    const char *e = std::getenv("BARTRT_SQ");  // 0: always evaluate every transmittance (A/B runs)
    return !(e && e[0] == '0');
  }();
  // (the specialised kernels rebuild their buffer descriptor per layer, so the
  // table may be of any size; one layer's pair of planes must stay below 4 GB)
  const bool plane_ok = 2ull * a.M * a.W * 8ull < (1ull << 31);
  if (kmode != "generic" && a.A == 5 && !a.ext && !a.intens_out && !a.tau_out && plane_ok &&
      sh <= 55 * 1024) {  // (the producer/consumer kernel adds 9 kB of its own)
    RtArgs b = a;
    const bool sq = allow_sq && order_angles_for_square(b);
    // too few single-wave columns to load the 1 024 SIMDs evenly -> several
    // waves per 64 wavenumbers: four 16-wavenumber waves that take four layers at
    // a time (quad-layer), or a producer / consumer pair
    const long columns = (long)a.nwalkers * ((a.W + 63) / 64);
    const int ntiles64 = (a.W + 63) / 64;
    const int nb64 = (ntiles64 + 7) / 8 * 8 * a.nwalkers;
    // the quad-layer kernel addresses the tables with per-lane 32-bit offsets
    // (a grid of 4 GB or more through a window that moves with the step's layers)
    const bool octo = kmode == "octo" || (kmode.empty() && columns <= kOctoMaxColumns);
    b.window = a.kappa_bytes >= (1ull << 32) - 4096 || force_window;
    const bool fits32 = a.cia_bytes < (1ull << 32) - 4096 && (!b.window || window_fits(a, octo ? 8 : 4));
    if ((kmode == "quad" || kmode == "octo" || (kmode.empty() && columns <= kQuadMaxColumns)) && fits32) {
      // the smallest launches take eight layers per step (8 wavenumbers per wave)
      b.ntiles = octo ? (a.W + 31) / 32 : ntiles64;
      const int nbq = (b.ntiles + 7) / 8 * 8 * a.nwalkers;
#define BARTRT_QUAD(MM, CC)                                                                                  \
  if (a.M == MM && a.C == CC) {                                                                              \
    if (octo) {                                                                                              \
      if (sq) hipLaunchKernelGGL((rt_eclipse_quad<5, MM, CC, true, 8>), dim3(nbq), dim3(256), sh, st, b);    \
      else hipLaunchKernelGGL((rt_eclipse_quad<5, MM, CC, false, 8>), dim3(nbq), dim3(256), sh, st, b);      \
    } else {                                                                                                 \
      if (sq) hipLaunchKernelGGL((rt_eclipse_quad<5, MM, CC, true, 4>), dim3(nbq), dim3(256), sh, st, b);    \
      else hipLaunchKernelGGL((rt_eclipse_quad<5, MM, CC, false, 4>), dim3(nbq), dim3(256), sh, st, b);      \
    }                                                                                                        \
    return hipGetLastError();                                                                                \
  }
      BARTRT_MC_LIST(BARTRT_QUAD)
#undef BARTRT_QUAD
    }
    if (kmode == "split" || (kmode.empty() && columns <= kSplitMaxColumns)) {
      b.ntiles = ntiles64;
      const size_t shs = sh + sizeof(double) * (1024 + 2 + 64);
#define BARTRT_SPLIT(MM, CC)                                                                               \
  if (a.M == MM && a.C == CC) {                                                                            \
    if (sq) hipLaunchKernelGGL((rt_eclipse_split<5, MM, CC, true>), dim3(nb64), dim3(128), shs, st, b);    \
    else hipLaunchKernelGGL((rt_eclipse_split<5, MM, CC, false>), dim3(nb64), dim3(128), shs, st, b);      \
    return hipGetLastError();                                                                              \
  }
      BARTRT_MC_LIST(BARTRT_SPLIT)
#undef BARTRT_SPLIT
    }
    b.ntiles = a.ntiles;
#define BARTRT_FAST(MM, CC)                                                                                \
  if (a.M == MM && a.C == CC) {                                                                            \
    if (sq) hipLaunchKernelGGL((rt_eclipse_fast<5, MM, CC, true>), dim3(nblocks), dim3(block), sh, st, b); \
    else hipLaunchKernelGGL((rt_eclipse_fast<5, MM, CC, false>), dim3(nblocks), dim3(block), sh, st, b);   \
    return hipGetLastError();                                                                              \
  }
    BARTRT_MC_LIST(BARTRT_FAST)
#undef BARTRT_FAST
  }
  if (a.A == 5) return launch_rt_t<5, -1, -1>(a, block, nblocks, sh, st);
  return launch_rt_t<0, -1, -1>(a, block, nblocks, sh, st);
}

}  // namespace bartrt
