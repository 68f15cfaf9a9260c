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
//  rt_eclipse_split  4-8 walkers: producer / consumer wave pair per column.
//  rt_eclipse_lp     1-3 walkers: one wave per chunk of layers per column.
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
// 157 columns per walker; microseconds per launch, layer-parallel / split / single-wave):
//   1 walker 26 / 39 / 54     3 walkers 39 / 44 / 49     4 walkers 51 / 47 / 50
//   6 walkers 65 / 54 / 55    8 walkers 79 / 68 / 75     9 walkers 91 / 80 / 80
//   10 walkers 101 / 85 / 86  (single-wave from here on: same time, half the waves)
constexpr long kLpMaxColumns = 500;
constexpr long kSplitMaxColumns = 1300;
constexpr int kLpChunk = 13;  // layers per wave of the layer-parallel kernel

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
  prep_body(p, w, sm);
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
    const double B = bnum * rcp_core(exp_core(fmin(c[1] * nu, 700.0)) - 1.0);
    const double hb = active ? 0.5 * (Bprev + B) : 0.0;
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      if (AT <= 0 && a >= A) break;
      const double E = exp_core(fmax(-tau * p.invmu[a], kExpMin));
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
  auto layer = [&](int k, const double (&r)[NR]) {
    const int kc = k < kend ? k : kend;
    const bool live = active && k <= kend;
    // the layer record is read from LDS in one batch (all reads issued
    // together, one wait); masking by multiplication keeps the reads of the
    // path length and of c2/T out of conditional blocks
    const double *c = sC + kc * NC;
    double cf[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) cf[j] = c[j];
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
    exp_core_n<AE + 1>(xs, ex);
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
  for (int k0 = 0; k0 <= kend; k0 += 4) {
    load_layer(clampk(k0 + 2), b0);
    load_layer(clampk(k0 + 3), b1);
    layer(k0, a0);
    layer(k0 + 1, a1);
    load_layer(clampk(k0 + 4), a0);
    load_layer(clampk(k0 + 5), a1);
    layer(k0 + 2, b0);
    layer(k0 + 3, b1);
    if (!__any(active)) break;
  }
  double F = 0.0;
  const bool surf = p.cloud_on && active;
#pragma unroll
  for (int a = 0; a < A; a++) F += p.wgt[a] * (I[a] + (surf ? Bprev * fprev[a] : 0.0));
  if (valid) p.spec[(size_t)w * W + i] = F;
}

// Few-walker variant (4-8 walkers at W = 1e4; below that the layer-parallel
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
      const double B = bnum * rcp_core(exp_core(fmin(cf[1] * nu, 700.0)) - 1.0);
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
          exp_core_n<AE>(xs, ex);
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
// Layer-parallel variant for small and medium batches.  Only the running sum
// tau_k is serial along a column; the extinction of a layer, its Planck term
// and its A transmittances are independent once tau_k is known.  A workgroup
// therefore takes ONE 64-wavenumber column and gives every wave a chunk of CH
// consecutive layers:
//   phase A  table loads + extinction of the chunk's layers, optical depth
//            relative to the chunk's first layer (kept in registers), and the
//            chunk's first / last extinction, total and maximum -> LDS
//   barrier
//   phase B  every wave rebuilds the optical depth at its chunk's start from
//            the published totals (same chain in every wave), finds out whether
//            the column was cut (`toomuch`) above it, and sums its layers' terms
//            of the transmittance trapezoid; per-wave partial fluxes meet in LDS.
// Work per column is that of the single-wave kernel plus one boundary layer per
// chunk, but it comes in L/CH short waves: 10 walkers x 157 columns become
// 12 560 waves that the dispatcher spreads evenly over the 1 024 SIMDs (1 570
// long waves leave half of the SIMDs with two and half with one), and a lone
// walker's latency drops by about the number of chunks.
template <int AT, int MT, int CT, int CH, bool SQ>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(2, BARTRT_WPE)))
void rt_eclipse_lp(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 3 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C;
  constexpr int NR = NLD > 0 ? NLD : 1;
  const int L = p.L, W = p.W;
  int tile, w;
  block_to_work(blockIdx.x, p.nwalkers, tile, w);
  if (tile >= p.ntiles) return;
  const int nwv = blockDim.x >> 6;  // waves = chunks

  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  double *sPub = smem + (size_t)L * NC + (size_t)L * NI;  // [chunk][first, last, total, max][64]
  double *sF = sPub + (size_t)nwv * 256;                  // [chunk][64] partial fluxes
  const int lane = threadIdx.x & 63;
  const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k0 = c * CH;
  const int i = tile * 64 + lane;
  const bool valid = i < W;
  const unsigned ii = valid ? (unsigned)i : (unsigned)(W - 1);
  const int kend = p.kstop[w];
  const double nu = p.wn[ii];
  auto clampk = [&](int k) { return k < kend ? k : kend; };
  {
    // every wave stages the records of its own chunk and reads them back without
    // a workgroup barrier; the other chunks' records are only needed after the
    // barrier that ends phase A
    const int n = (L - k0 < CH ? L - k0 : CH);
    const double *gC = p.coef + ((size_t)w * L + k0) * NC;
    const idx_t *gI = p.idx + ((size_t)w * L + k0) * NI;
    stage2_to_lds(sC + k0 * NC, gC, n * NC, sI + k0 * NI, gI, n * NI, lane, 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  // ---------------- phase A ----------------
  double tl[CH];  // tau_{k0+j} - tau_{k0}
#pragma unroll
  for (int j = 0; j < CH; j++) tl[j] = 0.0;
  if (k0 <= kend) {
    const double nu4 = (nu * nu) * (nu * nu);
    const TableLoader<M, C> tab(p, ii, sI);
    auto load_layer = [&](int k, double (&r)[NR]) { tab.load(k, r); };
    // D layers of loads are requested ahead of the arithmetic (the loop is fully
    // unrolled, and the compiler hoists further loads as registers allow:
    // D = 2..6 measured the same)
    constexpr int D = 2 < CH ? 2 : CH - 1;
    double r[D + 1][NR];
#pragma unroll
    for (int j = 0; j < D; j++) load_layer(clampk(k0 + j), r[j]);
    double efirst = 0.0, eprev = 0.0, t = 0.0, tmax = 0.0;
#pragma unroll
    for (int j = 0; j < CH; j++) {
      const int k = k0 + j;
      if (j + D < CH) load_layer(clampk(k + D), r[(j + D) % (D + 1)]);
      const double *cr = sC + clampk(k) * NC;
      double cf[NC];
#pragma unroll
      for (int q = 0; q < NC; q++) cf[q] = cr[q];
      double e = cf[2 + 2 * M + 2 * C] * nu4;
#pragma unroll
      for (int q = 0; q < NLD; q++) e = fma(cf[2 + q], r[j % (D + 1)][q], e);
      if (j == 0) {
        efirst = e;
      } else {
        t += (eprev + e) * cf[0] * (k <= kend ? 0.5 : 0.0);
        tmax = fmax(tmax, t);
      }
      tl[j] = t;
      eprev = e;
    }
    double *pb = sPub + c * 256 + lane;
    pb[0] = efirst;
    pb[64] = eprev;
    pb[128] = t;
    pb[192] = tmax;
  }
  __syncthreads();

  // ---------------- phase B ----------------
  double F = 0.0;
  if (k0 <= kend) {
    // optical depth at this chunk's first layer (t0) and at the layer above it
    // (tup), and whether the column was cut before this chunk
    double t0 = 0.0, tup = 0.0;
    bool cut = false;
    for (int cc = 0; cc < c; cc++) {
      const double *pb = sPub + cc * 256 + lane;
      cut = cut || (t0 + pb[192] > p.toomuch);
      tup = t0 + pb[128];
      const int kb = (cc + 1) * CH;  // <= k0 <= kend
      t0 = tup + (pb[64] + pb[256]) * sC[kb * NC] * 0.5;
    }
    if (__any(!cut)) {
      const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
      const double tcap = tau_cap(p, A);
      constexpr int AE = SQ ? A - 1 : A;
      double I[A], fprev[A], Bprev = 0.0;
#pragma unroll
      for (int a = 0; a < A; a++) { I[a] = 0.0; fprev[a] = 1.0; }
      if (c > 0) {
        double xs[AE + 1], ex[AE + 1];
        xs[AE] = fmin(sC[(k0 - 1) * NC + 1] * nu, 700.0);
#pragma unroll
        for (int a = 0; a < AE; a++) xs[a] = -fmin(tup, tcap) * p.invmu[a];
        exp_core_n<AE + 1>(xs, ex);
        Bprev = bnum * rcp_core(ex[AE] - 1.0);
#pragma unroll
        for (int a = 0; a < AE; a++) fprev[a] = ex[a];
        if (SQ) fprev[A - 1] = ex[0] * ex[0];
      }
#pragma unroll
      for (int j = 0; j < CH; j++) {
        const int k = k0 + j;
        const bool live = !cut && k <= kend;
        const double tau = t0 + tl[j], tc = fmin(tau, tcap);
        double xs[AE + 1], ex[AE + 1], es[A];
        xs[AE] = fmin(sC[clampk(k) * NC + 1] * nu, 700.0);
#pragma unroll
        for (int a = 0; a < AE; a++) xs[a] = -tc * p.invmu[a];
        exp_core_n<AE + 1>(xs, ex);
#pragma unroll
        for (int a = 0; a < AE; a++) es[a] = ex[a];
        if (SQ) es[A - 1] = ex[0] * ex[0];
        const double B = bnum * rcp_core(ex[AE] - 1.0);
        const double hb = (Bprev + B) * (live ? 0.5 : 0.0);
#pragma unroll
        for (int a = 0; a < A; a++) {
          I[a] = fma(hb, fprev[a] - es[a], I[a]);
          fprev[a] = es[a];
        }
        Bprev = B;
        cut = cut || (live && tau > p.toomuch);
        if (k == kend && p.cloud_on) {  // deck reached below toomuch: its surface emission
          const double bs = cut ? 0.0 : B;
#pragma unroll
          for (int a = 0; a < A; a++) I[a] = fma(bs, es[a], I[a]);
        }
        if (!__any(!cut)) break;
      }
#pragma unroll
      for (int a = 0; a < A; a++) F = fma(p.wgt[a], I[a], F);
    }
  }
  sF[c * 64 + lane] = F;
  __syncthreads();
  if (c == 0) {
    double s = 0.0;
    for (int cc = 0; cc < nwv; cc++) s += sF[cc * 64 + lane];
    if (valid) p.spec[(size_t)w * W + i] = s;
  }
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
  static const bool allow_sq = [] {
    const char *e = std::getenv("BARTRT_SQ");  // 0: always evaluate every transmittance (A/B runs)
    return !(e && e[0] == '0');
  }();
  // (the specialised kernels rebuild their buffer descriptor per layer, so the
  // table may be of any size; one layer's pair of planes must stay below 4 GB)
  const bool plane_ok = 2ull * a.M * a.W * 8ull < (1ull << 31);
  if (kmode != "generic" && a.A == 5 && !a.ext && !a.intens_out && !a.tau_out && plane_ok) {
    RtArgs b = a;
    const bool sq = allow_sq && order_angles_for_square(b);
    // too few single-wave columns to load the 1 024 SIMDs evenly -> several
    // waves per 64 wavenumbers: one per chunk of layers (layer-parallel), or a
    // producer / consumer pair
    const long columns = (long)a.nwalkers * ((a.W + 63) / 64);
    const int ntiles64 = (a.W + 63) / 64;
    const int nb64 = (ntiles64 + 7) / 8 * 8 * a.nwalkers;
    static const int lp_ch_env = [] {
      const char *e = std::getenv("BARTRT_LP_CH");
      return e ? std::atoi(e) : 0;
    }();
    const int lp_ch = lp_ch_env > 0 ? lp_ch_env : kLpChunk;
    const int lp_waves = (a.L + lp_ch - 1) / lp_ch;
    if ((kmode == "lp" || (kmode.empty() && columns <= kLpMaxColumns)) && lp_waves <= 16) {
      b.ntiles = ntiles64;
      const size_t shl = sh + sizeof(double) * (size_t)lp_waves * (256 + 64);
#define BARTRT_LP_CH(MM, CC, CHH)                                                                          \
  if (a.M == MM && a.C == CC && lp_ch == CHH) {                                                            \
    if (sq) hipLaunchKernelGGL((rt_eclipse_lp<5, MM, CC, CHH, true>), dim3(nb64), dim3(64 * lp_waves), shl, st, b);  \
    else hipLaunchKernelGGL((rt_eclipse_lp<5, MM, CC, CHH, false>), dim3(nb64), dim3(64 * lp_waves), shl, st, b);    \
    return hipGetLastError();                                                                              \
  }
#define BARTRT_LP(MM, CC) BARTRT_LP_CH(MM, CC, 13)
      BARTRT_MC_LIST(BARTRT_LP)
#ifdef BARTRT_LP_EXPERIMENT  // chunk-size A/B (BARTRT_LP_CH): 7, 10, 17 lose; 25 about equal
      BARTRT_LP_CH(4, 1, 7) BARTRT_LP_CH(4, 1, 10) BARTRT_LP_CH(4, 1, 17) BARTRT_LP_CH(4, 1, 25)
#endif
#undef BARTRT_LP
#undef BARTRT_LP_CH
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
