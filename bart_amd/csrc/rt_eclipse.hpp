// Specialised eclipse kernels (compile-time ray-angle / molecule / CIA counts and
// integration rule), shared by the per-rule translation units rt_eclipse_i*.hip.
//
//  rt_eclipse_fast   one lane per (walker, wavenumber) walks the layers from the
//                    top: buffer loads with scalar plane offsets, two pairs of
//                    register slots in flight, layer records from LDS.
//  rt_eclipse_split  5-8 walkers: producer / consumer wave pair per column.
//  rt_eclipse_quad   1-4 walkers: 16 wavenumbers x 4 layers (or 8 x 8) per wave
//                    and step, the optical depth by a lane-row prefix scan.
//
// INTEG: the integration rule (integ.hpp).  Rule 1 is the default and what the bench
// runs; its single-wave walk is its own kernel (rt_eclipse_s1.hpp, rt_eclipse_s1s.hpp for
// the per-angle `cut slant`); rules 0 and 2 are the walks below with their accumulators.
#pragma once
#include "integ.hpp"
#include "kernels.hpp"
#include "prep.hpp"

#ifndef __HIPCC_RTC__
#include <cmath>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <utility>
#endif

// upper bound on resident waves per SIMD the specialised kernels are compiled
// for: lets the compiler spend registers on loads in flight (measured best: 3-4)
#ifndef BARTRT_WPE
#define BARTRT_WPE 4
#endif

namespace bartrt {

// Kernel choice by 64-wavenumber columns per launch (measured at W = 1e4, L = 100:
// 157 columns per walker; microseconds per launch, quad-layer / split / single-wave,
// round 2, tools/ab_kernels.py):
//   1 walker 19 (8 rows) / 33 / 37   2 walkers 26 (8 rows) / 43 / 38   3: 31 / 43 / 39
//   4 walkers 39 / 45 / 41           5 walkers 48 / 47 / 44            6: 49 / 49 / 47
//   7 walkers 59 / 57 / 62           8 walkers 62 / 61 / 63            9: 68 / 70 / 67
//   10 walkers 76 / 76 / 71  (single-wave from here on)
// i.e. quad-layer while the columns leave SIMDs empty, single-wave while every column
// finds a SIMD of its own (<= 1 024), the producer/consumer pair for the first columns
// that have to share one, single-wave beyond.
// Under rule 1 (round 3, same tool with BARTRT_INTEG=1; quad-layer / split / rt_eclipse_simpson):
//   1 walker 22 (8 rows) / 38 / 38   2: 30 (8 rows) / 49 / 40   3: 39 / 50 / 41   4: 50 / 51 / 43
//   5: 60 / 54 / 47   6: 64 / 58 / 50   7: 78 / 74 / 64   8: 81 / 78 / 68   9: 91 / 79 / 69   10: 103 / 102 / 71
// -- the single-wave kernel from four walkers on, the producer / consumer pair never.
constexpr long kQuadMaxColumns = 640, kQuadMaxColumnsSimpson = 480;
// `cut slant`, rules 0 and 2: one ray per lane (rt_eclipse_quad<..., RAYS>) while the launch is a few thousand (walker,
// wavenumber) pairs -- it redoes the extinction five times over, so on the 1e4-sample grid ONE walker already takes
// what the single-wave kernel takes (63 against 58 us, +35 us per further walker).  (Rule 1 took this form on the demo
// shape until the all-rays form below got 16 and 32 rows: one walker 25.1 us against 15.3.)
constexpr long kQuadRaysMaxColumns = 80;
constexpr long kOctoRaysMaxColumns = 40;
// rule 1 under `cut slant`: all rays per lane in the layer-parallel walk (rt_eclipse_quad<..., ALLR>), R = 32 / 16 / 8
// layers per step by the number of 64-wide columns of the launch: the form holds two waves per SIMD (200 registers),
// 2 048 on the chip, and a launch of more waves than that runs in rounds -- so R <= 2 048 / columns, and the
// single-wave kernel beyond 256 columns.  us per RT launch (tools/ab_small.py, tools/ab_rows.sh):
//   demo shape (W = 2 501, one molecule), walkers 1 .. 5 = 40 .. 200 columns
//     R = 32: 15.3 18.1 23.7 26.7 33.9   R = 16: 18.1 21.5 21.4 25.9 34.2   R = 8: 26.4 26.7 26.9 32.7 32.6
//     one ray per lane, R = 8: 25.1 36.6 48.9 60.3 73.0
//   bench shape at W = 5 000, walkers 1 .. 3 = 79 / 158 / 237 columns
//     R = 32: 29.4 40.8 56.8   R = 16: 25.3 38.6 44.0   R = 8: 29.6 37.4 37.2   single wave: 55.8 56.0 56.0
//   bench shape (W = 1e4), one walker = 157 columns: R = 32 / 16 / 8 / 4 / team / single wave 42.8 / 40.2 / 37.6 / 47 / 43 / 58;
//     two walkers = 314 columns: R = 8 58.4, single wave 58.0
// The preparation folded into these kernels' prologue (every workgroup builds its walker's records itself: 6 us of
// latency-shaped work instead of a prep_profiles launch, 8 us + a boundary) pays while the workgroups run in ONE round
// (round 5, us per step, folded / not: demo shape one walker, 313 workgroups, 25.1 / 27.9, three walkers, 471: 29.4 /
// 32.2; two walkers, 626: 37.9 / 35.4, four, 628: 47.6 / 45.4; W = 1e4 one walker, 625: 48.2 / 47.4, two: 68.5 / 64.0)
constexpr int kFoldMaxWorkgroups = 512;
// (which of these forms serves which launch: kernel_table.inc, below.  The figures above are what its first version
// -- R = 32 to 64 columns, 16 to 128, 8 to 256; with one or two molecules 32 to 96, 16 to 176 -- was read from.)
constexpr long kQuadAllMaxColumns = 256;    // the range of the layer-parallel walk under BARTRT_ALLR_ROWS (the A/B tools' switch)
// `cut slant`, rule 1: a team of three waves per column (rt_eclipse_s1t.hpp) for ONE walker's worth of columns at
// W = 1e4 -- 44 against the single-wave kernel's 58 us; from two walkers on the team loses (59 / 59, four walkers 76 /
// 61, ten 135 / 98, 64: 639 / 429 us): its producer wave keeps its table loads one layer ahead only (the 128
// registers that five teams per CU allow), the Planck term and the panel weights are computed twice, and three
// waves meet at a barrier every six layers
constexpr long kTeamMaxColumns = 0;   // (round 4, later: the all-rays quad kernel takes that range at 38 us; the team stays behind BARTRT_KERNEL=team)
constexpr long kOctoMaxColumns = 400;  // eight layers per step (R = 8) below this
constexpr long kSplitMinColumns = 1025, kSplitMaxColumns = 1300;
constexpr long kIlpMaxColumns = 20000;  // single-wave kernel: the ILP-scheduled build below this (128 walkers at W = 1e4)

#ifndef __HIPCC_RTC__   // (host side: the launchers' business)
// Rule 1 under `cut slant` (the default conventions): the variant comes from a MEASURED table (kernel_table.inc, written
// by tools/tune_kernels.py from a sweep on the box; round 6 -- until then a thicket of hand-measured column intervals in
// launch_rt_spec).  tests/test_gpu_kernel_choice.py holds the default choice to within 7 % of the best forced variant
// on grids the table was not tuned on.
enum KernelVariant { kVarSingle = 0, kVarRows4 = 4, kVarRows8 = 8, kVarRows16 = 16, kVarRows32 = 32, kVarAdj8 = 108, kVarAdj16 = 116 };
struct KernelChoice { long max_columns; int variant, fallback; };
constexpr long kAllColumns = 0x7fffffffffffffffL;
// SCHED of rt_eclipse_simpson_slant (rt_eclipse_s1s.hpp) by shape: the record read-ahead up to twelve table loads per
// layer, none beyond (rt_eclipse_slant_ilp.hip)
constexpr int slant_sched(int M, int C, int dflt) { return 2 * M + 2 * C > 12 ? 0 : dflt; }
#include "kernel_table.inc"
inline const KernelChoice &slant_simpson_choice(int M, long columns) {
  static constexpr const KernelChoice *const tables[6] = {kSlantSimpsonM1, kSlantSimpsonM2, kSlantSimpsonM3,
                                                          kSlantSimpsonM4, kSlantSimpsonM5, kSlantSimpsonM6};
  const KernelChoice *t = tables[M < 1 ? 0 : M > 6 ? 5 : M - 1];   // (no table molecules: cross sections only -- the lightest table)
  int i = 0;
  while (columns > t[i].max_columns) i++;   // (the last entry holds kAllColumns)
  return t[i];
}
inline const char *kernel_variant_name(int v) {
  switch (v) {
    case kVarRows4: return "rows4"; case kVarRows8: return "rows8"; case kVarRows16: return "rows16"; case kVarRows32: return "rows32";
    case kVarAdj8: return "adj8"; case kVarAdj16: return "adj16"; default: return "single";
  }
}
#endif

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

// LDS beyond the layer records: rule 1 keeps the Simpson weights of the radius grid (three
// words per layer in the producer / consumer and quad-layer kernels, four in rt_eclipse_simpson)
template <int INTEG>
__host__ __device__ inline size_t integ_lds_doubles(int L) {
  return INTEG == kIntegSimpson ? 4 * (size_t)(L + kSimpsonPad) : 0;
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
// SCHED only names the instantiation: SCHED = 1 is compiled in its own translation
// unit (rt_eclipse_i0_ilp.hip) under the compiler's maximum-ILP scheduling strategy.
// The default (maximum-occupancy) schedule lays the A + 1 exponentials' Horner
// chains out one after the other -- nine dependent fp64 FMAs in a row, five times
// -- which a SIMD with three resident waves hides and a SIMD with one or two does
// not; the ILP schedule keeps the chains interleaved as written, at 211 instead of
// 138 VGPRs (two resident waves per SIMD instead of three).  Measured on the bench
// grid: 10 walkers 75 -> 71 us, 16: 113 -> 107, 64: 326 -> 324, 256: 1062 -> 1082;
// launch_rt_spec takes the ILP build below kIlpMaxColumns columns.
// EXT: the line-by-line path's hand-off -- the layer's line extinction ext[w][l][W]
// (atm layer order) is one more coalesced 8-byte load per layer and one more addend
// (line-by-line engines have no table: MT = 0).
// SLANT: the `toomuch` cut on each ray's slant depth (cfg `cut slant`): the rays' terms are masked
// one by one (ColumnFluxSlant, integ.hpp) and the column is walked while its longest-lived ray is alive.
template <int AT, int MT, int CT, bool SQ, int INTEG, int SCHED = 0, bool EXT = false, bool SLANT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, BARTRT_WPE)))
void rt_eclipse_fast(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 4 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C;
  constexpr int NR = NLD + (EXT ? 1 : 0) > 0 ? NLD + (EXT ? 1 : 0) : 1;
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
  const double *sW = smem + (size_t)L * NC + (size_t)L * NI;  // rule 1 only
  {
    const double *gC = p.coef + (size_t)w * L * NC;
    const idx_t *gI = p.idx + (size_t)w * L * NI;
    stage2_to_lds(sC, gC, L * NC, sI, gI, L * NI, threadIdx.x, blockDim.x);
  }
  __syncthreads();
  if (INTEG == kIntegSimpson) {
    simpson_radius_weights(const_cast<double *>(sW), sC, NC, L, threadIdx.x, blockDim.x);
    __syncthreads();
  }

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
    if (EXT) r[NLD] = extw[(size_t)(L - 1 - k) * W];
  };

  TauColumn<INTEG> tc;
  std::conditional_t<SLANT, ColumnFluxSlant<INTEG, A>, ColumnFlux<INTEG, A>> ci(p);
  double Bprev = 0.0;
  bool active = true;
  const int kraw = p.kstop[w], kend = kstop_layer(kraw);
  const bool deck_on = kstop_deck(kraw);
  const double tcap = tau_cap(p, A);
  // the optical depth that ends the lane's walk: toomuch itself, or (SLANT) the threshold of the ray that goes last
  double tstop = p.toomuch;
  if constexpr (SLANT) {
    tstop = p.thr[0];
#pragma unroll
    for (int a = 1; a < A; a++) tstop = p.thr[a] > tstop ? p.thr[a] : tstop;
  }

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
    double e = fma(cf[2 + 2 * M + 2 * C], nu4, cf[3 + 2 * M + 2 * C]);   // Rayleigh + grey cloud
#pragma unroll
    for (int j = 0; j < NLD; j++) e = fma(cf[2 + j], r[j], e);
    if (EXT) e += r[NLD];
    tc.layer(k, live, lv, e, cf[0], sW);
    // Planck exponent and the A slant-path exponents in one interleaved batch
    const double tcl = fmin(tc.tau, tcap);
    constexpr int AE = SQ ? A - 1 : A;  // transmittances that need an exponential
    double xs[AE + 1], ex[AE + 1], es[A];
    xs[AE] = fmin(cf[1] * nu, 700.0);
#pragma unroll
    for (int a = 0; a < AE; a++) xs[a] = -tcl * p.invmu[a];
    exp_rt_n<AE + 1>(xs, ex);
#pragma unroll
    for (int a = 0; a < AE; a++) es[a] = ex[a];
    if (SQ) es[A - 1] = ex[0] * ex[0];
    const double B = bnum * rcp_n1(ex[AE] - 1.0);
    ci.layer(p, A, live, lv, tc.tau, Bprev, B, es);
    Bprev = SLANT ? (live ? B : Bprev) : B;
    active = active && !(live && tc.tau > tstop);
  };
  auto clampk = [&](int k) { return k < kend ? k : kend; };

  // two pairs of slots; each pair is reloaded two layers before it is used and
  // the loads that cross the loop's back edge were issued two layers earlier
  double a0[NR], a1[NR], b0[NR], b1[NR];
  load_layer(clampk(0), a0);
  load_layer(clampk(1), a1);
  double cfE[NC], cfO[NC];
  read_rec(0, cfE);
  int k0 = 0;
  for (; k0 <= kend; k0 += 4) {
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
  const double F = ci.flux(p, A, SLANT ? deck_on : (deck_on && active), Bprev, L);
  if (valid) p.spec[(size_t)w * W + i] = F;
  if (p.walked_out && threadIdx.x == 0)  // diagnostics: layers this wave walked (bench.py's byte model)
    p.walked_out[(size_t)w * p.ntiles + tile] = (k0 + 4 < kend + 1 ? k0 + 4 : kend + 1);
}

// Few-walker variant (5-8 walkers at W = 1e4; below that the quad-layer
// kernel is faster still): the layer loop is split over TWO waves per 64
// wavenumbers.  Wave 0 (producer) streams the tables and advances the optical
// depth and the Planck term; wave 1 (consumer) turns each tau into the A
// transmittances and accumulates the intensities.  The halves are about equal
// in issue slots, so the serial time per layer halves at unchanged total work
// -- it pays while the single-wave columns cannot load the 1 024 SIMDs evenly
// (8 walkers: 68 vs 75 us; from 9 walkers on the single-wave kernel is as fast).
// Hand-off: an LDS ring of two 4-layer halves per lane -- rule 0:
// [tau, (B_{k-1}+B_k)/2 * live], rules 1 / 2: [tau, live ? B_k : -1] -- and ONE
// raw workgroup barrier per 4 layers (the consumer reads half b while the
// producer fills half b+1).
template <int AT, int MT, int CT, bool SQ, int INTEG>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, BARTRT_WPE)))
void rt_eclipse_split(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 4 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C;
  constexpr int NR = NLD > 0 ? NLD : 1;
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
  double *sX = smem + (size_t)L * NC + (size_t)L * NI;  // [2 halves][4 layers][tau, hb][64]
  int *sFlag = reinterpret_cast<int *>(sX + 1024);      // [half] producer saw every lane finished
  double *sEnd = sX + 1024 + 2;                         // [64] B of the last layer (cloud deck term)
  const double *sW = sEnd + 64;                         // rule 1 only
  {
    const double *gC = p.coef + (size_t)w * L * NC;
    const idx_t *gI = p.idx + (size_t)w * L * NI;
    stage2_to_lds(sC, gC, L * NC, sI, gI, L * NI, threadIdx.x, 128);
    if (threadIdx.x < 2) sFlag[threadIdx.x] = 0;
  }
  __syncthreads();
  if (INTEG == kIntegSimpson) {
    simpson_radius_weights(const_cast<double *>(sW), sC, NC, L, threadIdx.x, 128);
    __syncthreads();
  }

  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform
  const int i = tile * 64 + lane;
  const bool valid = i < W;
  const unsigned ii = valid ? (unsigned)i : (unsigned)(W - 1);
  const int kraw = p.kstop[w], kend = kstop_layer(kraw);
  const bool deck_on = kstop_deck(kraw);
  const int nblk = kend / 4 + 1;  // 4-layer blocks; both waves run the same count

  if (role == 0) {
    // ---------------- producer: extinction, tau, Planck ----------------
    const double nu = p.wn[ii];
    const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
    const double nu4 = (nu * nu) * (nu * nu);
    const TableLoader<M, C> tab(p, ii, sI);
    auto load_layer = [&](int k, double (&r)[NR]) { tab.load(k, r); };
    TauColumn<INTEG> tc;
    double Bprev = 0.0;
    bool active = true;
    auto layer = [&](int k, const double (&r)[NR]) {
      const int kc = k < kend ? k : kend;
      const bool live = active && k <= kend;
      const double *c = sC + kc * NC;
      double cf[NC];
#pragma unroll
      for (int j = 0; j < NC; j++) cf[j] = c[j];
      const double lv = live ? 0.5 : 0.0;
      double e = fma(cf[2 + 2 * M + 2 * C], nu4, cf[3 + 2 * M + 2 * C]);   // Rayleigh + grey cloud
#pragma unroll
      for (int j = 0; j < NLD; j++) e = fma(cf[2 + j], r[j], e);
      tc.layer(k, live, lv, e, cf[0], sW);
      const double B = bnum * rcp_n1(exp_rt(fmin(cf[1] * nu, 700.0)) - 1.0);
      double *slot = sX + (k & 7) * 128;   // half (k/4)&1, layer k&3
      slot[lane] = tc.tau;
      slot[64 + lane] = INTEG == kIntegTransmittance ? (Bprev + B) * lv : (live ? B : -1.0);
      Bprev = B;
      active = active && !(live && tc.tau > p.toomuch);
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
    int blk = 0;
    for (; blk < nblk; blk++) {
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
    sEnd[lane] = (deck_on && active) ? Bprev : 0.0;
    handoff();
    if (p.walked_out && lane == 0)
      p.walked_out[(size_t)w * p.ntiles + tile] = (4 * blk + 4 < kend + 1 ? 4 * blk + 4 : kend + 1);
  } else {
    // ---------------- consumer: transmittances and intensities ----------------
    ColumnFlux<INTEG, A> ci(p);
    const double tcap = tau_cap(p, A);
    for (int blk = 0; blk < nblk; blk++) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int stop = sFlag[blk & 1];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double *slot = sX + ((blk & 1) * 4 + u) * 128;
        const double tau = slot[lane], hb = slot[64 + lane];
        const double tcl = fmin(tau, tcap);
        constexpr int AE = SQ ? A - 1 : A;
        double xs[AE], es[A];
#pragma unroll
        for (int a = 0; a < AE; a++) xs[a] = -tcl * p.invmu[a];
        {
          double ex[AE];
          exp_rt_n<AE>(xs, ex);
#pragma unroll
          for (int a = 0; a < AE; a++) es[a] = ex[a];
          if (SQ) es[A - 1] = ex[0] * ex[0];
        }
        if constexpr (INTEG == kIntegTransmittance) {
          const double G = angle_sum<A>(p, es);     // the producer's hb already carries the 1/2 and the mask
          ci.F = fma(hb, ci.Gprev - G, ci.F);
          ci.Gprev = G;
        } else {
          const bool live = hb >= 0.0;   // the producer's "layer counts" flag
          ci.layer(p, A, live, live ? 0.5 : 0.0, tau, 0.0, live ? hb : 0.0, es);
        }
      }
      if (__builtin_amdgcn_readfirstlane(stop)) break;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const double bsurf = sEnd[lane];   // > 0: the deck was reached below toomuch
    const double F = ci.flux(p, A, bsurf != 0.0, bsurf, L);
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
//
// Rule 1 (Simpson) in this layout: the panel that ends on layer j is evaluated by
// the lane that owns j, from the values of the rows q-1 and q-2 (the two carries
// of rows 0 / 1 come from the previous step); R is even, so a lane's panels all
// end on the parity of q and each lane keeps ONE sum.  The padded point falls to
// the row after the one that passed `toomuch` (row 0 of the next step when that
// was the last row), and at the end the rows whose parity is that of the last
// point's index are added up.  The optical depth is the prefix sum of the even
// rows' radius panels plus, on odd rows, the trapezoid of the last interval.
//
// RAYS: one ray angle per lane -- a wave's 64 / R columns are (wavenumber, ray) pairs instead of wavenumbers (R = 4,
// five angles: three wavenumbers x five rays, one column idle).  Every lane then runs the single-ray form of the
// walk with its own ray's constants: one transmittance, ITS cut (`cut slant`: the ray's threshold RtArgs::thr, so the
// ballots that find a column's cut and the row that carries the padded point work per ray as they stand) and ITS pad
// width (mu of vertical depth = one unit of slant depth); the five rays of a wavenumber meet in a five-lane sum at
// the very end.  Extinction, optical depth and Planck term are computed five times over -- on launches that leave
// most of the chip idle anyway (one to three walkers) -- in exchange for a layer-parallel walk whose per-step work
// is a fifth of the five-ray lane's: this is how `cut slant`, whose rays cannot share a layer sum, keeps a
// few-walker kernel (the single-wave slant kernel walks a column's 100 layers serially: 50 us at any small batch).
//
// ALLR (rule 1, `cut slant`): all rays in every lane.  What `cut slant` takes away from this kernel is the single
// integrand that crossed the lane rows -- each ray needs its own last two integrands from the rows below -- not the
// layer parallelism: the rays' alive flags need no ballots at all (a ray is alive on layer j iff the largest optical
// depth of the layers above, an exclusive prefix MAXIMUM across the rows on top of the previous steps' carry, is at or
// below its threshold RtArgs::thr), the row after a ray's last layer is its padded point (alive at the start of layer
// j - 1, dead at the start of j: the same two prefix maxima), whose panel takes weights of its own (one unit of slant
// depth = mu_a of vertical depth: computed only in waves where some lane pads that ray), and every lane keeps one sum
// per ray (its layers all have the parity of its row).  At the end each ray's last point index is the largest padded
// row of its column (or the column's last layer) and the rows of that parity are added up.  Ten values cross the rows
// per step instead of two; extinction, optical depth and Planck term are computed once per (layer, wavenumber) --
// which the one-ray-per-lane form (RAYS) does five times.
// (BARTRT_QUAD_WPE, A/B builds: waves per SIMD the register allocation is held to.  Three -- 168 registers instead of
// 170-188 -- was measured in round 5: two walkers on the demo shape 23.8 -> 20.2 us, one walker 18.9 -> 19.6; left as it was.)
#ifndef BARTRT_QUAD_WPE
#define BARTRT_QUAD_WPE 1
#endif
template <int AT, int MT, int CT, bool SQ, int R, int INTEG, bool RAYS = false, bool ALLR = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BARTRT_QUAD_WPE))) void rt_eclipse_quad(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int A = AT, M = MT, C = CT;
  constexpr int NC = 4 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C, NR = NLD > 0 ? NLD : 1;
  constexpr int AL = RAYS ? 1 : A;                    // ray angles a lane evaluates
  constexpr int AE = RAYS ? 1 : (SQ ? A - 1 : A);     // transmittances that need an exponential
  constexpr int WN = 64 / R;                          // columns per wave
  constexpr int WNR = RAYS ? WN / A : WN;             // wavenumbers per wave
  static_assert(!RAYS || (WNR >= 1 && !SQ), "RAYS: the ray grid fits a lane row; no squared-transmittance shortcut");
  static_assert(!ALLR || (INTEG == kIntegSimpson && !RAYS), "ALLR: rule 1, all rays per lane");
  constexpr bool SIMPSON = INTEG == kIntegSimpson;
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

  // (16 rows and up: the rows of a wave read 16 / 32 different layer records at once -- an even record stride would
  // put them on 2-4 of the 64 LDS banks' positions; one double of padding per record spreads them over all)
  constexpr int NCS = (R >= 16 && NC % 2 == 0) ? NC + 1 : NC;
  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NCS);
  const double *sW = smem + (size_t)L * NCS + (size_t)L * NI;  // rule 1 only
  // nprep < 0: this launch prepares its own walkers -- every workgroup builds its walker's layer records in LDS itself
  // (prep_body, the same code and bits as the prep_profiles launch it replaces; launch_rt_folded)
  __shared__ int sKstopFold;
  const bool fold = p.nprep < 0;
  if (fold) {
    PrepFold pf;
    pf.coef = sC; pf.stride = NCS; pf.idx = sI; pf.kstop = &sKstopFold; pf.global = tile == 0;
    prep_block(p.prep_next, w, smem + (size_t)L * NCS + (size_t)L * NI + integ_lds_doubles<INTEG>(L), pf);
  } else if constexpr (NCS == NC) {
    stage2_to_lds(sC, p.coef + (size_t)w * L * NC, L * NC, sI, p.idx + (size_t)w * L * NI, L * NI,
                  threadIdx.x, 256);
  } else {
    const double *src = p.coef + (size_t)w * L * NC;
    const idx_t *srci = p.idx + (size_t)w * L * NI;
    for (int k = threadIdx.x; k < L * NC; k += 256) sC[(k / NC) * NCS + k % NC] = src[k];
    for (int k = threadIdx.x; k < L * NI; k += 256) sI[k] = srci[k];
  }
  __syncthreads();
  if (SIMPSON) {
    simpson_radius_weights(const_cast<double *>(sW), sC, NCS, L, threadIdx.x, 256);
    __syncthreads();
  }

  const int lane = threadIdx.x & 63;
  const int q = lane / WN, m = lane % WN;
  const int ray = RAYS ? m % A : 0, mw = RAYS ? m / A : m;   // column m = wavenumber mw (and ray)
  const bool col_on = !RAYS || m < WNR * A;                  // (RAYS: the columns past the last whole wavenumber idle)
  const int i0 = (tile * 4 + (threadIdx.x >> 6)) * WNR;  // this wave's first wavenumber
  if (i0 >= W) return;
  const unsigned ii = (col_on && i0 + mw < W) ? (unsigned)(i0 + mw) : (unsigned)(W - 1);
  // this lane's ray (RAYS) -- or, without, the column's cut and pad as they always were
  double invmu_l = p.invmu[0], wgt_l = p.wgt[0], wq_l = p.wq[0], thr_l = p.toomuch, padw_l = 1.0;
  if constexpr (RAYS) {
    if (p.cut_slant) { thr_l = p.thr[0]; padw_l = p.mu[0]; }
#pragma unroll
    for (int a = 1; a < A; a++)
      if (ray == a) {
        invmu_l = p.invmu[a]; wgt_l = p.wgt[a]; wq_l = p.wq[a];
        if (p.cut_slant) { thr_l = p.thr[a]; padw_l = p.mu[a]; }
      }
  }
  const double nu = p.wn[ii];
  const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
  const double nu4 = (nu * nu) * (nu * nu);
  const int kraw = fold ? sKstopFold : p.kstop[w], kend = kstop_layer(kraw);
  const bool deck_on = kstop_deck(kraw);
  const double tcap = tau_cap(p, A);

  // per-lane table addressing: plane offset of the lane's layer + the lane's
  // wavenumber inside the plane (kernels.hpp, "Table layout")
  const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.cia), 0, (int)p.cia_bytes, 0x00020000);
  const unsigned vk = ii * 8u * (unsigned)M, vc = ii * 16u, planeB = (unsigned)M * (unsigned)W * 8u;
  auto load_layer = [&](int j, double (&r)[NR]) {
    const idx_t *ix = sI + j * NI;
    if (M > 0) {
      const idx_t mine = ix[0];
      const long long base = p.window ? row_window_base<R>(mine) : 0ll;
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

  // the lanes of one wavenumber: bits m, WN + m, 2 WN + m, ... of a ballot
  constexpr unsigned long long kColBits = R == 4 ? 0x0001000100010001ull : R == 8 ? 0x0101010101010101ull
                                          : R == 16 ? 0x1111111111111111ull : R == 32 ? 0x5555555555555555ull : ~0ull;
  const unsigned long long col_bits = kColBits << m;
  const unsigned long long below_bits = col_bits & ((1ull << (WN * q)) - 1ull);
  const int from_below = (lane + 64 - WN) & 63;       // row q - 1 (the last row for row 0)
  const int from_below2 = (lane + 64 - 2 * WN) & 63;  // row q - 2 (rule 1)

  double I = 0.0;   // this lane's layers' terms, angle quadrature taken (ColumnFlux, integ.hpp)
  double Fs = 0.0;  // surface term of a cloud deck (one lane per wavenumber sets it)
  // carries of row 0: extinction, Planck term, transmittances of the layer just
  // above this step (row R - 1 of the previous step), and the optical depth there
  double c_e = 0.0, c_B = 0.0, c_tau = 0.0;
  double c_G = RAYS ? wgt_l : p.wgt[0];   // rule 0: sum_a w_a E_a of the layer above (all transmittances 1 at the top)
  if constexpr (!RAYS) {
#pragma unroll
    for (int a = 1; a < A; a++) c_G += p.wgt[a];
  }
  // rule 1: second carries (row R - 2 / R - 1 of the previous step for rows 0 / 1),
  // the running even-index Simpson sum of the optical depth, the index of the last
  // point of the intensity integral and the "next step's row 0 is the padded point" flag
  double c2_e = 0.0, c2_tau = 0.0, c_y = 0.0, c2_y = 0.0, c_S = 0.0;
  int nend = kend;
  bool pad_next = false;
  // ALLR: per ray the lane's sum, the carries of the two integrands above the step, the padded row it evaluated;
  // the running maximum of tau over the layers above the step (and above its last row)
  [[maybe_unused]] double Ia[A], ca_y[A], ca2_y[A], c_tm = 0.0, c_tm1 = 0.0, thr_max = 0.0;
  [[maybe_unused]] int Npad[A];
  [[maybe_unused]] const int kcut = kend < L - 2 ? kend : L - 2;
  if constexpr (ALLR) {
    thr_max = p.thr[0];
#pragma unroll
    for (int a = 0; a < A; a++) { Ia[a] = ca_y[a] = ca2_y[a] = 0.0; Npad[a] = -1; thr_max = p.thr[a] > thr_max ? p.thr[a] : thr_max; }
  }
  // no layer above this step passed `toomuch` (per column, all rows agree); columns that hold no sample of the grid
  // (past its end, or the idle columns of RAYS) are dead from the start: they must not keep the wave walking
  bool active = col_on && i0 + mw < W;

  auto step = [&](int s, const double (&rv)[NR]) {
    const int j = R * s + q, jc = clampk(j);
    const bool inrange = j <= kend;
    const double *c = sC + jc * NCS;
    double cf[NC];
#pragma unroll
    for (int x = 0; x < NC; x++) cf[x] = c[x];
    double e = fma(cf[2 + 2 * M + 2 * C], nu4, cf[3 + 2 * M + 2 * C]);   // Rayleigh + grey cloud
#pragma unroll
    for (int x = 0; x < NLD; x++) e = fma(cf[2 + x], rv[x], e);
    // extinction of the layer above
    const double e_below = __shfl(e, from_below);
    const double eprev = q == 0 ? c_e : e_below;
    c_e = e_below;
    const double tau_above_step = c_tau;   // optical depth of row R - 1 of the previous step
    double tau;
    if constexpr (!SIMPSON) {
      // optical depth: R-lane prefix sum of the step's increments + the running value
      double v = (eprev + e) * cf[0] * ((inrange && active) ? 0.5 : 0.0);
#pragma unroll
      for (int d = 1; d < R; d <<= 1) {
        const double t = __shfl(v, (lane + 64 - d * WN) & 63);
        if (q >= d) v += t;
      }
      tau = c_tau + v;
    } else {
      // even rows contribute the radius panel that ends on them; odd rows add the
      // trapezoid of their last interval on top of the even sum below them
      const double e_below2 = __shfl(e, from_below2);
      const double e2 = q < 2 ? c2_e : e_below2;
      c2_e = e_below2;
      const double *wS = sW + 3 * jc;
      double v = ((q & 1) == 0 && j >= 2 && inrange && active)
                     ? fma(wS[0], e2, fma(wS[1], eprev, wS[2] * e)) : 0.0;
#pragma unroll
      for (int d = 1; d < R; d <<= 1) {
        const double t = __shfl(v, (lane + 64 - d * WN) & 63);
        if (q >= d) v += t;
      }
      const double S = c_S + v;
      c_S = __shfl(S, (R - 1) * WN + m);
      tau = (q & 1) ? fma((eprev + e) * cf[0], 0.5, S) : S;
    }
    c_tau = __shfl(tau, (R - 1) * WN + m);
    if constexpr (ALLR) {
      // ---- all rays per lane (see the comment above the kernel)
      // largest optical depth of the layers above this one / above the one above (layers past kcut do not count:
      // a ray ends only where a deeper layer exists)
      double pm = j <= kcut ? tau : 0.0;
#pragma unroll
      for (int d = 1; d < R; d <<= 1) {
        const double t = __shfl(pm, (lane + 64 - d * WN) & 63);
        if (q >= d) pm = fmax(pm, t);
      }
      const double incl = fmax(c_tm, pm);
      const double incl_b = __shfl(incl, from_below);
      const double excl = q == 0 ? c_tm : incl_b;            // over the layers < j
      const double excl_b = __shfl(excl, from_below);
      const double excl1 = q == 0 ? c_tm1 : excl_b;          // over the layers < j - 1
      c_tm1 = __shfl(excl, (R - 1) * WN + m);
      c_tm = __shfl(incl, (R - 1) * WN + m);
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
      // the tau grid of this lane's panel (j - 2, j - 1, j)
      const double tau_b1 = __shfl(tau, from_below);
      const double tau1 = q == 0 ? tau_above_step : tau_b1;
      const double tau_b2 = __shfl(tau, from_below2);
      const double tau2 = q < 2 ? c2_tau : tau_b2;
      c2_tau = tau_b2;
      // Simpson weights of the panel (simpson_tau_weights, integ.hpp, with the reciprocal of the upper half-width kept:
      // the rays' padded panels share that half)
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
        const double yb = __shfl(y[a], from_below), yb2 = __shfl(y[a], from_below2);
        const double y1 = q == 0 ? ca_y[a] : yb, y2 = q < 2 ? ca2_y[a] : yb2;
        ca_y[a] = yb;
        ca2_y[a] = yb2;
        const bool alive = inrange && excl <= p.thr[a];
        const bool pad = j >= 1 && j < L && !(excl <= p.thr[a]) && excl1 <= p.thr[a];   // the ray died on layer j - 1
        double cterm = fma(w0, y2, fma(w1, y1, w2 * y[a]));
        if (__any(pad)) {
          // one unit of slant depth (mu_a of vertical depth) past the last point, integrand 0 there: the panel's first
          // two weights, no division (1 / h0 is the step's, 1 / mu_a the ray's constant)
          const double hp = h0 + p.mu[a], p6 = hp * (1.0 / 6.0);
          const double v0 = deg0 ? 0.5 * h0 : p6 * (2.0 - p.mu[a] * r0);
          const double v1 = deg0 ? 0.5 * hp : p6 * (hp * hp * (r0 * p.invmu[a]));
          cterm = pad ? fma(v0, y2, v1 * y1) : cterm;
          Npad[a] = pad ? j : Npad[a];
        }
        Ia[a] += ((alive || pad) && j >= 1) ? cterm : 0.0;
        if (deck_on && j == kend && alive && !(tau > p.thr[a])) Fs = fma(p.wgt[a], y[a], Fs);   // deck reached below the ray's cut
      }
      active = c_tm1 <= thr_max;   // some ray was alive when the step's last row began: its pad or more layers may follow
      return;
    }
    // which layers of this step are still above the cut
    const unsigned long long over = __ballot(inrange && active && tau > thr_l);
    const bool live = inrange && active && (over & below_bits) == 0ull;
    // Planck term and transmittances of this lane's layer
    const double tcl = fmin(tau, tcap);
    double xs[AE + 1], ex[AE + 1];
    xs[AE] = fmin(cf[1] * nu, 700.0);
#pragma unroll
    for (int a = 0; a < AE; a++) xs[a] = -tcl * (RAYS ? invmu_l : p.invmu[a]);
    exp_rt_n<AE + 1>(xs, ex);
    const double B = bnum * rcp_n1(ex[AE] - 1.0);
    double E[AL];
#pragma unroll
    for (int a = 0; a < AE; a++) E[a] = ex[a];
    if (SQ) E[AL - 1] = E[0] * E[0];
    // the angle quadratures of this lane: all rays, or (RAYS) its one ray with the ray's weight
    auto gsum = [&]() { if constexpr (RAYS) return wgt_l * E[0]; else return angle_sum<AL>(p, E); };
    auto qsum = [&]() { if constexpr (RAYS) return wq_l * E[0]; else return angle_sum_q<AL>(p, E); };
    if constexpr (INTEG == kIntegTransmittance) {
      // the layer above: row q - 1, or the carry for row 0
      const double B_below = __shfl(B, from_below);
      const double Bprev = q == 0 ? c_B : B_below;
      c_B = B_below;
      const double hb = (Bprev + B) * (live ? 0.5 : 0.0);
      // rule 0 is linear in the transmittances: the angle quadrature first (ColumnFlux,
      // integ.hpp), so ONE value crosses the lane rows instead of one per ray angle
      const double G = gsum();
      const double G_below = __shfl(G, from_below);
      const double Gprev = q == 0 ? c_G : G_below;
      c_G = G_below;
      I = fma(hb, Gprev - G, I);
      if (deck_on && j == kend && live && !(tau > thr_l)) Fs = fma(B, G, Fs);  // deck reached below toomuch
    } else {
      // rules 1 / 2 are linear in Y = B sum_a (w_a / mu_a) E_a with angle-independent
      // weights (integ.hpp): ONE integrand per layer crosses the lane rows
      const double tau_b1 = __shfl(tau, from_below);
      const double tau1 = q == 0 ? tau_above_step : tau_b1;
      const double y = B * qsum();
      const double yb = __shfl(y, from_below);
      const double y1 = q == 0 ? c_y : yb;
      c_y = yb;
      if constexpr (INTEG == kIntegTrapzTau) {
        I = fma(y1 + y, (tau - tau1) * (live ? 0.5 : 0.0), I);
      } else {
        const double tau_b2 = __shfl(tau, from_below2);
        const double tau2 = q < 2 ? c2_tau : tau_b2;
        c2_tau = tau_b2;
        const double yb2 = __shfl(y, from_below2);
        const double y2 = q < 2 ? c2_y : yb2;
        c2_y = yb2;
        // the cut: the first row of this wavenumber that passed toomuch
        const unsigned long long mine = over & col_bits;
        const int f = mine ? (int)(__ffsll((long long)mine) - 1) / WN : -1;
        const bool cut_here = active && f >= 0;
        // the padded point: the row after the cut (row 0 of the next step when the
        // cut was the last row), as long as the atmosphere has a layer there
        const bool pad = j < L && ((cut_here && q == f + 1) || (pad_next && q == 0));
        if (cut_here) nend = R * s + f + ((R * s + f + 1 < L) ? 1 : 0);
        pad_next = cut_here && f == R - 1;
        const double x = pad ? tau1 + padw_l : tau;
        double w0, w1, w2;
        simpson_tau_weights(tau1 - tau2, x - tau1, w0, w1, w2);
        if (j == 1) { w0 = 0.0; w1 = 0.5 * (x - tau1); w2 = w1; }  // the first interval: a trapezoid
        const bool counts = (live || pad) && j >= 1;
        const double cterm = fma(w0, y2, fma(w1, y1, w2 * (pad ? 0.0 : y)));
        I += counts ? cterm : 0.0;
      }
      if (deck_on && j == kend && live && !(tau > thr_l))   // deck reached below toomuch
        Fs = fma(B, gsum(), Fs);
    }
    active = active && (over & col_bits) == 0ull;
  };

  // rule 1 may need the row after the column's last layer for the padded point
  const int klast = SIMPSON ? (kend + 1 < L ? kend + 1 : L - 1) : kend;
  // Loads in flight ahead of the step that uses them: two steps' worth.  (Four, for the eight-row form that serves
  // the smallest launches, was measured in round 4 and changes nothing there -- one walker on the demo grid 25.4
  // against 24.3 us: those launches are bound by the steps' own dependent arithmetic and cross-row traffic, not by
  // memory round trips -- and costs the two-walker launch on the bench grid its occupancy, 42.7 against 34.7 us.)
  constexpr int NBUF = 2;
  double rbuf[NBUF][NR];
#pragma unroll
  for (int b = 0; b < NBUF - 1; b++) load_layer(clampk(R * b + q), rbuf[b]);
  int s = 0;
  bool gone = false;
  for (; R * s <= klast && !gone; s += NBUF) {
#pragma unroll
    for (int b = 0; b < NBUF; b++) {
      if (!gone && R * (s + b) <= klast) {
        load_layer(clampk(R * (s + b + NBUF - 1) + q), rbuf[(b + NBUF - 1) % NBUF]);
        step(s + b, rbuf[b]);
        if (!__any(active || pad_next)) { gone = true; s += b - NBUF; }   // (s: the last step walked, after the loop's increment)
      }
    }
  }
  if (!gone) s -= 1;   // the loop ran out: the last step walked is the one before s (clamped by the record below)
  // the rows of a wavenumber hold its layers' terms: sum them; row 0 writes
  const bool mine_counts = !SIMPSON || ((q & 1) == (nend & 1));
  double F = Fs + (mine_counts ? I : 0.0);   // the lane's layers, angle quadrature already taken
  if constexpr (ALLR) {
    F = Fs;
#pragma unroll
    for (int a = 0; a < A; a++) {
      int np = Npad[a];                    // the ray's padded row, known to the lane that evaluated it
      for (int o = WN; o < 64; o <<= 1) { const int t = __shfl_xor(np, o); np = t > np ? t : np; }
      const int nlast = np >= 0 ? np : kend;   // its last point's index
      F = fma(p.wq[a], ((q & 1) == (nlast & 1)) ? Ia[a] : 0.0, F);
    }
  }
  for (int o = WN; o < 64; o <<= 1) F += __shfl_xor(F, o);
  if constexpr (RAYS) {   // the rays of a wavenumber: columns mw A .. mw A + A - 1 of row 0, in ray order
    double Ft = 0.0;
#pragma unroll
    for (int a = 0; a < A; a++) Ft += __shfl(F, (mw * A + a) & 63);
    F = Ft;
  }
  if (q == 0 && ray == 0 && col_on && i0 + mw < W) p.spec[(size_t)w * W + i0 + mw] = F;
  if (p.walked_out && lane == 0) {
    const int layers = R * (s + 1) < kend + 1 ? R * (s + 1) : kend + 1;
    p.walked_out[(size_t)w * (4 * p.ntiles) + tile * 4 + (threadIdx.x >> 6)] = layers;
  }
}

}  // namespace bartrt
#include "rt_eclipse_s1.hpp"   // rule 1's single-wave kernel
#include "rt_eclipse_s1s.hpp"  // ... with the `toomuch` cut on each ray's slant depth
#include "rt_eclipse_s1t.hpp"  // ... as a team of three waves per column
namespace bartrt {

#ifndef __HIPCC_RTC__   // ---- host side: the launchers (to the end of the file)
}  // namespace bartrt
#include "rtc.hpp"
#include <cstdarg>
#include <cstdio>
namespace bartrt {
// A shape the ahead-of-time set does not hold: the same kernel template, instantiated at run time (rtc.hpp).
// fmt / ...: the template-id in namespace bartrt.  false: no compiler at hand -- the caller falls through to the
// generic kernel as before.
inline bool rtc_try(RtLaunchInfo *info, bool ilp, dim3 grid, dim3 block, size_t sh, hipStream_t st, const RtArgs &b,
                    hipError_t &err, const char *fmt, ...) {
  // (engines without an opacity table -- cross sections only, the line-by-line hand-off -- keep the generic kernel / the
  // EXT builds: the table kernels were never instantiated, let alone tested, for zero molecules)
  if (b.M < 1) return false;
  char ex[192];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(ex, sizeof ex, fmt, ap);
  va_end(ap);
  if (!rtc_launch(ex, ilp, grid, block, sh, st, b, err)) return false;
  if (info) info->rtc = true;
  return true;
}
inline const char *tf(bool b) { return b ? "true" : "false"; }
// The single-wave kernels keep three slots of 2 M + 2 C loads in flight: beyond 20 loads per layer (the widest shapes
// of the ahead-of-time list: eight molecules with two slots, six with four) they spill and the generic kernel is as
// fast or faster at ten walkers and up (round 5, W = 1e4: 7 molecules + 4 slots 507 against 585 us at ten walkers,
// 2 258 / 2 250 at 64; 9 + 2: 448 / 456 and 1 885 / 1 688) -- such shapes are instantiated for the layer-parallel
// kernels only (one walker: 58 against 245 us), their batches stay with the generic kernel.
inline bool rtc_single_wave_ok(const RtArgs &a) { return 2 * a.M + 2 * a.C <= 20; }
// If one ray angle has exactly half the cosine of another (0 and 60 degrees of
// the usual raygrid 0 20 40 60 80), put that pair first and last: the SQ kernels
// take the last transmittance as the square of the first.
inline bool order_angles_for_square(RtArgs &r) {
  for (int i = 0; i < r.A; i++)
    for (int j = 0; j < r.A; j++) {
      if (i == j || std::fabs(r.invmu[j] - 2.0 * r.invmu[i]) > 8.9e-16 * r.invmu[j]) continue;
      auto swap_angles = [&](int x, int y) {
        std::swap(r.invmu[x], r.invmu[y]);
        std::swap(r.wgt[x], r.wgt[y]);
        std::swap(r.wq[x], r.wq[y]);
        std::swap(r.mu[x], r.mu[y]);
        std::swap(r.thr[x], r.thr[y]);
        std::swap(r.drank[x], r.drank[y]);
        std::swap(r.thrb[x], r.thrb[y]);
      };
      swap_angles(0, i);
      if (j == 0) j = i;  // the doubled angle sat in slot 0 and moved to i
      swap_angles(r.A - 1, j);
      return true;
    }
  return false;
}

// Launches the specialised kernel for this shape and batch size under rule INTEG;
// returns false when the shape has none (the caller falls back to the generic
// kernel).  `kmode`: forced variant (BARTRT_KERNEL), empty = by batch size.
// info (optional): what was launched (name, wavenumbers per recorded column).
// the single-wave kernel of rule 0 in its ILP-scheduled build (rt_eclipse_i0_ilp.hip);
// false: no instantiation for this shape
bool launch_rt_fast_ilp(const RtArgs &b, bool sq, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err);
// ... and with the line-by-line extinction array as input (no table, 0-2 CIA pairs)
bool launch_rt_fast_ext(const RtArgs &b, bool sq, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err);

// the `cut slant` kernels of rules 0 / 1 for five angles (rt_eclipse_slant_ilp.hip), table and line-by-line input
bool launch_rt_slant(const RtArgs &b, int integ, bool sq, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err);
bool launch_rt_slant_ext(const RtArgs &b, int integ, bool sq, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err);
bool launch_rt_slant_out(const RtArgs &b, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err);

// the layer-parallel walk of rule 1 / `cut slant` with a column's rows on adjacent lanes (rt_eclipse_qadj.hpp, built in
// rt_eclipse_qadj.hip): five angles, rows = 8 or 16; false: no instantiation for this shape
bool launch_rt_qadj(const RtArgs &b, bool sq, int rows, int nblocks, size_t sh, hipStream_t st, hipError_t &err);

// ... and for the ray-grid sizes other than five built ahead of time (rt_eclipse_angles.hip, one object per size; the
// list comes from bart_amd/build.py -- empty by default since round 6: every other size is instantiated at run time)
#ifndef BARTRT_ANGLE_SIZES
#define BARTRT_ANGLE_SIZES(X)
#endif
#define BARTRT_DECL_ANGLES(N) \
  bool launch_rt_angles_##N(const RtArgs &b, int integ, int block, int nblocks, size_t sh, hipStream_t st, hipError_t &err);
BARTRT_ANGLE_SIZES(BARTRT_DECL_ANGLES)
#undef BARTRT_DECL_ANGLES

// fold (optional): the preparation of this launch's walkers, NOT yet launched -- only the kernels that can run it in
// their own prologue (the all-rays layer-parallel forms of rule 1 under `cut slant`) are considered then, and false
// means "launch the preparation, then call again without it".
template <int INTEG>
bool launch_rt_spec(const RtArgs &a, int block, hipStream_t st, const std::string &kmode, bool force_window,
                    bool allow_sq, hipError_t &err, RtLaunchInfo *info, const PrepArgs *fold = nullptr) {
  if (fold && (INTEG != kIntegSimpson || !a.cut_slant || a.A != 5 || a.ext || a.intens_out || a.tau_out || a.nprep != 0)) return false;
  const size_t sh_fold = fold ? sizeof(double) * prep_lds_doubles(fold->L, fold->S, fold->Nt, fold->ncia_temps) : 0;
  const int ntiles8 = (a.ntiles + 7) / 8 * 8;
  const int nblocks = ntiles8 * a.nwalkers;
  const size_t sh = sizeof(double) * ((size_t)a.L * coef_stride(a.M, a.C) + integ_lds_doubles<INTEG>(a.L)) +
                    sizeof(idx_t) * (size_t)a.L * idx_stride(a.C);
  // (the specialised kernels rebuild their buffer descriptor per layer, so the
  // table may be of any size; one layer's pair of planes must stay below 4 GB)
  const bool plane_ok = 2ull * a.M * a.W * 8ull < (1ull << 31);
  if (a.ext) {
    // line-by-line hand-off: the single-wave kernel with the extinction array as one
    // more load per layer (rules 0 and 1, no table; anything else takes the generic kernel)
    if (!(INTEG != kIntegTrapzTau && a.A == 5 && a.M == 0 && (a.C <= 2 || a.C == 4) && !a.intens_out && !a.tau_out &&
          sh <= 55 * 1024 && kmode != "generic"))
      return false;
    RtArgs b = a;
    const bool sq = allow_sq && order_angles_for_square(b);
    b.ntiles = a.ntiles;
    if (info) { info->kernel = "rt_eclipse_fast (line-by-line extinction)"; info->wn_per_column = block; info->ncolumns = b.ntiles; }
    err = hipSuccess;
    if (a.cut_slant) {
      if (info) info->kernel = "single-wave `cut slant` kernel (line-by-line extinction)";
      return launch_rt_slant_ext(b, INTEG, sq, block, nblocks, sh, st, err);
    }
    if (INTEG == kIntegSimpson ? launch_rt_simpson_ext(b, sq, block, nblocks, sh, st, err)
                               : launch_rt_fast_ext(b, sq, block, nblocks, sh, st, err))
      return true;
    return false;
  }
  if ((a.intens_out || a.tau_out) && a.cut_slant && INTEG == kIntegSimpson && a.A == 5 && a.slog && a.nwalkers == 1 &&
      a.nprep == 0 && plane_ok && sh <= 55 * 1024 && kmode.empty()) {
    // tau.dat / outintens of the default conventions: the single-wave slant kernel writes them on its way
    if (info) { info->kernel = "rt_eclipse_simpson_slant (with optical-depth / intensity outputs)"; info->wn_per_column = block; info->ncolumns = a.ntiles; info->prep_fused = false; }
    err = hipSuccess;
    if (launch_rt_slant_out(a, block, nblocks, sh, st, err)) return true;
    if (rtc_single_wave_ok(a) &&
        rtc_try(info, false, dim3(nblocks), dim3(block), sh, st, a, err, "rt_eclipse_simpson_slant<5, %d, %d, false, 0, false, true>", a.M, a.C))
      return true;
  }
  if (!(!a.intens_out && !a.tau_out && plane_ok && sh <= 55 * 1024)) return false;
  // (the event log serves rule 1's single-wave kernels only -- rt_eclipse_s1s.hpp, rt_eclipse_s1t.hpp; rule 2 on other
  // ray grids: generic kernel)
  if (a.cut_slant && ((INTEG == kIntegSimpson && !a.slog) || (INTEG == kIntegTrapzTau && a.A != 5))) return false;
  if (a.A != 5) {
    // other ray-grid sizes: the single-wave kernel of rule 0 / rule 1 at every batch size
    if (INTEG == kIntegTrapzTau || a.A < 1 || a.A > kMaxAngles || kmode == "quad" || kmode == "octo" || kmode == "split") return false;
    size_t sha = sh;
    int nba = nblocks;
    if (a.nprep > 0) {
      sha = std::max(sh, sizeof(double) * prep_lds_doubles(a.prep_next.L, a.prep_next.S, a.prep_next.Nt, a.prep_next.ncia_temps));
      nba += prep_slots(a.nprep);
    }
    if (info) {
      info->kernel = a.cut_slant ? (INTEG == kIntegSimpson ? "rt_eclipse_simpson_slant (ray grid of another size)"
                                                          : "rt_eclipse_fast<SLANT> (ray grid of another size)")
                     : INTEG == kIntegSimpson ? "rt_eclipse_simpson (ray grid of another size)" : "rt_eclipse_fast (ray grid of another size)";
      info->wn_per_column = block; info->ncolumns = a.ntiles; info->prep_fused = a.nprep > 0;
    }
    err = hipSuccess;
    bool done = false;
    switch (a.A) {
#define BARTRT_CASE_ANGLES(N) case N: done = launch_rt_angles_##N(a, INTEG, block, nba, sha, st, err); break;
      BARTRT_ANGLE_SIZES(BARTRT_CASE_ANGLES)
#undef BARTRT_CASE_ANGLES
      default: break;
    }
    if (!done && rtc_single_wave_ok(a)) {
      // (a ray grid of ten and more angles, or a (molecules, slots) pair outside the ahead-of-time list)
      const dim3 g(nba), bl(block);
      if (a.cut_slant)
        done = INTEG == kIntegTransmittance
                   ? rtc_try(info, true, g, bl, sha, st, a, err, "rt_eclipse_fast<%d, %d, %d, false, 0, 1, false, true>", a.A, a.M, a.C)
                   : rtc_try(info, true, g, bl, sha, st, a, err, "rt_eclipse_simpson_slant<%d, %d, %d, false, %d>", a.A, a.M, a.C, a.A <= 6 ? 1 : 0);
      else
        done = INTEG == kIntegTransmittance
                   ? rtc_try(info, true, g, bl, sha, st, a, err, "rt_eclipse_fast<%d, %d, %d, false, 0, 1>", a.A, a.M, a.C)
                   : rtc_try(info, true, g, bl, sha, st, a, err, "rt_eclipse_simpson<%d, %d, %d, false, 1>", a.A, a.M, a.C);
    }
    if (!done && info) info->prep_fused = false;
    return done;
  }
  // (the producer/consumer kernel adds 9 kB of its own)
  RtArgs b = a;
  const bool sq = allow_sq && order_angles_for_square(b);
  // every specialised kernel carries the next batch's preparation (RtArgs::nprep) at the head of
  // its grid: prep_slots(nprep) more workgroups, LDS for the larger of the two jobs
  const int pslots = a.nprep > 0 ? prep_slots(a.nprep) : 0;
  size_t shp = 0;   // LDS the preparation needs beyond the RT workgroups'
  if (a.nprep > 0) {
    const size_t need = sizeof(double) * prep_lds_doubles(a.prep_next.L, a.prep_next.S, a.prep_next.Nt, a.prep_next.ncia_temps);
    shp = need > sh ? need - sh : 0;
    if (info) info->prep_fused = true;
  }
  // too few single-wave columns to load the 1 024 SIMDs evenly -> several
  // waves per 64 wavenumbers: four 16-wavenumber waves that take four layers at
  // a time (quad-layer), or a producer / consumer pair
  const int nsel = a.nsel > a.nwalkers ? a.nsel : a.nwalkers;   // (the batch the variant is chosen for: RtArgs::nsel)
  const long columns = (long)nsel * (((a.Wfull > 0 ? a.Wfull : a.W) + 63) / 64);   // (of the whole grid: RtArgs::Wfull)
  const int ntiles64 = (a.W + 63) / 64;
  const int nb64 = (ntiles64 + 7) / 8 * 8 * a.nwalkers;
  // the quad-layer kernel addresses the tables with per-lane 32-bit offsets
  // (a grid of 4 GB or more through a window that moves with the step's layers)
  const bool octo = kmode == "octo" || (kmode.empty() && columns <= kOctoMaxColumns);
  b.window = a.kappa_bytes >= (1ull << 32) - 4096 || force_window;
  const bool fits32 = a.cia_bytes < (1ull << 32) - 4096 && (!b.window || window_fits(a, octo ? 8 : 4));
  err = hipSuccess;
  constexpr bool SQOK = true;   // exp(-2 tau / mu) = exp(-tau / mu)^2 under every rule
  constexpr long quad_max = INTEG == kIntegSimpson ? kQuadMaxColumnsSimpson : kQuadMaxColumns;
  if (a.cut_slant) {
    // the per-ray cut: launches that leave the chip mostly idle take the layer-parallel walk with ONE RAY PER LANE
    // (rt_eclipse_quad<..., RAYS>: three wavenumbers x five rays x four layers per wave and step) ...
    // which of the two layer-parallel forms: one ray per lane for the smallest launches and for rules 0 / 2, all rays per
    // lane (rule 1) above that (BARTRT_KERNEL: quad / octo = the rule's own choice of form, quadrays / octorays = one ray
    // per lane)
    const bool rays_forced = kmode == "quadrays" || kmode == "octorays";
    const bool lp_forced = kmode == "quad" || kmode == "octo";
    const bool use_rays = rays_forced || (INTEG != kIntegSimpson && (lp_forced || (kmode.empty() && columns <= kQuadRaysMaxColumns)));
    if constexpr (INTEG == kIntegSimpson) {
      // rule 1: the layer-parallel walk with ALL rays per lane (rt_eclipse_quad<..., ALLR>)
      // rows = layers per step (and 64 / rows wavenumbers per wave): the fewer the columns, the more rows
      static const int rows_env = [] { const char *v = std::getenv("BARTRT_ALLR_ROWS"); return v && *v ? atoi(v) : 0; }();
      // the launch's variant: forced (BARTRT_KERNEL, BARTRT_ALLR_ROWS), else the measured table's (kernel_table.inc)
      static const int adj_env = [] { const char *v = std::getenv("BARTRT_ADJ"); return v && *v ? atoi(v) : -1; }();
      const KernelChoice &entry = slant_simpson_choice(a.M, columns);
      const bool adj_ok = adj_env != 0 && rows_env == 0;
      const bool adj_named = entry.variant == kVarAdj16 || entry.variant == kVarAdj8;
      // (the layer-parallel form that runs should the adjacent one not launch: the entry's fallback)
      const int other = adj_named ? entry.fallback : entry.variant;
      int rows = kmode == "quad" ? 4 : kmode == "octo" ? 8 : kmode == "hexa" ? 16 : kmode == "r32" ? 32 : other == kVarSingle ? 4 : other;
      if (rows_env == 4 || rows_env == 8 || rows_env == 16 || rows_env == 32) rows = rows_env;
      const bool lpa_forced = lp_forced || kmode == "hexa" || kmode == "r32" || kmode == "adj8" || kmode == "adj16";
      while (rows > 4 && b.window && !window_fits(a, rows)) rows /= 2;
      // rt_eclipse_qadj (rows on adjacent lanes: DPP row shifts instead of ds_bpermute, carries in place) where the table
      // names it; BARTRT_KERNEL=adj8 / adj16 force it, BARTRT_ADJ=0 switches it off
      int adj_rows = kmode == "adj8" ? 8 : kmode == "adj16" ? 16 : 0;
      if (kmode.empty() && adj_ok && adj_named) adj_rows = entry.variant == kVarAdj16 ? 16 : 8;
      // (the layer-parallel walk at all: forced, or the table names one of its forms; BARTRT_ALLR_ROWS, the A/B tools'
      // switch, keeps round 5's range for it)
      const bool lp_by_table = kmode.empty() && (rows_env != 0 ? columns <= kQuadAllMaxColumns : other != kVarSingle);
      if (adj_rows && !use_rays && a.cia_bytes < (1ull << 32) - 4096 && (!b.window || window_fits(a, adj_rows))) {
        const int awn = 64 / adj_rows;
        RtArgs ba = b;
        ba.ntiles = (a.W + 4 * awn - 1) / (4 * awn);
        const int nba = (ba.ntiles + 7) / 8 * 8 * a.nwalkers + pslots;
        size_t sha = sh + shp + (adj_rows >= 16 ? sizeof(double) * (size_t)a.L : 0);
        const bool folds = fold && sha + sh_fold <= 64 * 1024 && (ba.ntiles + 7) / 8 * 8 * nsel + pslots <= kFoldMaxWorkgroups;
        if (folds) { ba.nprep = -1; ba.prep_next = *fold; sha += sh_fold; }
        if (fold && !folds) return false;
        RtLaunchInfo keep;
        if (info) {
          keep = *info;
          info->kernel = adj_rows == 16 ? "rt_eclipse_qadj<R=16> (rows on adjacent lanes)" : "rt_eclipse_qadj<R=8> (rows on adjacent lanes)";
          info->wn_per_column = awn; info->ncolumns = 4 * ba.ntiles;
        }
        err = hipSuccess;
        if (info) info->prep_folded = folds;
        if (launch_rt_qadj(ba, sq, adj_rows, nba, sha, st, err)) return true;
        if (rtc_try(info, false, dim3(nba), dim3(256), sha, st, ba, err, "rt_eclipse_qadj<5, %d, %d, %s, %d>", a.M, a.C, tf(sq), adj_rows)) return true;
        if (info) *info = keep;    // (neither an instantiation nor a compiler: the choice before it)
      }
      if (!use_rays && (lpa_forced || lp_by_table) && a.cia_bytes < (1ull << 32) - 4096 &&
          (!b.window || window_fits(a, rows))) {
        const int wnw = 64 / rows;   // wavenumbers per wave
        b.ntiles = (a.W + 4 * wnw - 1) / (4 * wnw);
        const int nbq = (b.ntiles + 7) / 8 * 8 * a.nwalkers + pslots;
        size_t shq = sh + shp + (rows >= 16 ? sizeof(double) * (size_t)a.L : 0);   // (padded records: NCS)
        const bool folds = fold && shq + sh_fold <= 64 * 1024 && (b.ntiles + 7) / 8 * 8 * nsel + pslots <= kFoldMaxWorkgroups;
        if (fold && !folds) return false;
        if (folds) { b.nprep = -1; b.prep_next = *fold; shq += sh_fold; }
        if (info) info->prep_folded = folds;
        if (info) {
          info->kernel = rows == 32 ? "rt_eclipse_quad<R=32, all rays per lane>" : rows == 16 ? "rt_eclipse_quad<R=16, all rays per lane>"
                         : rows == 8 ? "rt_eclipse_quad<R=8, all rays per lane>" : "rt_eclipse_quad<R=4, all rays per lane>";
          info->wn_per_column = wnw; info->ncolumns = 4 * b.ntiles;
        }
#define BARTRT_QUADALL_R(MM, CC, RR)                                                                                         \
      if (sq) BARTRT_RT_LAUNCH((rt_eclipse_quad<5, MM, CC, true, RR, INTEG, false, true>), dim3(nbq), dim3(256), shq, st, b); \
      else BARTRT_RT_LAUNCH((rt_eclipse_quad<5, MM, CC, false, RR, INTEG, false, true>), dim3(nbq), dim3(256), shq, st, b);
#define BARTRT_QUADALL(MM, CC)                                                                                               \
  if (a.M == MM && a.C == CC) {                                                                                              \
    if (rows == 32) { BARTRT_QUADALL_R(MM, CC, 32) }                                                                         \
    else if (rows == 16) { BARTRT_QUADALL_R(MM, CC, 16) }                                                                    \
    else if (rows == 8) { BARTRT_QUADALL_R(MM, CC, 8) }                                                                      \
    else { BARTRT_QUADALL_R(MM, CC, 4) }                                                                                     \
    err = hipGetLastError();                                                                                                 \
    return true;                                                                                                             \
  }
        BARTRT_MC_LIST(BARTRT_QUADALL)
#undef BARTRT_QUADALL
#undef BARTRT_QUADALL_R
        if (rtc_try(info, false, dim3(nbq), dim3(256), shq, st, b, err, "rt_eclipse_quad<5, %d, %d, %s, %d, %d, false, true>", a.M, a.C,
                    tf(sq), rows, INTEG))
          return true;
        if (folds) { b.nprep = 0; if (info) info->prep_folded = false; }
      }
      if (fold) return false;   // (no kernel that prepares its own walkers serves this launch: the caller launches prep_profiles)
    }
    if (use_rays && fits32) {
      // (the smallest launches -- one walker on the demo shape -- eight layers per step: one wavenumber x five rays per wave)
      const bool octor = kmode == "octo" || kmode == "octorays" || (kmode.empty() && columns <= kOctoRaysMaxColumns);
      const bool win_ok = !b.window || window_fits(a, octor ? 8 : 4);
      if (win_ok) {
        b.ntiles = octor ? (a.W + 3) / 4 : (a.W + 11) / 12;          // a workgroup: four waves of one / three wavenumbers
        const int nbq = (b.ntiles + 7) / 8 * 8 * a.nwalkers + pslots;
        const size_t shq = sh + shp;
        if (info) {
          info->kernel = octor ? "rt_eclipse_quad<R=8, one ray per lane>" : "rt_eclipse_quad<R=4, one ray per lane>";
          info->wn_per_column = octor ? 1 : 3; info->ncolumns = 4 * b.ntiles;
        }
#define BARTRT_QUADRAYS(MM, CC)                                                                                          \
  if (a.M == MM && a.C == CC) {                                                                                          \
    if (octor) BARTRT_RT_LAUNCH((rt_eclipse_quad<5, MM, CC, false, 8, INTEG, true>), dim3(nbq), dim3(256), shq, st, b);  \
    else BARTRT_RT_LAUNCH((rt_eclipse_quad<5, MM, CC, false, 4, INTEG, true>), dim3(nbq), dim3(256), shq, st, b);        \
    err = hipGetLastError();                                                                                             \
    return true;                                                                                                         \
  }
        BARTRT_MC_LIST(BARTRT_QUADRAYS)
#undef BARTRT_QUADRAYS
        if (rtc_try(info, false, dim3(nbq), dim3(256), shq, st, b, err, "rt_eclipse_quad<5, %d, %d, false, %d, %d, true>", a.M, a.C,
                    octor ? 8 : 4, INTEG))
          return true;
      }
    }
    // ... a team of three waves per column (rule 1) while single-wave columns would load the SIMDs unevenly ...
    if (INTEG == kIntegSimpson && (kmode == "team" || (kmode.empty() && columns <= kTeamMaxColumns))) {
      b.ntiles = ntiles64;
      const size_t sht = sh + sizeof(double) * team_lds_doubles() + shp;
      if (info) { info->kernel = "rt_eclipse_slant_team (three waves per column)"; info->wn_per_column = 64; info->ncolumns = b.ntiles; }
      if (launch_rt_slant_team(b, sq, nb64 + pslots, sht, st, err)) return true;
    }
    // ... everything else the single-wave kernels (each ray its own sums in one lane)
    b.ntiles = a.ntiles;
    if (info) {
      info->kernel = INTEG == kIntegSimpson ? "rt_eclipse_simpson_slant (ILP-scheduled build)" : "rt_eclipse_fast<SLANT> (ILP-scheduled build)";
      info->wn_per_column = block; info->ncolumns = b.ntiles;
    }
    if (launch_rt_slant(b, INTEG, sq, block, nblocks + pslots, sh + shp, st, err)) return true;
    if (!rtc_single_wave_ok(a)) { if (info) info->prep_fused = false; return false; }
    if (INTEG == kIntegSimpson
            ? rtc_try(info, true, dim3(nblocks + pslots), dim3(block), sh + shp, st, b, err, "rt_eclipse_simpson_slant<5, %d, %d, %s, %d>", a.M, a.C, tf(sq),
                      slant_sched(a.M, a.C, 1))
            : rtc_try(info, true, dim3(nblocks + pslots), dim3(block), sh + shp, st, b, err, "rt_eclipse_fast<5, %d, %d, %s, %d, 1, false, true>", a.M,
                      a.C, tf(sq), INTEG))
      return true;
    if (info) info->prep_fused = false;
    return false;
  }
  if ((kmode == "quad" || kmode == "octo" || (kmode.empty() && columns <= quad_max)) && fits32) {
    // the smallest launches take eight layers per step (8 wavenumbers per wave)
    b.ntiles = octo ? (a.W + 31) / 32 : ntiles64;
    const int nbq = (b.ntiles + 7) / 8 * 8 * a.nwalkers + pslots;
    const size_t shq = sh + shp;
    if (info) { info->kernel = octo ? "rt_eclipse_quad<R=8>" : "rt_eclipse_quad<R=4>"; info->wn_per_column = octo ? 8 : 16; info->ncolumns = 4 * b.ntiles; }
#define BARTRT_QUAD(MM, CC)                                                                                          \
  if (a.M == MM && a.C == CC) {                                                                                      \
    if (octo) {                                                                                                      \
      if (sq) BARTRT_RT_LAUNCH((rt_eclipse_quad<5, MM, CC, SQOK, 8, INTEG>), dim3(nbq), dim3(256), shq, st, b);     \
      else BARTRT_RT_LAUNCH((rt_eclipse_quad<5, MM, CC, false, 8, INTEG>), dim3(nbq), dim3(256), shq, st, b);       \
    } else {                                                                                                         \
      if (sq) BARTRT_RT_LAUNCH((rt_eclipse_quad<5, MM, CC, SQOK, 4, INTEG>), dim3(nbq), dim3(256), shq, st, b);     \
      else BARTRT_RT_LAUNCH((rt_eclipse_quad<5, MM, CC, false, 4, INTEG>), dim3(nbq), dim3(256), shq, st, b);       \
    }                                                                                                                \
    err = hipGetLastError();                                                                                         \
    return true;                                                                                                     \
  }
    BARTRT_MC_LIST(BARTRT_QUAD)
#undef BARTRT_QUAD
    if (rtc_try(info, false, dim3(nbq), dim3(256), shq, st, b, err, "rt_eclipse_quad<5, %d, %d, %s, %d, %d>", a.M, a.C, tf(sq), octo ? 8 : 4, INTEG))
      return true;
  }
  if (kmode == "split" ||
      (kmode.empty() && INTEG != kIntegSimpson && columns >= kSplitMinColumns && columns <= kSplitMaxColumns)) {
    b.ntiles = ntiles64;
    const size_t shs = sh + sizeof(double) * (1024 + 2 + 64) + shp;
    const int nbs = nb64 + pslots;
    if (info) { info->kernel = "rt_eclipse_split"; info->wn_per_column = 64; info->ncolumns = b.ntiles; }
#define BARTRT_SPLIT(MM, CC)                                                                                       \
  if (a.M == MM && a.C == CC) {                                                                                    \
    if (sq) BARTRT_RT_LAUNCH((rt_eclipse_split<5, MM, CC, SQOK, INTEG>), dim3(nbs), dim3(128), shs, st, b);     \
    else BARTRT_RT_LAUNCH((rt_eclipse_split<5, MM, CC, false, INTEG>), dim3(nbs), dim3(128), shs, st, b);       \
    err = hipGetLastError();                                                                                       \
    return true;                                                                                                   \
  }
    BARTRT_MC_LIST(BARTRT_SPLIT)
#undef BARTRT_SPLIT
    if (rtc_try(info, false, dim3(nbs), dim3(128), shs, st, b, err, "rt_eclipse_split<5, %d, %d, %s, %d>", a.M, a.C, tf(sq), INTEG)) return true;
  }
  b.ntiles = a.ntiles;
  if (info) { info->kernel = "rt_eclipse_fast"; info->wn_per_column = block; info->ncolumns = b.ntiles; }
  const size_t sh1 = sh + shp;
  const int nblocks1 = nblocks + pslots;
  [[maybe_unused]] const bool ilp = kmode != "mono_occ" && (kmode == "mono_ilp" || columns < kIlpMaxColumns);
  if constexpr (INTEG == kIntegSimpson) {
    // rule 1 has its own single-wave kernel (rt_eclipse_s1.hpp), built under the ILP schedule
    // only: that build is the faster one at every batch size (10 walkers 73 against 80 us,
    // 64: 333 / 375, 256: 1 138 / 1 220 -- the default schedule needs 182 registers for two
    // resident waves, or drops the record read-ahead for three and waits on LDS instead)
    if (info) info->kernel = "rt_eclipse_simpson (ILP-scheduled build)";
    if (launch_rt_simpson_ilp(b, sq, block, nblocks1, sh1, st, err)) return true;
    if (rtc_single_wave_ok(a) && rtc_try(info, true, dim3(nblocks1), dim3(block), sh1, st, b, err, "rt_eclipse_simpson<5, %d, %d, %s, 1>", a.M, a.C, tf(sq))) return true;
  } else {
    if (INTEG == kIntegTransmittance && ilp) {
      if (info) info->kernel = "rt_eclipse_fast (ILP-scheduled build)";
      if (launch_rt_fast_ilp(b, sq, block, nblocks1, sh1, st, err)) return true;
    }
#define BARTRT_FAST(MM, CC)                                                                                        \
  if (a.M == MM && a.C == CC) {                                                                                    \
    if (sq) BARTRT_RT_LAUNCH((rt_eclipse_fast<5, MM, CC, SQOK, INTEG>), dim3(nblocks1), dim3(block), sh1, st, b); \
    else BARTRT_RT_LAUNCH((rt_eclipse_fast<5, MM, CC, false, INTEG>), dim3(nblocks1), dim3(block), sh1, st, b);   \
    err = hipGetLastError();                                                                                       \
    return true;                                                                                                   \
  }
    BARTRT_MC_LIST(BARTRT_FAST)
#undef BARTRT_FAST
    if (rtc_single_wave_ok(a) && rtc_try(info, false, dim3(nblocks1), dim3(block), sh1, st, b, err, "rt_eclipse_fast<5, %d, %d, %s, %d>", a.M, a.C, tf(sq), INTEG)) return true;
  }
  return false;
}

#endif  // !__HIPCC_RTC__
}  // namespace bartrt
