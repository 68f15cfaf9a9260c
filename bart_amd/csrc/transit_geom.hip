// Transit (transmission) geometry: modulation spectrum.
//
// One lane per (walker, wavenumber), one 64-lane wave per workgroup.  Layers
// are visited from the top; for the chord with impact parameter b = r_k
//     tau_k = sum_{j=1..k} (e_{j-1} + e_j) * ds[k][j],   ds = s_{j-1} - s_j,
//     s_j = sqrt(r_j^2 - r_k^2)
// (trapezoid in the path coordinate s; the factor 2 for the two halves of the
// chord cancels the trapezoid's 1/2).  The pair sums e_{j-1}+e_j of the lane's
// own column live in LDS ([layer][lane], conflict-free); ds depends only on the
// walker's radii, is built by prep_profiles and is read through wave-uniform
// (scalar) loads.  The chord loop stops at the first tau_k > toomuch: deeper
// rays are opaque and their layers are never read.
//     M = (r_top^2 - 2 int exp(-tau(b)) b db) / R_star^2      (trapezoid in b)
//
// rt_transit is the generic form (any layer count, runtime molecule / CIA
// counts, tau output).  rt_transit_mfma is the one the
// batched path runs.  Per wavenumber the chord depths are a lower-triangular
// matrix-vector product tau = DS P (L^2/2 multiply-adds, P_j = e_{j-1} + e_j), so
// for 16 wavenumbers at a time it is a [16 x L] x [L x 16-chord tile] matrix
// product: v_mfma_f64_16x16x4 with M = wavenumber, N = chord, K = layer.  A wave
// takes 16 wavenumbers; lane (q = l/16, m = l%16) computes the extinction of
// wavenumber m in the layers j = 4 s + q only, which is exactly the A operand
// of step s, so the pair sums never leave the lane's registers; the B operands
// are 512-byte coalesced loads of the chord table prep_profiles wrote in that
// order.  The result tile leaves chord 16 kt + m of wavenumbers q, q+4, q+8, q+12
// in lane (q, m): transmission and the trapezoid in b are then per lane, the
// `toomuch` cut is a ballot over the 16 lanes of a row, and the sum over chords
// a 16-lane reduction at the very end.
#include "kernels.hpp"

#include <cstdlib>
#include <string>

namespace bartrt {

// STAGE: the walker's layer records are copied to LDS in front of the pair sums;
// without it (deep columns with many molecules, where both do not fit) they are
// read from global memory where they lie.
template <bool STAGE>
__global__ __launch_bounds__(64) void rt_transit(RtArgs p) {
  extern __shared__ double smem[];
  const int M = p.M, C = p.C, L = p.L, W = p.W;
  const int NC = coef_stride(M, C), NI = idx_stride(C);
  const int b = blockIdx.x;
  const int xcd = b & 7, jb = b >> 3;
  const int w = jb % p.nwalkers;
  const int tile = (jb / p.nwalkers) * 8 + xcd;
  if (tile >= p.ntiles) return;

  const double *gC = p.coef + (size_t)w * L * NC;
  const idx_t *gI = p.idx + (size_t)w * L * NI;
  const size_t nrec = STAGE ? (size_t)L * NC + (size_t)L * NI : 0;
  double *sP = smem + nrec;  // pair sums, [L][64], after the records
  const double *sC = gC;
  const idx_t *sI = gI;
  if constexpr (STAGE) {
    double *lC = smem;
    idx_t *lI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
    stage2_to_lds(lC, gC, L * NC, lI, gI, L * NI, threadIdx.x, 64);
    __syncthreads();
    sC = lC;
    sI = lI;
  }

  const int i = tile * 64 + threadIdx.x;
  const bool valid = i < W;
  const int ii = valid ? i : W - 1;
  const double nu = p.wn[ii];
  const double nu4 = (nu * nu) * (nu * nu);
  const size_t MW = (size_t)M * W;
  const double *rt = p.rtop + (size_t)w * L;
  const double *dsw = p.ds + (size_t)w * chord_table_size(L);

  const int kend = kstop_layer(p.kstop[w]);
  double eprev = 0.0, tau = 0.0, integ = 0.0, gprev = rt[0];
  bool active = true;
  int last = 0;
  for (int k = 0; k <= kend; ++k) {
    const double *c = sC + k * NC;
    const idx_t *ix = sI + k * NI;
    const int l = L - 1 - k;
    double e = c[2 + 2 * M + 2 * C] * nu4 + c[3 + 2 * M + 2 * C];   // Rayleigh + grey cloud
    if (p.ext) e += p.ext[((size_t)w * L + l) * W + ii];
    // grid [plane][W][M], CIA [pair plane][W][2] (kernels.hpp, "Table layout")
    const double *kb = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p.kappa) + ix[0]) + (size_t)ii * M;
    for (int m = 0; m < M; m++)
      e += c[2 + 2 * m] * kb[m] + c[3 + 2 * m] * kb[MW + m];
    for (int cc = 0; cc < C; cc++) {
      const double *ab = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p.cia) + ix[1 + cc]) + 2 * (size_t)ii;
      e += c[2 + 2 * M + 2 * cc] * ab[0] + c[3 + 2 * M + 2 * cc] * ab[1];
    }
    if (k > 0) {
      sP[(size_t)k * 64 + threadIdx.x] = eprev + e;
      const double *dk = dsw + chord_table_index(L, k, 0);   // + (j / 4) * 64 + (j % 4) * 16 per layer j
      double t = 0.0;
      for (int j = 1; j <= k; j++) t = fma(sP[(size_t)j * 64 + threadIdx.x], dk[(j >> 2) * 64 + (j & 3) * 16], t);
      if (active) {
        tau = t;
        const double g = exp(-t) * rt[k];
        integ += 0.5 * (gprev + g) * (rt[k - 1] - rt[k]);
        gprev = g;
      }
    }
    eprev = e;
    if (p.tau_out && valid) p.tau_out[(size_t)i * L + k] = tau;
    if (active) {
      last = k;
      if (tau > p.toomuch) active = false;
    }
    if (!__any(active)) break;
  }
  if (valid) {
    // below the last chord the planet is an opaque disc -- unless the cfg says
    // `transparent`: then those rays keep the last chord's transmission
    const double core = p.transparent ? exp(-tau) * rt[last] * rt[last] : 0.0;
    p.spec[(size_t)w * W + i] = (rt[0] * rt[0] - 2.0 * integ - core) * p.inv_starrad2;
    if (p.tau_out) {
      for (int k = last + 1; k < L; k++) p.tau_out[(size_t)i * L + k] = tau;
      p.last_out[i] = last;
    }
  }
}

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kMfmaTiles = 8;      // row tiles of 16 chords: L <= 128
constexpr int kMfmaTilesDeep = 16;  // L <= 256: twice the pair-sum registers, half the waves per SIMD
constexpr int kMfmaTilesMax = 20;   // L <= 320 (the most layers the transit geometry takes at all): one wave per SIMD,
                                    // the pair sums spill into AGPRs -- still the matrix tiles, not the scalar kernel

// EXT: the line-by-line hand-off -- the layer's line extinction ext[w][l][W] (atm layer order) is
// one more 8-byte load per (layer, wavenumber) and one more addend (such engines have no table).
// (Registers: the 100-layer build took 180 + 8 and ran two waves per SIMD; the launch is a chain of dependent trips to
// memory per wave -- a tile's rows, its matrix products, the next tile -- so the waves a SIMD holds are what hides
// them.  Held to 168 registers it keeps three, without a spill; deeper columns keep their registers.)
#ifndef BARTRT_TRANSIT_WPE
#define BARTRT_TRANSIT_WPE 3
#endif
template <int MT, int CT, int KT, bool EXT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KT <= 8 ? BARTRT_TRANSIT_WPE : 1)))
void rt_transit_mfma(RtArgs p) {
  extern __shared__ double smem[];
  constexpr int M = MT, C = CT;
  constexpr int NC = 4 + 2 * M + 2 * C, NI = 1 + C, NLD = 2 * M + 2 * C, NR = NLD > 0 ? NLD : 1;
  const int L = p.L, W = p.W;
  const int b = blockIdx.x;
  const int xcd = b & 7, jb = b >> 3;
  const int w = jb % p.nwalkers;
  const int tile = (jb / p.nwalkers) * 8 + xcd;  // 64 wavenumbers per workgroup
  if (tile >= p.ntiles) return;

  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  double *sRt = smem + (size_t)L * NC + (size_t)L * NI;  // radii top -> bottom, [L]
  stage2_to_lds(sC, p.coef + (size_t)w * L * NC, L * NC, sI, p.idx + (size_t)w * L * NI, L * NI,
                threadIdx.x, 256);
  for (int l = threadIdx.x; l < L; l += 256) sRt[l] = p.rtop[(size_t)w * L + l];   // (L can exceed the 256 lanes)
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, m = lane & 15;
  const int i0 = tile * 64 + (threadIdx.x >> 6) * 16;  // this wave's first wavenumber
  if (i0 >= W) return;
  const unsigned ii = i0 + m < W ? (unsigned)(i0 + m) : (unsigned)(W - 1);
  const double nu = p.wn[ii];
  const double nu4 = (nu * nu) * (nu * nu);
  const int kend = kstop_layer(p.kstop[w]);
  const int nkt = (L + 15) / 16, ns = 4 * nkt;
  const double *__restrict__ dsm = p.ds + (size_t)w * chord_table_size(L);

  // table loads: per-lane byte offset = plane offset of the lane's layer + the lane's
  // wavenumber inside the plane (kernels.hpp, "Table layout")
  const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.cia), 0, (int)p.cia_bytes, 0x00020000);
  const unsigned vk = ii * 8u * (unsigned)M, vc = ii * 16u, planeB = (unsigned)M * (unsigned)W * 8u;
  auto load_layer = [&](int j, double (&r)[NR]) {
    const idx_t *ix = sI + j * NI;
    if (M > 0) {
      const idx_t mine = ix[0];
      const long long base = p.window ? row_window_base<4>(mine) : 0ll;
      const unsigned long long left = p.kappa_bytes - (unsigned long long)base;
      const auto rs_k = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.kappa) + base), 0,
          (int)(unsigned)(left < 0xffffffffull ? left : 0xffffffffull), 0x00020000);
      load_table_lane<M>(rs_k, (unsigned)(mine - base) + vk, 0, planeB, r);
    }
#pragma unroll
    for (int cc = 0; cc < C; cc++) load_cia_lane(rs_c, (unsigned)ix[1 + cc] + vc, 0, r + 2 * M + 2 * cc);
  };

  double P[4 * KT];  // pair sums e_{j-1} + e_j of this lane's layers j = 4 s + q
  double ecarry = 0.0;       // lanes q = 0: extinction of layer j - 1, from row q = 3 one step back
  double integ[4] = {0.0, 0.0, 0.0, 0.0};
  double gcarry[4];          // exp(-tau) r of the last chord of the previous tile
  bool active[4] = {true, true, true, true};
  // `transparent` (no opaque core): exp(-tau) r^2 of the deepest chord counted, per lane; the
  // deepest among a wavenumber's lanes is picked at the end
  int klast[4] = {-1, -1, -1, -1};
  double vlast[4] = {0.0, 0.0, 0.0, 0.0};
  const double r_top = sRt[0];
#pragma unroll
  for (int r = 0; r < 4; r++) gcarry[r] = r_top;

#pragma unroll
  for (int kt = 0; kt < KT; kt++) {
    const int k0 = 16 * kt;
    if (k0 > kend) break;
    // ---- extinction of the tile's 16 layers: 4 per lane row
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int s = 4 * kt + t, j = 4 * s + q;
      const int jc = j < kend ? j : kend;
      double rv[NR];
      load_layer(jc, rv);
      const double *c = sC + jc * NC;
      double e = fma(c[2 + 2 * M + 2 * C], nu4, c[3 + 2 * M + 2 * C]);   // Rayleigh + grey cloud
#pragma unroll
      for (int x = 0; x < NLD; x++) e = fma(c[2 + x], rv[x], e);
      if constexpr (EXT) e += p.ext[((size_t)w * L + (size_t)(L - 1 - jc)) * W + ii];
      const double below = __shfl(e, (lane + 48) & 63);  // row q - 1, i.e. layer j - 1 (q >= 1)
      const double eprev = q == 0 ? ecarry : below;
      ecarry = below;
      P[s] = (j >= 1 && j <= kend) ? eprev + e : 0.0;
    }
    // ---- tau of chords k0 .. k0 + 15 for the wave's 16 wavenumbers
    // (one accumulation chain; four interleaved ones and an explicit one-step
    // prefetch of the table values both measured slower: the loop is fully unrolled
    // and the compiler already hoists the loads)
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    const double *__restrict__ bt = dsm + (size_t)kt * ns * 64 + lane;
#pragma unroll
    for (int s = 0; s < 4 * kt + 4; s++)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(P[s], bt[(size_t)s * 64], acc, 0, 0, 0);
    // ---- lane (q, m): chord k of wavenumbers q + 4 r
    const int k = k0 + m;
    const bool kvalid = k >= 1 && k <= kend;
    const double rk = k < L ? sRt[k] : 0.0;
    const double dr = kvalid ? sRt[k - 1] - rk : 0.0;
    bool any_active = false;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const double tau = acc[r];
      const double g = exp_rt(fmax(-tau, kExpMin)) * rk;
      double gprev = __shfl_up(g, 1, 16);
      if (m == 0) gprev = gcarry[r];
      const unsigned long long over = __ballot(kvalid && tau > p.toomuch);
      const unsigned rowbits = (unsigned)(over >> (16 * q)) & 0xffffu;
      const bool counts = kvalid && active[r] && (rowbits & ((1u << m) - 1u)) == 0u;
      integ[r] += counts ? 0.5 * (gprev + g) * dr : 0.0;
      if (counts) { klast[r] = k; vlast[r] = g * rk; }
      active[r] = active[r] && rowbits == 0u;
      gcarry[r] = __shfl(g, 16 * q + 15);
      any_active = any_active || active[r];
    }
    if (!__any(any_active)) break;
  }
  // sum over the 16 chords a row holds; lane m = 0 of row q writes wavenumbers q + 4 r
#pragma unroll
  for (int r = 0; r < 4; r++) {
    double s = integ[r];
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
    double core = 0.0;   // transparent: the rays below the last chord keep its transmission
    if (p.transparent) {
      int kl = klast[r];
      double vl = vlast[r];
      for (int o = 8; o > 0; o >>= 1) {
        const int ko = __shfl_xor(kl, o, 16);
        const double vo = __shfl_xor(vl, o, 16);
        if (ko > kl) { kl = ko; vl = vo; }
      }
      core = kl >= 0 ? vl : r_top * r_top;   // no chord below the top: tau = 0 there
    }
    const int iw = i0 + q + 4 * r;
    if (m == 0 && iw < W) p.spec[(size_t)w * W + iw] = (r_top * r_top - 2.0 * s - core) * p.inv_starrad2;
  }
}

// Chord table from the radii prep_profiles wrote: one lane per entry the RT
// kernels read (row tile kt uses steps s < 4 kt + 4), DS(k, j) = s_{j-1} - s_j.
__global__ __launch_bounds__(256) void chord_table_fill(int L, const double *rtop, double *ds) {
  const int w = blockIdx.y, kt = blockIdx.x;
  const int nkt = (L + 15) / 16, ns = 4 * nkt;
  const double *rt = rtop + (size_t)w * L;
  double *tile = ds + (size_t)w * chord_table_size(L) + (size_t)kt * ns * 64;
  for (int t = threadIdx.x; t < (4 * kt + 4) * 64; t += 256) {
    const int lane = t & 63, k = 16 * kt + (lane & 15), j = 4 * (t >> 6) + (lane >> 4);
    double v = 0.0;
    if (k < L && j >= 1 && j <= k) {
      const double rk = rt[k], r0 = rt[j - 1], r1 = rt[j];
      const double s0 = sqrt((r0 - rk) * (r0 + rk));
      const double s1 = (j == k) ? 0.0 : sqrt((r1 - rk) * (r1 + rk));
      v = s0 - s1;
    }
    tile[t] = v;
  }
}

hipError_t launch_chord_table(const PrepArgs &a, hipStream_t st) {
  if (a.nwalkers <= 0 || !a.rtop || !a.ds) return hipSuccess;
  hipLaunchKernelGGL(chord_table_fill, dim3((a.L + 15) / 16, a.nwalkers), dim3(256), 0, st, a.L,
                     a.rtop, a.ds);
  return hipGetLastError();
}

// dynamic LDS above the 64 kB default has to be opted into, once per kernel
template <class K>
static hipError_t allow_lds(K kernel, size_t bytes, size_t &allowed) {
  if (bytes <= allowed) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) allowed = bytes;
  return e;
}

hipError_t launch_transit(const RtArgs &a, hipStream_t st) {
  if (a.nwalkers <= 0 || a.W <= 0) return hipSuccess;
  const size_t sh = sizeof(double) * ((size_t)a.L * coef_stride(a.M, a.C) +
                                      (size_t)a.L * idx_stride(a.C) + (size_t)a.L * 64);
  const size_t sh_pairs = sizeof(double) * (size_t)a.L * 64;
  if (sh_pairs > 160 * 1024) return hipErrorInvalidValue;  // more than 320 layers
  static const bool generic_only = [] {
    const char *e = std::getenv("BARTRT_KERNEL");
    return e && std::string(e) == "generic";
  }();
  static const bool force_window = std::getenv("BARTRT_WINDOW") != nullptr;  // (tests)
  const bool window = a.kappa_bytes >= (1ull << 32) - 4096 || force_window;
  const bool fits32 = a.cia_bytes < (1ull << 32) - 4096 && (!window || window_fits(a, 4));
  if (!generic_only && a.ext && a.M == 0 && (a.C <= 2 || a.C == 4) && !a.tau_out && a.cia_bytes < (1ull << 32) - 4096 &&
      a.L <= 16 * kMfmaTilesMax) {
    // line-by-line engines (no table): the matrix-tile kernel with the extinction array as input
    RtArgs b = a;
    b.window = 0;
    b.ntiles = (a.W + 63) / 64;
    const int nb = (b.ntiles + 7) / 8 * 8 * a.nwalkers;
    const size_t shm = sizeof(double) * ((size_t)a.L * coef_stride(0, a.C) + (size_t)a.L * idx_stride(a.C) + (size_t)a.L);
#define BARTRT_TRANSIT_EXT(CC)                                                                       \
  if (a.C == CC) {                                                                                   \
    if (a.L <= 16 * kMfmaTiles)                                                                      \
      BARTRT_RT_LAUNCH((rt_transit_mfma<0, CC, kMfmaTiles, true>), dim3(nb), dim3(256), shm, st, b);  \
    else if (a.L <= 16 * kMfmaTilesDeep)                                                             \
      BARTRT_RT_LAUNCH((rt_transit_mfma<0, CC, kMfmaTilesDeep, true>), dim3(nb), dim3(256), shm, st, b); \
    else                                                                                             \
      BARTRT_RT_LAUNCH((rt_transit_mfma<0, CC, kMfmaTilesMax, true>), dim3(nb), dim3(256), shm, st, b); \
    return hipGetLastError();                                                                        \
  }
    BARTRT_EXT_C_LIST(BARTRT_TRANSIT_EXT)
#undef BARTRT_TRANSIT_EXT
  }
  // (the matrix-tile kernels run on the default 64 kB of dynamic LDS: a column whose records need more --
  // eight molecules + two CIA pairs from 293 layers, seven from 316 -- takes the scalar kernel, which opts in)
  const size_t shm_tab = sizeof(double) * ((size_t)a.L * coef_stride(a.M, a.C) +
                                           (size_t)a.L * idx_stride(a.C) + (size_t)a.L);
  if (!generic_only && !a.ext && !a.tau_out && fits32 && a.L <= 16 * kMfmaTilesMax && shm_tab <= 64 * 1024) {
    RtArgs b = a;
    b.window = window;
    b.ntiles = (a.W + 63) / 64;
    const int nb = (b.ntiles + 7) / 8 * 8 * a.nwalkers;
    const size_t shm = shm_tab;
#define BARTRT_TRANSIT(MM, CC)                                                              \
  if (a.M == MM && a.C == CC) {                                                             \
    if (a.L <= 16 * kMfmaTiles)                                                             \
      BARTRT_RT_LAUNCH((rt_transit_mfma<MM, CC, kMfmaTiles>), dim3(nb), dim3(256), shm, st, b); \
    else if (a.L <= 16 * kMfmaTilesDeep)                                                    \
      BARTRT_RT_LAUNCH((rt_transit_mfma<MM, CC, kMfmaTilesDeep>), dim3(nb), dim3(256), shm, st, b); \
    else                                                                                    \
      BARTRT_RT_LAUNCH((rt_transit_mfma<MM, CC, kMfmaTilesMax>), dim3(nb), dim3(256), shm, st, b); \
    return hipGetLastError();                                                               \
  }
    BARTRT_MC_LIST(BARTRT_TRANSIT)
#undef BARTRT_TRANSIT
  }
  // the scalar kernel: one wave per 64 wavenumbers, whatever workgroup size the eclipse kernels run with
  // (BARTRT_BLOCK: a.ntiles counts tiles of that size)
  RtArgs b = a;
  b.ntiles = (a.W + 63) / 64;
  const int nb = (b.ntiles + 7) / 8 * 8 * a.nwalkers;
  static size_t allowed = 48 * 1024, allowed_pairs = 48 * 1024;
  if (sh <= 160 * 1024) {
    hipError_t e = allow_lds(rt_transit<true>, sh, allowed);
    if (e != hipSuccess) return e;
    BARTRT_RT_LAUNCH(rt_transit<true>, dim3(nb), dim3(64), sh, st, b);
  } else {
    hipError_t e = allow_lds(rt_transit<false>, sh_pairs, allowed_pairs);
    if (e != hipSuccess) return e;
    BARTRT_RT_LAUNCH(rt_transit<false>, dim3(nb), dim3(64), sh_pairs, st, b);
  }
  return hipGetLastError();
}

}  // namespace bartrt
