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
//  rt_eclipse_fast / _split / _quad: the same walk specialised at compile time
//                    (rt_eclipse.hpp, one translation unit per integration rule:
//                    rt_eclipse_i0.hip, _i1.hip, _i2.hip).
//
// The walker's coefficient records are staged in LDS once per workgroup and
// read back as wave-uniform broadcasts.  Interpolation + reduction, fp64 VALU;
// the transit geometry (transit_geom.hip) is where MFMA fits.
#include "integ.hpp"
#include "kernels.hpp"
#include "prep.hpp"

#include <cmath>
#include <cstdlib>
#include <string>
#include <utility>

namespace bartrt {

// specialised kernels, one instantiation set per integration rule (rt_eclipse_i*.hip)
template <int INTEG>
bool launch_rt_spec(const RtArgs &a, int block, hipStream_t st, const std::string &kmode, bool force_window,
                    bool allow_sq, hipError_t &err, RtLaunchInfo *info, const PrepArgs *fold = nullptr);
extern template bool launch_rt_spec<0>(const RtArgs &, int, hipStream_t, const std::string &, bool, bool, hipError_t &, RtLaunchInfo *, const PrepArgs *);
extern template bool launch_rt_spec<1>(const RtArgs &, int, hipStream_t, const std::string &, bool, bool, hipError_t &, RtLaunchInfo *, const PrepArgs *);
extern template bool launch_rt_spec<2>(const RtArgs &, int, hipStream_t, const std::string &, bool, bool, hipError_t &, RtLaunchInfo *, const PrepArgs *);

__global__ __launch_bounds__(256) void prep_profiles(PrepArgs p) {
  extern __shared__ double sm[];
  prep_block(p, blockIdx.x, sm);
}

// XCD-aware block -> (tile, walker) map (see rt_eclipse.hpp)
__device__ inline void block_to_work_g(int b, int nwalkers, int &tile, int &walker) {
  const int xcd = b & 7, j = b >> 3;
  walker = j % nwalkers;
  tile = (j / nwalkers) * 8 + xcd;
}

// Generic kernel: runtime ray-angle / molecule / CIA counts (AT > 0 / MT, CT >= 0
// fix them at compile time), plain pointer arithmetic, also carries the
// line-by-line extinction and the tau / per-angle outputs.  INTEG: integ.hpp.
template <int AT, int MT, int CT, int INTEG>
__global__ __launch_bounds__(256) void rt_eclipse(RtArgs p) {
  extern __shared__ double smem[];
  const int A = AT > 0 ? AT : p.A;
  const int M = MT >= 0 ? MT : p.M;
  const int C = CT >= 0 ? CT : p.C;
  const int L = p.L, W = p.W;
  const int NC = coef_stride(M, C), NI = idx_stride(C);
  int tile, w;
  block_to_work_g(blockIdx.x, p.nwalkers, tile, w);
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
  const int ii = valid ? i : W - 1;  // keep every lane's loads in range
  const double nu = p.wn[ii];
  const double bnum = 2.0 * kH * nu * nu * nu * kLS * kLS;
  const double nu4 = (nu * nu) * (nu * nu);
  const size_t MW = (size_t)M * W;

  constexpr int AMAX = AT > 0 ? AT : kMaxAngles;
  TauColumn<INTEG> tc;
  double Bprev = 0.0;
  bool active = true;
  int last = 0;
  const int kraw = p.kstop[w], kend = kstop_layer(kraw);
  const bool deck_on = kstop_deck(kraw);
  int k = 0;
  // extinction of layer k for this lane
  auto extinction = [&](int kk) {
    const double *c = sC + kk * NC;
    const idx_t *ix = sI + kk * NI;
    const int l = L - 1 - kk;
    double e = c[2 + 2 * M + 2 * C] * nu4 + c[3 + 2 * M + 2 * C];   // Rayleigh + grey cloud
    if (p.ext) e += p.ext[((size_t)w * L + l) * W + ii];
    // grid [plane][W][M], CIA [pair plane][W][2] (kernels.hpp, "Table layout")
    const double *kb = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p.kappa) + ix[0]) + (size_t)ii * M;
#pragma unroll
    for (int m = 0; m < (MT >= 0 ? MT : kMaxMol); m++) {
      if (MT < 0 && m >= M) break;
      e += c[2 + 2 * m] * kb[m] + c[3 + 2 * m] * kb[MW + m];
    }
#pragma unroll
    for (int cc = 0; cc < (CT >= 0 ? CT : kMaxCia); cc++) {
      if (CT < 0 && cc >= C) break;
      const double *ab = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p.cia) + ix[1 + cc]) + 2 * (size_t)ii;
      e += c[2 + 2 * M + 2 * cc] * ab[0] + c[3 + 2 * M + 2 * cc] * ab[1];
    }
    return e;
  };
  double Ia[AMAX];
  double F = 0.0;
  if (p.cut_slant) {
    // `cut slant`: every ray angle ends on the first layer whose SLANT depth passes
    // toomuch; the column is walked while any of them is alive (the vertical ray is the last to go)
    SlantRay<INTEG> ray[AMAX];
    bool alive[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; a++) alive[a] = a < A;
    for (; k <= kend; ++k) {
      const double *c = sC + k * NC;
      tc.layer(k, active, active ? 0.5 : 0.0, extinction(k), c[0], sW);
      const double B = bnum * rcp_n1(exp_rt(fmin(c[1] * nu, 700.0)) - 1.0);
      bool any = false;
#pragma unroll
      for (int a = 0; a < AMAX; a++) {
        if (AT <= 0 && a >= A) continue;
        const double x = tc.tau * p.invmu[a];
        ray[a].point(alive[a], x, B, exp_rt(fmax(-x, kExpMin)));
        alive[a] = alive[a] && !(x > p.toomuch);
        any = any || alive[a];
      }
      if (p.tau_out && valid) p.tau_out[(size_t)i * L + k] = tc.tau;
      if (active) last = k;
      active = any;     // the optical depth is needed while a ray is alive
      if (!__any(active)) break;
    }
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      if (AT <= 0 && a >= A) { Ia[a] = 0.0; continue; }
      Ia[a] = ray[a].result(deck_on && alive[a], L);
      F += p.wgt[a] * Ia[a];
    }
  } else {
    ColumnIntens<INTEG, AMAX> ci;
    for (; k <= kend; ++k) {
      const double *c = sC + k * NC;
      tc.layer(k, active, active ? 0.5 : 0.0, extinction(k), c[0], sW);
      const double B = bnum * rcp_n1(exp_rt(fmin(c[1] * nu, 700.0)) - 1.0);
      double E[AMAX];
#pragma unroll
      for (int a = 0; a < AMAX; a++) {
        if (AT <= 0 && a >= A) { E[a] = 0.0; continue; }
        E[a] = exp_rt(fmax(-tc.tau * p.invmu[a], kExpMin));
      }
      ci.layer(A, active, active ? 0.5 : 0.0, tc.tau, Bprev, B, E);
      Bprev = B;
      if (p.tau_out && valid) p.tau_out[(size_t)i * L + k] = tc.tau;
      if (active) {
        last = k;
        if (tc.tau > p.toomuch) active = false;
      }
      if (!__any(active)) break;
    }
    F = ci.flux(p, A, deck_on && active, Bprev, L, Ia);
  }
  if (p.intens_out && valid) {
    for (int a = 0; a < A; a++) p.intens_out[(size_t)a * W + i] = Ia[a];
  }
  if (valid) {
    p.spec[(size_t)w * W + i] = F;
    if (p.tau_out) {
      for (int kk = last + 1; kk < L; kk++) p.tau_out[(size_t)i * L + kk] = tc.tau;
      p.last_out[i] = last;
    }
  }
  if (p.walked_out && (threadIdx.x & 63) == 0)
    p.walked_out[(size_t)w * (p.ntiles * (blockDim.x / 64)) + tile * (blockDim.x / 64) + (threadIdx.x >> 6)] =
        k < kend ? k + 1 : kend + 1;
}

// The opacity file's order [planes][M][W] -> the kernels' [planes][W][M]
// (kernels.hpp, "Table layout"); one workgroup per (plane, 64-wavenumber tile)
__global__ __launch_bounds__(64) void grid_transpose(const double *src, double *dst, int M, int W, int ntiles) {
  const long plane = blockIdx.x / ntiles;
  const int i = (int)(blockIdx.x % ntiles) * 64 + threadIdx.x;
  if (i >= W) return;
  for (int m = 0; m < M; m++)
    dst[(plane * W + i) * M + m] = src[(plane * M + m) * W + i];
}

hipError_t launch_grid_transpose(const double *src, double *dst, long planes, int M, int W, hipStream_t st) {
  if (planes <= 0 || M <= 0 || W <= 0) return hipSuccess;
  const int ntiles = (W + 63) / 64;
  if (planes * ntiles > 0x7fffffffL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(grid_transpose, dim3((unsigned)(planes * ntiles)), dim3(64), 0, st, src, dst, M, W, ntiles);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
hipError_t launch_prep(const PrepArgs &a, hipStream_t st) {
  if (a.nwalkers <= 0) return hipSuccess;
  const size_t sh = sizeof(double) * prep_lds_doubles(a.L, a.S, a.Nt, a.ncia_temps);
  // (256 lanes: two per layer in the record loop up to 128 layers, prep_body)
  hipLaunchKernelGGL(prep_profiles, dim3(a.nwalkers), dim3(256), sh, st, a);
  return hipGetLastError();
}

template <int AT, int INTEG>
static hipError_t launch_rt_t(const RtArgs &a, int block, int nblocks, hipStream_t st) {
  const size_t sh = sizeof(double) * ((size_t)a.L * coef_stride(a.M, a.C) +
                                      (INTEG == kIntegSimpson ? simpson_lds_doubles(a.L) : 0)) +
                    sizeof(idx_t) * (size_t)a.L * idx_stride(a.C);
  // layer records above the 64 kB default (deep columns, many molecules): opt in once
  static size_t allowed = 64 * 1024;
  if (sh > allowed) {
    if (sh > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rt_eclipse<AT, -1, -1, INTEG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return e;
    allowed = sh;
  }
  BARTRT_RT_LAUNCH((rt_eclipse<AT, -1, -1, INTEG>), dim3(nblocks), dim3(block), sh, st, a);
  return hipGetLastError();
}

static const std::string &rt_kmode() {
  static const std::string kmode = [] {
    const char *e = std::getenv("BARTRT_KERNEL");  // generic | mono | split | quad | octo (A/B runs)
    return std::string(e ? e : "");
  }();
  return kmode;
}
static bool rt_force_window() {
  static const bool v = std::getenv("BARTRT_WINDOW") != nullptr;  // windowed addressing on any grid (tests)
  return v;
}
static bool rt_allow_sq() {
  static const bool v = [] {
    const char *e = std::getenv("BARTRT_SQ");  // 0: always evaluate every transmittance (A/B runs)
    return !(e && e[0] == '0');
  }();
  return v;
}

// One to four walkers under the default conventions: the layer-parallel kernels build their walker's layer records in
// their own prologue (RtArgs::nprep < 0) -- no prep_profiles launch, no launch boundary in front of the RT kernel.
// *folded = false: no such kernel serves this launch; the caller launches the preparation and calls launch_rt.
hipError_t launch_rt_folded(const RtArgs &a, const PrepArgs &prep, int block, hipStream_t st, RtLaunchInfo *info, bool *folded) {
  *folded = false;
  static const bool on = [] { const char *e = std::getenv("BARTRT_FOLD"); return !(e && e[0] == '0'); }();
  if (!on || a.nwalkers <= 0 || a.W <= 0 || a.integ != kIntegSimpson || rt_kmode() == "generic") return hipSuccess;
  hipError_t err = hipSuccess;
  *folded = launch_rt_spec<1>(a, block, st, rt_kmode(), rt_force_window(), rt_allow_sq(), err, info, &prep);
  return *folded ? err : hipSuccess;
}

// block: threads per workgroup (64 or 256); a.ntiles must be ceil(W/block).
hipError_t launch_rt(const RtArgs &a, int block, hipStream_t st, RtLaunchInfo *info) {
  if (a.nwalkers <= 0 || a.W <= 0) return hipSuccess;
  if (a.integ < 0 || a.integ >= kIntegCount) return hipErrorInvalidValue;
  const int ntiles8 = (a.ntiles + 7) / 8 * 8;
  const int nblocks = ntiles8 * a.nwalkers;
  const std::string &kmode = rt_kmode();
  const bool force_window = rt_force_window(), allow_sq = rt_allow_sq();
  if (kmode != "generic") {
    hipError_t err = hipSuccess;
    bool done = false;
    switch (a.integ) {
      case kIntegTransmittance: done = launch_rt_spec<0>(a, block, st, kmode, force_window, allow_sq, err, info); break;
      case kIntegSimpson: done = launch_rt_spec<1>(a, block, st, kmode, force_window, allow_sq, err, info); break;
      default: done = launch_rt_spec<2>(a, block, st, kmode, force_window, allow_sq, err, info); break;
    }
    if (done) return err;
  }
  if (info) {   // (the generic kernel does not carry a prefetched preparation)
    info->kernel = "rt_eclipse (generic)"; info->wn_per_column = 64; info->ncolumns = a.ntiles * (block / 64);
    info->prep_fused = false;
  }
  switch (a.integ * 2 + (a.A == 5 ? 1 : 0)) {
    case 0: return launch_rt_t<0, 0>(a, block, nblocks, st);
    case 1: return launch_rt_t<5, 0>(a, block, nblocks, st);
    case 2: return launch_rt_t<0, 1>(a, block, nblocks, st);
    case 3: return launch_rt_t<5, 1>(a, block, nblocks, st);
    case 4: return launch_rt_t<0, 2>(a, block, nblocks, st);
    default: return launch_rt_t<5, 2>(a, block, nblocks, st);
  }
}

}  // namespace bartrt
