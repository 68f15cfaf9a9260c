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
#include "kernels.hpp"

namespace bartrt {

__global__ __launch_bounds__(64) void rt_transit(RtArgs p) {
  extern __shared__ double smem[];
  const int M = p.M, C = p.C, L = p.L, W = p.W;
  const int NC = coef_stride(M, C), NI = idx_stride(C);
  const int b = blockIdx.x;
  const int xcd = b & 7, jb = b >> 3;
  const int w = jb % p.nwalkers;
  const int tile = (jb / p.nwalkers) * 8 + xcd;
  if (tile >= p.ntiles) return;

  double *sC = smem;
  idx_t *sI = reinterpret_cast<idx_t *>(smem + (size_t)L * NC);
  // pair sums, [L][64], after the offset records
  double *sP = smem + (size_t)L * NC + (size_t)L * NI;
  {
    const double *gC = p.coef + (size_t)w * L * NC;
    const idx_t *gI = p.idx + (size_t)w * L * NI;
    stage2_to_lds(sC, gC, L * NC, sI, gI, L * NI, threadIdx.x, 64);
  }
  __syncthreads();

  const int i = tile * 64 + threadIdx.x;
  const bool valid = i < W;
  const int ii = valid ? i : W - 1;
  const double nu = p.wn[ii];
  const double nu4 = (nu * nu) * (nu * nu);
  const size_t MW = (size_t)M * W;
  const double *rt = p.rtop + (size_t)w * L;
  const double *dsw = p.ds + (size_t)w * L * L;

  const int kend = p.kstop[w];
  double eprev = 0.0, tau = 0.0, integ = 0.0, gprev = rt[0];
  bool active = true;
  int last = 0;
  for (int k = 0; k <= kend; ++k) {
    const double *c = sC + k * NC;
    const idx_t *ix = sI + k * NI;
    const int l = L - 1 - k;
    double e = c[2 + 2 * M + 2 * C] * nu4;
    if (p.ext) e += p.ext[((size_t)w * L + l) * W + ii];
    const double *kb = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p.kappa) + ix[0]) + ii;
    for (int m = 0; m < M; m++)
      e += c[2 + 2 * m] * kb[(size_t)m * W] + c[3 + 2 * m] * kb[MW + (size_t)m * W];
    for (int cc = 0; cc < C; cc++) {
      const double *ab = reinterpret_cast<const double *>(reinterpret_cast<const char *>(p.cia) + ix[1 + cc]) + ii;
      e += c[2 + 2 * M + 2 * cc] * ab[0] + c[3 + 2 * M + 2 * cc] * ab[W];
    }
    if (k > 0) {
      sP[(size_t)k * 64 + threadIdx.x] = eprev + e;
      const double *dk = dsw + (size_t)k * L;
      double t = 0.0;
      for (int j = 1; j <= k; j++) t = fma(sP[(size_t)j * 64 + threadIdx.x], dk[j], t);
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
    p.spec[(size_t)w * W + i] = (rt[0] * rt[0] - 2.0 * integ) * p.inv_starrad2;
    if (p.tau_out) {
      for (int k = last + 1; k < L; k++) p.tau_out[(size_t)i * L + k] = tau;
      p.last_out[i] = last;
    }
  }
}

hipError_t launch_transit(const RtArgs &a, hipStream_t st) {
  if (a.nwalkers <= 0 || a.W <= 0) return hipSuccess;
  const int ntiles8 = (a.ntiles + 7) / 8 * 8;
  const int nblocks = ntiles8 * a.nwalkers;
  const size_t sh = sizeof(double) * ((size_t)a.L * coef_stride(a.M, a.C) +
                                      (size_t)a.L * idx_stride(a.C) + (size_t)a.L * 64);
  if (sh > 160 * 1024) return hipErrorInvalidValue;
  if (sh > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rt_transit),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(rt_transit, dim3(nblocks), dim3(64), sh, st, a);
  return hipGetLastError();
}

}  // namespace bartrt
