// CDNA4 (gfx950) kernels of the forward radiative-transfer path.
//
//  prep_profiles   one workgroup per walker: mean molecular mass, hydrostatic
//                  radii (reference law code/makeatm.py:183-263), densities and
//                  the per-layer interpolation weights -> coefficient records.
//  rt_eclipse      one thread per (walker, wavenumber): streams the opacity
//                  grid layer by layer from the top (coalesced along the
//                  wavenumber axis, the grid's fastest index), accumulates the
//                  optical depth and the emergent intensity per ray angle,
//                  stops the wave once every lane passed `toomuch`.
//
// The walker's coefficient records are staged in LDS once per workgroup and
// read back as wave-uniform broadcasts.  HBM-bound: no MFMA anywhere.
#include "kernels.hpp"

namespace bartrt {

// ---------------------------------------------------------------------------
__device__ inline int bracket_dev(const double *g, int n, double t) {
  int j = 0;
  while (j < n - 2 && g[j + 1] <= t) j++;
  return j;
}

__global__ __launch_bounds__(128) void prep_profiles(PrepArgs p) {
  extern __shared__ double sm[];
  const int L = p.L, S = p.S, M = p.M, C = p.C;
  double *sT = sm;           // temperature, atm order
  double *sMu = sm + L;      // mean molecular mass
  double *sR = sm + 2 * L;   // radius
  const int w = blockIdx.x;
  const double *prof = p.prof + (size_t)w * (S + 1) * L;
  __shared__ int sBad;
  if (threadIdx.x == 0) sBad = 0;
  __syncthreads();
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    double T = prof[l];
    double mu = 0.0;
    for (int s = 0; s < S; s++) mu += prof[(size_t)(s + 1) * L + l] * p.mass[s];
    sT[l] = T;
    sMu[l] = mu;
    if (!(T > 0.0) || !(T < 1e30) || !(mu > 0.0)) sBad = 1;
  }
  __syncthreads();
  const bool bad = sBad != 0;
  if (threadIdx.x == 0 && !bad) {
    // Hydrostatic radii: sequential recurrence with g ~ 1/r^2
    // (makeatm.py:229-258), layers bottom -> top.
    const double rgas = kKB / kAMU;
    const int ix = p.ref_idx;
    const double r0 = p.refradius, g0 = p.gsurf;
    double g;
    if (!p.ref_exact) {
      const int b = p.ref_ib;
      double t0 = sT[b] + p.ref_f * (sT[b + 1] - sT[b]);
      double m0 = sMu[b] + p.ref_f * (sMu[b + 1] - sMu[b]);
      sR[ix] = r0 + 0.5 * (sT[ix] / sMu[ix] + t0 / m0) * (rgas * p.ref_lnp / g0);
      g = g0 * r0 * r0 / (sR[ix] * sR[ix]);
    } else {
      sR[ix] = r0;
      g = g0;
    }
    double gi = g;
    for (int i = ix - 1; i >= 0; i--) {
      sR[i] = sR[i + 1] - 0.5 * (sT[i] / sMu[i] + sT[i + 1] / sMu[i + 1]) *
                              (rgas * p.dlnp[i] / gi);
      gi = gi * sR[i + 1] * sR[i + 1] / (sR[i] * sR[i]);
    }
    gi = g;
    for (int i = ix + 1; i < L; i++) {
      sR[i] = sR[i - 1] + 0.5 * (sT[i] / sMu[i] + sT[i - 1] / sMu[i - 1]) *
                              (rgas * p.dlnp[i - 1] / gi);
      gi = gi * sR[i - 1] * sR[i - 1] / (sR[i] * sR[i]);
    }
  }
  __syncthreads();
  const int NC = coef_stride(M, C), NI = idx_stride(C);
  double *coef = p.coef + (size_t)w * L * NC;
  int *idx = p.idx + (size_t)w * L * NI;
  for (int k = threadIdx.x; k < L; k += blockDim.x) {
    const int l = L - 1 - k;
    double *c = coef + (size_t)k * NC;
    int *ix = idx + (size_t)k * NI;
    if (bad) {
      for (int j = 0; j < NC; j++) c[j] = 0.0;
      for (int j = 0; j < NI; j++) ix[j] = 0;
      c[1] = 1.0;
      continue;
    }
    const double T = sT[l];
    const double nd = p.press[l] / (kKB * T);
    c[0] = (k == 0) ? 0.0 : (sR[l + 1] - sR[l]);
    c[1] = (kH * kLS / kKB) / T;
    int j = 0;
    double f = 0.0;
    if (M > 0) {
      j = bracket_dev(p.tgrid, p.Nt, T);
      f = (T - p.tgrid[j]) / (p.tgrid[j + 1] - p.tgrid[j]);
    }
    ix[0] = j;
    for (int m = 0; m < M; m++) {
      const int s = p.opmol[m];
      const double rho = prof[(size_t)(s + 1) * L + l] * p.mass[s] * kAMU * nd;
      c[2 + 2 * m] = rho * (1.0 - f);
      c[3 + 2 * m] = rho * f;
    }
    for (int cc = 0; cc < C; cc++) {
      const int nt = p.cia_nt[cc];
      const double *tg = p.cia_temp + p.cia_toff[cc];
      const double Tc = T < tg[0] ? tg[0] : (T > tg[nt - 1] ? tg[nt - 1] : T);
      const double n1 = prof[(size_t)(p.cia_s1[cc] + 1) * L + l] * nd / kAMAGAT;
      const double n2 = prof[(size_t)(p.cia_s2[cc] + 1) * L + l] * nd / kAMAGAT;
      int jc = 0;
      double fc = 0.0;
      if (nt > 1) {
        jc = bracket_dev(tg, nt, Tc);
        fc = (Tc - tg[jc]) / (tg[jc + 1] - tg[jc]);
      }
      // a single-temperature table is stored twice so plane jc+1 exists
      ix[1 + cc] = p.cia_toff[cc] + cc + jc;
      c[2 + 2 * M + 2 * cc] = n1 * n2 * (1.0 - fc);
      c[3 + 2 * M + 2 * cc] = n1 * n2 * fc;
    }
    double ray = 0.0;
    if (p.scat_flag == 1 && p.iH2 >= 0) {
      const double l4 = (kRayLambda0 * kRayLambda0) * (kRayLambda0 * kRayLambda0);
      ray = pow(10.0, p.scat_value) * kRaySigma0 * prof[(size_t)(p.iH2 + 1) * L + l] * nd * l4;
    } else if (p.scat_flag == 2) {
      const double k0 = 128.0 * (kPI * kPI * kPI * kPI * kPI) / 3.0;
      if (p.iH2 >= 0) ray += kPolH2 * kPolH2 * prof[(size_t)(p.iH2 + 1) * L + l] * nd;
      if (p.iHe >= 0) ray += kPolHe * kPolHe * prof[(size_t)(p.iHe + 1) * L + l] * nd;
      ray *= k0;
    }
    c[2 + 2 * M + 2 * C] = ray;
  }
  if (threadIdx.x == 0) {
    int ks = L - 1;
    if (p.has_cloud) {
      for (int k = 0; k < L; k++)
        if (p.press[L - 1 - k] >= p.cloudtop) { ks = k; break; }
    }
    p.kstop[w] = ks;
    if (p.ok) p.ok[w] = bad ? 0 : 1;
  }
}

// ---------------------------------------------------------------------------
// XCD-aware block -> (tile, walker) map.  Blocks b and b+8 share an XCD (and
// its L2), so all walkers of one wavenumber tile are placed on one XCD, walker
// index fastest: they stream the same grid rows at about the same time and
// the XCD's L2 serves the repeats.
__device__ inline void block_to_work(int b, int ntiles8, int nwalkers, int &tile, int &walker) {
  const int xcd = b & 7, j = b >> 3;
  walker = j % nwalkers;
  tile = (j / nwalkers) * 8 + xcd;
  (void)ntiles8;
}

template <int AT, int MT, int CT>
__global__ __launch_bounds__(256) void rt_eclipse(RtArgs p) {
  extern __shared__ double smem[];
  const int A = AT > 0 ? AT : p.A;
  const int M = MT >= 0 ? MT : p.M;
  const int C = CT >= 0 ? CT : p.C;
  const int L = p.L, W = p.W, Nt = p.Nt;
  const int NC = coef_stride(M, C), NI = idx_stride(C);
  int tile, w;
  block_to_work(blockIdx.x, 0, p.nwalkers, tile, w);
  if (tile >= p.ntiles) return;

  double *sC = smem;
  int *sI = reinterpret_cast<int *>(smem + (size_t)L * NC);
  {
    const double *gC = p.coef + (size_t)w * L * NC;
    const int *gI = p.idx + (size_t)w * L * NI;
    for (int t = threadIdx.x; t < L * NC; t += blockDim.x) sC[t] = gC[t];
    for (int t = threadIdx.x; t < L * NI; t += blockDim.x) sI[t] = gI[t];
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
  for (int a = 0; a < AMAX; a++) { I[a] = 0.0; fprev[a] = 0.0; }

  double tau = 0.0, eprev = 0.0;
  bool active = true;
  int last = 0;
  const int kend = p.kstop[w];
  for (int k = 0; k <= kend; ++k) {
    const double *c = sC + k * NC;
    const int *ix = sI + k * NI;
    const int l = L - 1 - k;
    double e = c[2 + 2 * M + 2 * C] * nu4;
    const double *kb = p.kappa + ((size_t)l * Nt + ix[0]) * MW + ii;
#pragma unroll
    for (int m = 0; m < (MT >= 0 ? MT : kMaxMol); m++) {
      if (MT < 0 && m >= M) break;
      e += c[2 + 2 * m] * kb[(size_t)m * W] + c[3 + 2 * m] * kb[MW + (size_t)m * W];
    }
#pragma unroll
    for (int cc = 0; cc < (CT >= 0 ? CT : kMaxCia); cc++) {
      if (CT < 0 && cc >= C) break;
      const double *ab = p.cia + (size_t)ix[1 + cc] * W + ii;
      e += c[2 + 2 * M + 2 * cc] * ab[0] + c[3 + 2 * M + 2 * cc] * ab[W];
    }
    const double dtau = active ? 0.5 * (eprev + e) * c[0] : 0.0;
    tau += dtau;
    const double B = bnum / (exp(c[1] * nu) - 1.0);
#pragma unroll
    for (int a = 0; a < AMAX; a++) {
      if (AT <= 0 && a >= A) break;
      const double f = B * exp(-tau * p.invmu[a]);
      I[a] += 0.5 * (fprev[a] + f) * dtau;
      fprev[a] = f;
    }
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
    F += p.wgt[a] * (I[a] * p.invmu[a] + (surf ? fprev[a] : 0.0));
  }
  if (valid) {
    p.spec[(size_t)w * W + i] = F;
    if (p.tau_out) {
      for (int k = last + 1; k < L; k++) p.tau_out[(size_t)i * L + k] = tau;
      p.last_out[i] = last;
    }
  }
}

// ---------------------------------------------------------------------------
hipError_t launch_prep(const PrepArgs &a, hipStream_t st) {
  if (a.nwalkers <= 0) return hipSuccess;
  size_t sh = sizeof(double) * 3 * a.L;
  hipLaunchKernelGGL(prep_profiles, dim3(a.nwalkers), dim3(128), sh, st, a);
  return hipGetLastError();
}

template <int AT, int MT, int CT>
static hipError_t launch_rt_t(const RtArgs &a, int block, int nblocks, size_t sh, hipStream_t st) {
  hipLaunchKernelGGL((rt_eclipse<AT, MT, CT>), dim3(nblocks), dim3(block), sh, st, a);
  return hipGetLastError();
}

// block: threads per workgroup (64 or 256); a.ntiles must be ceil(W/block).
hipError_t launch_rt(const RtArgs &a, int block, hipStream_t st) {
  if (a.nwalkers <= 0 || a.W <= 0) return hipSuccess;
  const int ntiles8 = (a.ntiles + 7) / 8 * 8;
  const int nblocks = ntiles8 * a.nwalkers;
  const size_t sh = sizeof(double) * (size_t)a.L * coef_stride(a.M, a.C) +
                    sizeof(int) * (size_t)a.L * idx_stride(a.C);
  if (a.A == 5 && a.M == 4 && a.C == 1) return launch_rt_t<5, 4, 1>(a, block, nblocks, sh, st);
  if (a.A == 5 && a.M == 1 && a.C == 1) return launch_rt_t<5, 1, 1>(a, block, nblocks, sh, st);
  if (a.A == 5) return launch_rt_t<5, -1, -1>(a, block, nblocks, sh, st);
  return launch_rt_t<0, -1, -1>(a, block, nblocks, sh, st);
}

}  // namespace bartrt
