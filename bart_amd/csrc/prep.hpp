// prep_profiles' body as a device function: shared by the stand-alone kernel
// (kernels.hip) and by the fused T(p) + prep kernel of the per-step path
// (step.hip), which builds the walker's profile in LDS itself.
#pragma once
#include "kernels.hpp"

// Phase clock of the latency-shaped preparation kernels (a development build only: tools/ab_build.py with
// -DBARTRT_PHASE_CLOCK, read back by tools/debug/phase_clock.py): workgroup 0's lane 0 stamps the 100 MHz
// real-time counter at the phase boundaries.
#ifdef BARTRT_PHASE_CLOCK
extern __device__ unsigned long long g_phase_clock[32];
#define BARTRT_PHASE(n) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_clock[n] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BARTRT_PHASE(n) do {} while (0)
#endif

namespace bartrt {

// Largest j with g[j] <= t, clamped to [0, n-2].  Starts from the uniform-grid
// guess (tlow/thigh/tempdelt grids are uniform; ginv[0] = 1/(g[1]-g[0])) and
// walks to the exact bracket.
__device__ inline int bracket_dev(const double *g, const double *ginv, int n, double t) {
  if (n <= 2) return 0;
  double x = (t - g[0]) * ginv[0];
  int j = x > 0.0 ? (x < (double)(n - 2) ? (int)x : n - 2) : 0;
  // both ends of the guessed interval in one trip to LDS: on a uniform grid the guess is the bracket, and the two
  // walks below -- a dependent LDS read per test, two even when nothing moves -- are skipped
  const double a = g[j], b = g[j + 1];
  if ((j == n - 2 || b > t) && (j == 0 || a <= t)) return j;
  while (j < n - 2 && g[j + 1] <= t) j++;
  while (j > 0 && g[j] > t) j--;
  return j;
}


// LDS layout, in doubles from `sm`: T[L], mu[L], r[L], H[L], the walker's
// profile [(S+1)][L], then the per-engine constants in the order of
// PrepArgs::consts.
__host__ __device__ inline size_t prep_lds_doubles(int L, int S, int Nt, int ncia_temps) {
  return (size_t)4 * L + (size_t)(S + 1) * L + 2 * (size_t)L + S + 2 * (size_t)Nt + 2 * (size_t)ncia_temps + 1;
}
__device__ inline double *prep_lds_profile(double *sm, int L) { return sm + 4 * L; }
__device__ inline double *prep_lds_consts(double *sm, int L, int S) { return sm + 4 * L + (size_t)(S + 1) * L; }

// Expects the walker's profile and the constants written to LDS by this workgroup
// (the body's first barrier makes them visible); w = walker index; 256 lanes in the stand-alone and per-step kernels (two
// per layer in the record loop up to 128 layers), 64-256 when folded into an RT kernel.
// over3 (optional; LDS or global): this walker's own reference radius (cm), cloud-top
// pressure (barye) and scattering value, NaN = keep the engine's setting -- the
// radius / cloud / scattering parameters of a retrieval step (BARTfunc.py:350-360).
// Folded into an RT kernel's prologue (PrepFold: the layer-parallel kernels prepare their own walker when the launch
// is one to four walkers -- rt_eclipse.hpp, launch_rt_folded): the records go to the workgroup's LDS arrays instead
// of global memory, and only one workgroup per walker (`global`) writes the walker's flags and radii out.
struct PrepFold {
  double *coef = nullptr;   // [L][stride] LDS
  int stride = 0;
  idx_t *idx = nullptr;     // [L][1 + C] LDS
  int *kstop = nullptr;     // LDS word
  bool global = true;       // write kstop / ok / radii to the PrepArgs outputs as the stand-alone kernel does
};

__device__ inline void prep_body(const PrepArgs &p, int w, double *sm, const double *over3, const PrepFold fold = PrepFold{}) {
  const int L = p.L, S = p.S, M = p.M, C = p.C;
  double refradius = p.refradius, cloudtop = p.cloudtop, scat_value = p.scat_value;
  int has_cloud = p.has_cloud;
  if (over3) {
    if (over3[0] == over3[0]) refradius = over3[0];
    if (over3[1] == over3[1]) { cloudtop = over3[1]; has_cloud = 1; }
    if (over3[2] == over3[2]) scat_value = over3[2];
  }
  double *sT = sm;           // temperature, atm order
  double *sMu = sm + L;      // mean molecular mass
  double *sR = sm + 2 * L;   // radius
  double *sH = sm + 3 * L;   // hydrostatic step terms
  double *sProf = prep_lds_profile(sm, L);          // [(S+1)][L]
  double *sPress = prep_lds_consts(sm, L, S);       // [L]
  double *sDlnp = sPress + L;                       // [L]
  double *sMass = sDlnp + L;                        // [S]
  double *sTg = sMass + S;                          // [Nt]
  double *sTgInv = sTg + p.Nt;                      // [Nt]
  double *sCiaT = sTgInv + p.Nt;                    // [ncia_temps]
  double *sCiaTInv = sCiaT + p.ncia_temps;          // [ncia_temps]
  const double *prof = sProf;
  __shared__ int sBad;
  if (threadIdx.x == 0) sBad = 0;
  __syncthreads();
  BARTRT_PHASE(8);
  // Hydrostatic radii, makeatm.py:229-258 (layers bottom -> top).  The
  // reference steps r_i = r_{i+-1} -+ H_i / g and rescales g by (r_old/r_new)^2,
  // i.e. g r^2 stays g0 R0^2: the step is r -+ (H_i / (g0 R0^2)) r^2.  The
  // layer terms H_i are formed in parallel (each lane also evaluates its upper
  // neighbour's T/mu rather than wait for it); only the two-flop recurrence is
  // serial (lane 0 walks down from the reference layer, lane 64 walks up).
  const double rgas = kKB / kAMU;
  {
    const double invG = 1.0 / (p.gsurf * refradius * refradius);
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
      const int lu = l + 1 < L ? l + 1 : l;
      const double T = prof[l], Tu = prof[lu];
      double mu = 0.0, muu = 0.0;
      for (int s = 0; s < S; s++) {
        mu += prof[(size_t)(s + 1) * L + l] * sMass[s];
        muu += prof[(size_t)(s + 1) * L + lu] * sMass[s];
      }
      sT[l] = T;
      sMu[l] = mu;
      sH[l] = 0.5 * (T / mu + Tu / muu) * (rgas * sDlnp[l]) * invG;
      if (!(T > 0.0) || !(T < 1e30) || !(mu > 0.0)) sBad = 1;
    }
  }
  __syncthreads();
  BARTRT_PHASE(9);
  const bool bad = sBad != 0;
  // (a 64-lane workgroup -- the preparation fused into an RT launch -- walks both chains in one wave)
  const unsigned up_lane = blockDim.x >= 128 ? 64 : blockDim.x / 2;
  if (!bad && (threadIdx.x == 0 || threadIdx.x == up_lane)) {
    const int ix = p.ref_idx;
    double r = refradius;
    if (!p.ref_exact) {
      const int b = p.ref_ib;
      const double t0 = sT[b] + p.ref_f * (sT[b + 1] - sT[b]);
      const double m0 = sMu[b] + p.ref_f * (sMu[b + 1] - sMu[b]);
      r += 0.5 * (sT[ix] / sMu[ix] + t0 / m0) * (rgas * p.ref_lnp / p.gsurf);
    }
    // blocks of 8 terms are fetched from LDS ahead of the dependent chain
    // (the NEXT block's terms are requested before the current block's chain starts: the LDS trip runs under it)
    if (threadIdx.x == 0) {
      sR[ix] = r;
      int i = ix - 1;
      double h[8], hn[8];
      if (i >= 7) {
#pragma unroll
        for (int j = 0; j < 8; j++) h[j] = sH[i - j];
      }
      for (; i >= 7; i -= 8) {
        if (i - 8 >= 7) {
#pragma unroll
          for (int j = 0; j < 8; j++) hn[j] = sH[i - 8 - j];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { r = fma(-h[j], r * r, r); sR[i - j] = r; }
#pragma unroll
        for (int j = 0; j < 8; j++) h[j] = hn[j];
      }
      for (; i >= 0; i--) { r = fma(-sH[i], r * r, r); sR[i] = r; }
    } else {
      int i = ix + 1;
      double h[8], hn[8];
      if (i + 7 < L) {
#pragma unroll
        for (int j = 0; j < 8; j++) h[j] = sH[i + j - 1];
      }
      for (; i + 7 < L; i += 8) {
        if (i + 8 + 7 < L) {
#pragma unroll
          for (int j = 0; j < 8; j++) hn[j] = sH[i + 8 + j - 1];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { r = fma(h[j], r * r, r); sR[i + j] = r; }
#pragma unroll
        for (int j = 0; j < 8; j++) h[j] = hn[j];
      }
      for (; i < L; i++) { r = fma(sH[i - 1], r * r, r); sR[i] = r; }
    }
  }
  __syncthreads();
  BARTRT_PHASE(10);
  const int NC = coef_stride(M, C), NI = idx_stride(C);
  double *coef = fold.coef ? fold.coef : p.coef + (size_t)w * L * NC;
  idx_t *idx = fold.idx ? fold.idx : p.idx + (size_t)w * L * NI;
  const int cstride = fold.coef ? fold.stride : NC;
  // A workgroup with two lanes per layer (256 lanes, up to 128 layers) splits a layer's record: the first half of the
  // lanes writes the path length, c2 / T and the table's plane offset and weights, the second half the cross-section
  // tables, Rayleigh and grey-cloud terms -- the same values from the same operations, half the dependent chain each
  // (the kernel is latency: 3.6 of its 16 us were this loop, phase clock of round 5).
  const int half = (int)(blockDim.x >> 1);
  const bool two = half >= L && half >= 64;
  const bool do_tab = !two || (int)threadIdx.x < half, do_cs = !two || (int)threadIdx.x >= half;
  for (int k = two ? (int)threadIdx.x % half : (int)threadIdx.x; k < L; k += two ? half : (int)blockDim.x) {
    const int l = L - 1 - k;
    double *c = coef + (size_t)k * cstride;
    idx_t *ix = idx + (size_t)k * NI;
    if (bad) {
      if (do_tab) {
        for (int j = 0; j < 2 + 2 * M; j++) c[j] = 0.0;
        ix[0] = 0;
        c[1] = 1.0;
      }
      if (do_cs) {
        for (int j = 2 + 2 * M; j < NC; j++) c[j] = 0.0;
        for (int j = 1; j < NI; j++) ix[j] = 0;
      }
      continue;
    }
    // one division per layer (1/T); grid spacings come as reciprocals
    const double T = sT[l];
    const double invT = 1.0 / T;
    const double nd = sPress[l] * invT * (1.0 / kKB);
    if (do_tab) {
      c[0] = (k == 0) ? 0.0 : (sR[l + 1] - sR[l]);
      c[1] = (kH * kLS / kKB) * invT;
      int j = 0;
      double f = 0.0;
      if (M > 0) {
        j = bracket_dev(sTg, sTgInv, p.Nt, T);
        f = (T - sTg[j]) * sTgInv[j];
      }
      ix[0] = (idx_t)(((size_t)l * p.Nt + j) * M * p.W) * 8;
      for (int m = 0; m < M; m++) {
        const int s = p.opmol[m];
        const double rho = prof[(size_t)(s + 1) * L + l] * sMass[s] * kAMU * nd;
        c[2 + 2 * m] = rho * (1.0 - f);
        c[3 + 2 * m] = rho * f;
      }
    }
    if (!do_cs) continue;
    for (int cc = 0; cc < C; cc++) {
      const int nt = p.cia_nt[cc];
      const double *tg = sCiaT + p.cia_toff[cc], *tginv = sCiaTInv + p.cia_toff[cc];
      const double Tc = T < tg[0] ? tg[0] : (T > tg[nt - 1] ? tg[nt - 1] : T);
      const double n1 = prof[(size_t)(p.cia_s1[cc] + 1) * L + l] * nd * (1.0 / kAMAGAT);
      const double n2 = prof[(size_t)(p.cia_s2[cc] + 1) * L + l] * nd * (1.0 / kAMAGAT);
      int jc = 0;
      double fc = 0.0;
      if (nt > 1) {
        jc = bracket_dev(tg, tginv, nt, Tc);
        fc = (Tc - tg[jc]) * tginv[jc];
      }
      // pair plane jc of table cc holds (alpha_jc, alpha_jc+1) per wavenumber (a
      // single-temperature table: one pair plane with its values twice)
      ix[1 + cc] = (idx_t)(p.cia_poff[cc] + jc) * p.W * 16;
      if (p.cia_kind[cc] == 0) {
        c[2 + 2 * M + 2 * cc] = n1 * n2 * (1.0 - fc);
        c[3 + 2 * M + 2 * cc] = n1 * n2 * fc;
      } else {
        // the spline's curvature terms on the planes of second derivatives (natural cubic spline in T)
        const double sa = 1.0 - fc, h = nt > 1 ? tg[jc + 1] - tg[jc] : 0.0, s6 = n1 * n2 * h * h * (1.0 / 6.0);
        c[2 + 2 * M + 2 * cc] = s6 * (sa * sa * sa - sa);
        c[3 + 2 * M + 2 * cc] = s6 * (fc * fc * fc - fc);
      }
    }
    double ray = 0.0;
    if (p.scat_flag == 1 && p.iH2 >= 0) {
      const double l4 = (kRayLambda0 * kRayLambda0) * (kRayLambda0 * kRayLambda0);
      ray = pow(10.0, scat_value) * kRaySigma0 * prof[(size_t)(p.iH2 + 1) * L + l] * nd * l4;
    } else if (p.scat_flag == 2) {
      const double k0 = 128.0 * (kPI * kPI * kPI * kPI * kPI) / 3.0;
      if (p.iH2 >= 0) ray += kPolH2 * kPolH2 * prof[(size_t)(p.iH2 + 1) * L + l] * nd;
      if (p.iHe >= 0) ray += kPolHe * kPolHe * prof[(size_t)(p.iHe + 1) * L + l] * nd;
      ray *= k0;
    }
    c[2 + 2 * M + 2 * C] = ray;
    // radius-ramp cloud on this walker's own hydrostatic radii
    double grey = 0.0;
    if (p.cloud_ext > 0.0) {
      const double r = sR[l];
      grey = r >= p.cloud_rup ? 0.0
             : (r <= p.cloud_rdown ? p.cloud_ext
                                   : p.cloud_ext * (p.cloud_rup - r) / (p.cloud_rup - p.cloud_rdown));
    }
    c[3 + 2 * M + 2 * C] = grey;
  }
  if (p.rad_out && fold.global)
    for (int l = threadIdx.x; l < L; l += blockDim.x) p.rad_out[(size_t)w * L + l] = bad ? 0.0 : sR[l];
  if (p.rtop && !bad && fold.global) {
    // the chord table itself is filled from these radii by chord_table_fill
    double *rt = p.rtop + (size_t)w * L;
    for (int k = threadIdx.x; k < L; k += blockDim.x) rt[k] = sR[L - 1 - k];
  }
  if (threadIdx.x == 0) {
    int ks = L - 1;
    if (has_cloud) {
      for (int k = 0; k < L; k++)
        if (sPress[L - 1 - k] >= cloudtop) { ks = k | kDeckBit; break; }
    }
    if (fold.kstop) *fold.kstop = ks;
    if (fold.global) {
      p.kstop[w] = ks;
      if (p.ok) p.ok[w] = bad ? 0 : 1;
    }
  }
  BARTRT_PHASE(11);
}

// One workgroup's whole job for walker w: everything the body reads from HBM (the walker's
// profile and the block of per-engine constants) is pulled into LDS in ONE batch of
// independent loads; the phases of prep_body then run out of LDS.  The kernel is pure latency:
// each dependent trip to memory it avoids is worth most of a microsecond.
__device__ inline void prep_block(const PrepArgs &p, int w, double *sm, const PrepFold fold = PrepFold{}) {
  const int L = p.L, S = p.S;
  BARTRT_PHASE(7);
  stage2_to_lds(prep_lds_profile(sm, L), p.prof + (size_t)w * (S + 1) * L, (S + 1) * L,
                prep_lds_consts(sm, L, S), p.consts, 2 * L + S + 2 * p.Nt + 2 * p.ncia_temps,
                threadIdx.x, blockDim.x);
  prep_body(p, w, sm, p.over ? p.over + (size_t)3 * w : nullptr, fold);
}

}  // namespace bartrt
