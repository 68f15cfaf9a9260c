// Device-side argument blocks shared by the kernels and the host engine.
#pragma once
#include <hip/hip_runtime.h>

namespace bartrt {

// Physical constants, cgs.  H, LS, KB: the values BART copies from transit
// (reference code/constants.py:13-16).
constexpr double kH = 6.6260755e-27;
constexpr double kLS = 2.99792458e10;
constexpr double kKB = 1.380658e-16;
constexpr double kAMU = 1.66053886e-24;
constexpr double kAMAGAT = 2.68679e19;
constexpr double kPI = 3.141592653589793;
constexpr double kRaySigma0 = 2.52e-28, kRayLambda0 = 7.5e-5;
constexpr double kPolH2 = 0.8059e-24, kPolHe = 0.2051e-24;

constexpr int kMaxAngles = 16;
constexpr int kMaxCia = 4;
constexpr int kMaxMol = 16;

// Per-layer coefficient record produced by prep_profiles, consumed by the RT
// kernel from LDS.  Layer order is top -> bottom (k = 0 is the top layer).
//   [0]            dr_k   = r_{k-1} - r_k (cm), 0 for k = 0
//   [1]            c2/T_k = (h c / k_B) / T_k  (cm)
//   [2 .. 2+2M)    (rho_m (1-f), rho_m f) per table molecule
//   [.. +2C)       (n1 n2 (1-f), n1 n2 f) / amagat^2 per CIA table
//   [last]         Rayleigh coefficient (multiplies wn^4)
__host__ __device__ inline int coef_stride(int M, int C) { return 3 + 2 * M + 2 * C; }
// Offset record (64-bit BYTE offsets, so the kernels add one scalar to a base
// pointer per layer): [0] start of the (layer, lower temperature) plane of the
// opacity grid, [1+c] start of the lower CIA plane of pair c.
using idx_t = long long;
__host__ __device__ inline int idx_stride(int C) { return 1 + C; }

// Cooperative copy of two arrays of 8-byte words from global memory into LDS.
// A thread issues up to U independent loads of each array before its first LDS
// store.  (The plain `for (t = tid; t < n; t += nthreads) dst[t] = src[t]` loop
// compiles, for a runtime trip count, to load / wait / store per element: n /
// nthreads serial trips to memory at the head of every workgroup.)
template <int U = 8, typename TA, typename TB>
__device__ __forceinline__ void stage2_to_lds(TA *dstA, const TA *srcA, int nA, TB *dstB, const TB *srcB,
                                              int nB, int tid, int nthreads) {
  static_assert(sizeof(TA) == 8 && sizeof(TB) == 8, "8-byte words");
  const int span = U * nthreads;
  for (int base = 0; base < nA || base < nB; base += span) {
    TA a[U];
    TB b[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int t = base + j * nthreads + tid;
      a[j] = t < nA ? srcA[t] : TA(0);
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int t = base + j * nthreads + tid;
      b[j] = t < nB ? srcB[t] : TB(0);
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int t = base + j * nthreads + tid;
      if (t < nA) dstA[t] = a[j];
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int t = base + j * nthreads + tid;
      if (t < nB) dstB[t] = b[j];
    }
  }
}

struct PrepArgs {
  int L, S, M, Nt, C, W, nwalkers;
  const double *prof;      // [nw][(S+1)][L]
  // per-engine constants in one block, in the order prep_profiles lays them out in
  // LDS: press[L] (barye, atm order, 0 = bottom), dlnp[L] (log(p[i]/p[i+1]), last
  // 0), mass[S] (amu), tgrid[Nt], 1/(tgrid[j+1]-tgrid[j]) [Nt], cia_temp[ncia_temps]
  // (concatenated), its reciprocal spacings [ncia_temps]
  const double *consts;
  int opmol[kMaxMol];      // [M] species index
  int cia_s1[kMaxCia], cia_s2[kMaxCia], cia_nt[kMaxCia], cia_toff[kMaxCia];
  int ncia_temps;          // length of the concatenated CIA temperature grids
  // hydrostatic reference (code/makeatm.py:183-263)
  int ref_idx;             // layer closest to refpress
  int ref_exact;           // press[ref_idx] == refpress
  int ref_ib;              // bracket [ib, ib+1] holding refpress in log10 p
  double ref_f;            // interpolation fraction inside the bracket
  double ref_lnp;          // log(refpress / press[ref_idx])
  double gsurf, refradius;
  // scattering / cloud
  int scat_flag, iH2, iHe, has_cloud;
  double scat_value, cloudtop;
  // outputs
  double *coef;            // [nw][L][coef_stride]
  idx_t *idx;              // [nw][L][idx_stride]
  int *kstop;              // [nw] deepest layer index k to integrate to
  unsigned char *ok;       // [nw]
  double *rad_out;         // optional [nw][L] hydrostatic radii, cm, atm layer order
  // transit geometry only (null otherwise): radii top -> bottom and the chord
  // segments ds[k][j] = s_{j-1} - s_j, s_j = sqrt(r_j^2 - r_k^2), j = 1..k
  double *rtop;            // [nw][L]
  double *ds;              // [nw][L][L]
};

struct RtArgs {
  int L, M, Nt, C, A, W, nwalkers, ntiles;
  const double *kappa;     // [L][Nt][M][W]
  const double *cia;       // [planes][W]
  unsigned long long kappa_bytes, cia_bytes;  // extents of the two tables
  const double *ext;       // optional line-by-line extinction [nw][L][W] (atm layer order)
  const double *wn;        // [W]
  const double *coef;
  const idx_t *idx;
  const int *kstop;
  int cloud_on;            // kstop marks a cloud deck (adds surface emission)
  double toomuch;
  double invmu[kMaxAngles];
  double wgt[kMaxAngles];  // pi (sin^2 hi - sin^2 lo)
  double *spec;            // [nw][W]
  double *tau_out;         // optional [W][L] (single walker), may be null
  int *last_out;           // optional [W]
  double *intens_out;      // optional [A][W] intensities per ray angle (single walker)
  // transit geometry
  const double *rtop, *ds;
  double inv_starrad2;
};

}  // namespace bartrt
