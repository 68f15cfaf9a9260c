// Per-step converters around the engine, batched over walkers on the GPU:
//   params -> T(p) ("line" model of Line et al. 2013, reference
//   code/PT.py:589-701 with xi() of PT.py:722-739), temperature-bounds check,
//   abundance scaling and H2/He renormalisation (code/BARTfunc.py:327-347),
//   then, after the RT kernel, the energy-balance check (BARTfunc.py:366-383)
//   and the band integration (code/wine.py:177-199, BARTfunc.py:386-396).
#include "step.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace bartrt {

// ---------------------------------------------------------------------------
// Exponential integral E_2(x), x >= 0.  Power series (Abramowitz & Stegun
// 5.1.12) for x <= 1, Lentz-free continued fraction (A&S 5.1.22) for x > 1:
// the two classical evaluations scipy.special.expn(2, x) is built on, so the
// reference's xi() is matched to rounding.
__device__ inline double expint_e2(double x) {
  const double EUL = 0.57721566490153286061, EPS = 1.11022302462515654042e-16;
  const double BIG = 1.44115188075855872e+17;
  if (x > 7.09782712893383996843e2) return 0.0;
  if (x == 0.0) return 1.0;
  if (x <= 1.0) {
    // E_2(x) = -x (psi(2) - ln x) - sum_{m != 1} (-x)^m / ((m - 1) m!)
    const double psi = -EUL - log(x) + 1.0;
    const double z = -x;
    double xk = 0.0, yk = 1.0, pk = -1.0, ans = -1.0, t;
    do {
      xk += 1.0;
      yk *= z / xk;
      pk += 1.0;
      if (pk != 0.0) ans += yk / pk;
      t = (ans != 0.0) ? fabs(yk / ans) : 1.0;
    } while (t > EPS);
    return z * psi - ans;
  }
  int k = 1;
  double pkm2 = 1.0, qkm2 = x, pkm1 = 1.0, qkm1 = x + 2.0, ans = pkm1 / qkm1, t;
  do {
    k += 1;
    double yk, xk;
    if (k & 1) { yk = 1.0; xk = 2.0 + (double)((k - 1) / 2); }
    else       { yk = x;   xk = (double)(k / 2); }
    const double pk = pkm1 * yk + pkm2 * xk, qk = qkm1 * yk + qkm2 * xk;
    if (qk != 0.0) {
      const double r = pk / qk;
      t = fabs((ans - r) / r);
      ans = r;
    } else {
      t = 1.0;
    }
    pkm2 = pkm1; pkm1 = pk; qkm2 = qkm1; qkm1 = qk;
    if (fabs(pk) > BIG) { pkm2 /= BIG; pkm1 /= BIG; qkm2 /= BIG; qkm1 /= BIG; }
  } while (t > EPS);
  return ans * exp(-x);
}

// PT.py:722-739
__device__ inline double xi_line(double gamma, double tau) {
  return (2.0 / 3) * (1 + (1. / gamma) * (1 + (0.5 * gamma * tau - 1) * exp(-gamma * tau)) +
                      gamma * (1 - 0.5 * tau * tau) * expint_e2(gamma * tau));
}

struct StepDev {
  int L, S, nPT, pttype, nmolfit, npars, tint_thorngren, iH2, iHe;
  double rstar, tstar, tint, sma, grav, tmin, tmax;
  const double *abund, *ratio, *pbar;
  const int *imol;
  const unsigned char *metal;
  // smoothing of the Madhusudhan and Piette profiles (scipy gaussian_filter1d,
  // mode 'nearest'): weights gw[0..2*grad], radius grad
  const double *gw;
  int grad;
  double ptop, pbot;      // min / max pressure of the grid, bar
  int pnode[8];           // piette: layers closest to top, 0.01, 0.1, 1, 3.2, 10, 32 bar, bottom
};

// Raw (unsmoothed) temperature of one layer; sets *bad for parameter sets the
// reference rejects with ValueError (PT.py:332-335, 546-548).
__device__ inline double pt_raw(const StepDev &a, const double *par, double p, int l, int *bad,
                                double kappa, double g1, double g2, double alpha, double Tirr,
                                double Tint) {
  switch (a.pttype) {
    case PT_LINE: {  // PT.py:687-699
      const double tau = kappa * (p * 1e6) / a.grav;
      const double x1 = xi_line(g1, tau), x2 = xi_line(g2, tau);
      const double ti4 = Tint * Tint * Tint * Tint, tr4 = Tirr * Tirr * Tirr * Tirr;
      return pow(0.75 * (ti4 * (2.0 / 3.0 + tau) + tr4 * (1 - alpha) * x1 + tr4 * alpha * x2), 0.25);
    }
    case PT_ISO:  // PT.py:719
      return par[0];
    case PT_MADHU_NOINV: {  // PT.py:539-576
      const double a1 = par[0], a2 = par[1], p1 = par[2], p3 = par[3], T3 = par[4], p0 = a.ptop;
      const double d31 = log(p3 / p1) / a2, d10 = log(p1 / p0) / a1;
      const double T1 = T3 - d31 * d31, T0 = T1 - d10 * d10;
      if (T0 < 0 || T1 < 0 || T3 < 0) *bad = 1;
      // the reference assigns the three regions in turn: the last match wins
      if (p >= p3 && p <= a.pbot) return T3;
      if (p >= p1 && p < p3) { const double d = log(p / p1) / a2; return d * d + T1; }
      if (p >= p0 && p < p1) { const double d = log(p / p0) / a1; return d * d + T0; }
      return 0.0;
    }
    case PT_MADHU_INV: {  // PT.py:316-370
      const double a1 = par[0], a2 = par[1], p1 = par[2], p2 = par[3], p3 = par[4], T3 = par[5];
      const double p0 = a.ptop;
      const double d32 = log(p3 / p2) / a2, d12 = log(p1 / p2) / -a2, d10 = log(p1 / p0) / a1;
      const double T2 = T3 - d32 * d32;
      const double T0 = T2 + d12 * d12 - d10 * d10;
      const double T1 = T0 + d10 * d10;
      if (T0 < 0 || T1 < 0 || T2 < 0 || T3 < 0) *bad = 1;
      // four regions assigned in turn (PT.py:367-370): the last match wins,
      // which matters when p3 < p2
      if (p >= p3 && p <= a.pbot) return T3;
      if (p >= p2 && p < p3) { const double d = log(p / p2) / a2; return d * d + T2; }
      if (p >= p1 && p < p2) { const double d = log(p / p2) / -a2; return d * d + T2; }
      if (p >= p0 && p < p1) { const double d = log(p / p0) / a1; return d * d + T0; }
      return 0.0;
    }
    case PT_ADIABATIC: {  // PT.py:747-749
      const double T0 = par[0], gam = par[1], p0 = pow(10.0, par[2]);
      return T0 / (1 + (gam - 1) / gam * log(p0 / p));
    }
    case PT_PIETTE: {  // PT.py:786-809: node temperatures, then linear in log10 p
      // nodes ordered top -> bottom: top, 0.01, 0.1, 1, 3.2, 10, 32 bar, bottom
      double tn[8];
      tn[4] = par[0];
      tn[5] = par[0] + par[3];
      tn[6] = tn[5] + par[2];
      tn[7] = tn[6] + par[1];
      tn[3] = par[0] - par[4];
      tn[2] = tn[3] - par[5];
      tn[1] = tn[2] - par[6];
      tn[0] = tn[1] - par[7];
      const double x = log10(p);
      int s = 0;  // segment [s, s+1] with x between the nodes' log10 p
      for (int j = 1; j < 7; j++)
        if (x >= log10(a.pbar[a.pnode[j]])) s = j;
      const double x0 = log10(a.pbar[a.pnode[s]]), x1 = log10(a.pbar[a.pnode[s + 1]]);
      if (l == a.pnode[s]) return tn[s];
      if (l == a.pnode[s + 1]) return tn[s + 1];
      return ((x1 - x) * tn[s] + (x - x0) * tn[s + 1]) / (x1 - x0);
    }
  }
  return 0.0;
}

// One workgroup per walker; lanes over layers.
__global__ __launch_bounds__(128) void step_profiles(StepDev a, const double *params,
                                                     double *prof, int *status) {
  extern __shared__ double sTraw[];
  const int w = blockIdx.x, L = a.L, S = a.S;
  const double *par = params + (size_t)w * a.npars;
  double *pr = prof + (size_t)w * (S + 1) * L;
  __shared__ int sBadT, sBadQ;
  if (threadIdx.x == 0) { sBadT = 0; sBadQ = 0; }
  __syncthreads();
  double kappa = 0, g1 = 0, g2 = 0, alpha = 0, Tirr = 0, Tint = a.tint;
  if (a.pttype == PT_LINE) {  // PT.py:675-688
    kappa = pow(10.0, par[0]);
    g1 = pow(10.0, par[1]);
    g2 = pow(10.0, par[2]);
    alpha = par[3];
    const double teq = sqrt(a.rstar / (2.0 * a.sma)) * a.tstar;
    if (a.tint_thorngren) {
      // PT.py:680-685, sigma_SB of scipy.constants (CODATA 2018)
      const double F = 4.0 * 5.670374419e-8 * teq * teq * teq * teq;
      const double d = log(F) - 0.14;
      Tint = 1.24 * teq * exp(-(d * d) / 2.96);
    }
    Tirr = par[4] * teq;
  }
  const bool smooth = a.grad > 0;
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    int bad = 0;
    const double T = pt_raw(a, par, a.pbar[l], l, &bad, kappa, g1, g2, alpha, Tirr, Tint);
    if (bad) sBadT = 1;   // the reference raises ValueError: rejected here
    sTraw[l] = T;
  }
  __syncthreads();
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    double T = sTraw[l];
    if (smooth) {
      // scipy.ndimage correlate1d, symmetric kernel, edge values repeated
      const int r = a.grad;
      T = sTraw[l] * a.gw[r];
      for (int j = r; j >= 1; j--) {
        const int lo = l - j < 0 ? 0 : l - j, hi = l + j > L - 1 ? L - 1 : l + j;
        T += (sTraw[lo] + sTraw[hi]) * a.gw[r - j];
      }
    }
    pr[l] = T;
    // BARTfunc.py:327 (NaN compares false on both sides, as in numpy)
    if (T < a.tmin || T > a.tmax) sBadT = 1;
    // BARTfunc.py:333-347
    const double *ab = a.abund + (size_t)l * S;
    double sm = 0.0;  // numpy sums the metals in species order, then 1 - sum
    for (int s = 0; s < S; s++) {
      double v = ab[s];
      for (int i = 0; i < a.nmolfit; i++)
        if (a.imol[i] == s) v = ab[s] * pow(10.0, par[a.npars - a.nmolfit + i]);
      pr[(size_t)(s + 1) * L + l] = v;
      if (a.metal[s]) sm += v;
    }
    const double q = 1.0 - sm;
    if (q < 0.0) sBadQ = 1;
    const double r = a.ratio[l];
    if (a.iH2 >= 0) pr[(size_t)(a.iH2 + 1) * L + l] = r * q / (1.0 + r);
    if (a.iHe >= 0) pr[(size_t)(a.iHe + 1) * L + l] = q / (1.0 + r);
  }
  __syncthreads();
  if (threadIdx.x == 0) status[w] = sBadT ? 1 : (sBadQ ? 2 : 0);
}

// One workgroup per (filter, walker): trapezoid of spectrum * weights over the
// filter's window of the wavenumber grid.
__global__ __launch_bounds__(64) void step_bandflux(int F, int Wfull, int solution, double rprs,
                                                    int ebalance, double e_in, double e_fac,
                                                    const int *idx0, const int *npts,
                                                    const int *woff, const double *nif,
                                                    const double *star, const double *wn,
                                                    const double *spec, int *status,
                                                    double *band) {
  const int f = blockIdx.x, w = blockIdx.y;
  const double *sp = spec + (size_t)w * Wfull;
  if (status[w] != 0 && status[w] != 3) {
    if (threadIdx.x == 0) band[(size_t)w * F + f] = -1.0;
    return;
  }
  if (ebalance) {
    // BARTfunc.py:377: e_out = trapz(spectrum, specwn) * 4 (Rp*100)^2
    double s = 0.0;
    for (int j = threadIdx.x; j + 1 < Wfull; j += 64)
      s += 0.5 * (sp[j] + sp[j + 1]) * (wn[j + 1] - wn[j]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (s * e_fac > e_in) {
      if (threadIdx.x == 0) { band[(size_t)w * F + f] = -1.0; status[w] = 3; }
      return;
    }
  }
  const int i0 = idx0[f], n = npts[f], o = woff[f];
  double s = 0.0;
  for (int j = threadIdx.x; j + 1 < n; j += 64) {
    double y0 = sp[i0 + j] * nif[o + j], y1 = sp[i0 + j + 1] * nif[o + j + 1];
    if (solution == 0) {
      // BARTfunc.py:388: (spectrum / istarfl) * rprs^2, then times the filter
      y0 = (sp[i0 + j] / star[o + j]) * rprs * rprs * nif[o + j];
      y1 = (sp[i0 + j + 1] / star[o + j + 1]) * rprs * rprs * nif[o + j + 1];
    }
    s += (wn[i0 + j + 1] - wn[i0 + j]) * (y0 + y1) / 2.0;
  }
  for (int o2 = 32; o2 > 0; o2 >>= 1) s += __shfl_xor(s, o2);
  if (threadIdx.x == 0) band[(size_t)w * F + f] = s;
}

// ---------------------------------------------------------------------------
StepArgs::~StepArgs() {
  auto fr = [](void *p) { if (p) (void)hipFree(p); };
  fr(d_abund); fr(d_ratio); fr(d_pbar); fr(d_imol); fr(d_metal); fr(d_idx0);
  fr(d_npts); fr(d_woff); fr(d_gw); fr(d_nifilter); fr(d_istarfl); fr(d_params);
  fr(d_prof); fr(d_spec); fr(d_band); fr(d_status);
}

template <class T>
static T *up(const T *h, size_t n) {
  T *d = nullptr;
  HIPCHK(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)));
  if (n) HIPCHK(hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

void step_setup(Engine &e, const double *ptargs5, int tint_thorngren, int pttype,
                double tmin, double tmax, const double *abund, int nmolfit,
                const int *imol, int nfilters, const int *idx0, const int *npts,
                const double *nifilter, const double *istarfl, double rprs, int solution) {
  static const int npt_of[] = {5, 1, 5, 6, 3, 8};
  if (pttype < 0 || pttype > PT_PIETTE) throw IoError{"step_setup: unknown PT model"};
  if (!abund || nmolfit < 0 || nfilters < 0) throw IoError{"step_setup: bad arguments"};
  delete e.step;
  e.step = nullptr;
  StepArgs *s = new StepArgs();
  e.step = s;
  s->pttype = pttype;
  s->nPT = npt_of[pttype];
  if (ptargs5) std::memcpy(s->ptargs, ptargs5, sizeof(double) * 5);
  s->tint_thorngren = tint_thorngren;
  s->tmin = tmin; s->tmax = tmax;
  s->nmolfit = nmolfit; s->nfilters = nfilters; s->solution = solution; s->rprs = rprs;
  s->iH2 = e.iH2; s->iHe = e.iHe;
  const int L = e.L, S = e.S;
  for (int i = 0; i < nmolfit; i++)
    if (imol[i] < 0 || imol[i] >= S) throw IoError{"step_setup: imol out of range"};
  std::vector<double> ratio(L, 1.0), pbar(L);
  std::vector<unsigned char> metal(S);
  for (int sidx = 0; sidx < S; sidx++) {
    const std::string &n = e.atm.species[sidx];
    metal[sidx] = !(n == "He" || n == "H2" || n == "H-" || n == "e-");  // BARTfunc.py:196-197
  }
  for (int l = 0; l < L; l++) {
    pbar[l] = e.atm.press[l] / 1e6;
    if (e.iH2 >= 0 && e.iHe >= 0) ratio[l] = abund[(size_t)l * S + e.iH2] / abund[(size_t)l * S + e.iHe];
  }
  // Gaussian smoothing weights exactly as scipy.ndimage.gaussian_filter1d
  // builds them: radius int(4 sigma + 0.5), exp(-x^2 / (2 sigma^2)) / sum
  double sigma = 0.0;
  if (pttype == PT_MADHU_NOINV || pttype == PT_MADHU_INV) sigma = 4.0;                 // PT.py:377,583
  if (pttype == PT_PIETTE && L > 1)                                                    // PT.py:811
    sigma = 0.3 / std::fabs(std::log10(pbar[L - 1]) - std::log10(pbar[L - 2]));
  std::vector<double> gw(1, 1.0);
  s->grad = 0;
  if (sigma > 0) {
    const int r = (int)(4.0 * sigma + 0.5);
    gw.assign(2 * r + 1, 0.0);
    double sum = 0.0;
    for (int x = -r; x <= r; x++) { gw[x + r] = std::exp(-0.5 / (sigma * sigma) * x * x); sum += gw[x + r]; }
    for (auto &g : gw) g /= sum;
    s->grad = r;
  }
  s->d_gw = up(gw.data(), gw.size());
  s->ptop = pbar[L - 1]; s->pbot = pbar[0];
  for (int l = 0; l < L; l++) { s->ptop = std::min(s->ptop, pbar[l]); s->pbot = std::max(s->pbot, pbar[l]); }
  {
    // np.argmin(|p - v|) on the top -> bottom array the reference works on
    const double targets[6] = {0.01, 0.1, 1.0, 3.2, 10.0, 32.0};
    auto nearest = [&](double v) {
      int best = L - 1;
      for (int l = L - 1; l >= 0; l--)
        if (std::fabs(pbar[l] - v) < std::fabs(pbar[best] - v)) best = l;
      return best;
    };
    int itop = L - 1, ibot = 0;
    for (int l = L - 1; l >= 0; l--) { if (pbar[l] < pbar[itop]) itop = l; }
    for (int l = L - 1; l >= 0; l--) { if (pbar[l] > pbar[ibot]) ibot = l; }
    s->pnode[0] = itop; s->pnode[7] = ibot;
    for (int j = 0; j < 6; j++) s->pnode[1 + j] = nearest(targets[j]);
  }
  s->d_abund = up(abund, (size_t)L * S);
  s->d_ratio = up(ratio.data(), L);
  s->d_pbar = up(pbar.data(), L);
  s->d_imol = up(imol, nmolfit);
  s->d_metal = up(metal.data(), S);
  std::vector<int> woff(nfilters);
  int tot = 0;
  for (int f = 0; f < nfilters; f++) {
    if (idx0[f] < 0 || npts[f] < 0 || idx0[f] + npts[f] > e.Wfull)
      throw IoError{"step_setup: filter window outside the wavenumber grid"};
    woff[f] = tot;
    tot += npts[f];
  }
  s->nwin = tot;
  s->d_idx0 = up(idx0, nfilters);
  s->d_npts = up(npts, nfilters);
  s->d_woff = up(woff.data(), nfilters);
  s->d_nifilter = up(nifilter, tot);
  std::vector<double> ones;
  if (!istarfl) { ones.assign(tot, 1.0); istarfl = ones.data(); }
  s->d_istarfl = up(istarfl, tot);
}

void step_set_ebalance(Engine &e, int on, double e_in, double e_fac) {
  if (!e.step) throw IoError{"step_set_ebalance: call step_setup first"};
  e.step->ebalance = on; e.step->e_in = e_in; e.step->e_fac = e_fac;
}

void step_ensure(Engine &e, int n) {
  StepArgs *s = e.step;
  if (n <= s->cap) return;
  int cap = std::max(n, 2 * s->cap);
  HIPCHK(hipDeviceSynchronize());
  auto re = [&](auto *&p, size_t count) {
    if (p) HIPCHK(hipFree(p));
    p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(*p)));
  };
  re(s->d_params, (size_t)cap * 64);
  re(s->d_prof, (size_t)cap * (e.S + 1) * e.L);
  re(s->d_spec, (size_t)cap * e.Wfull);
  re(s->d_band, (size_t)cap * std::max(s->nfilters, 1));
  re(s->d_status, (size_t)cap);
  s->cap = cap;
  e.ensure_walkers(cap);
}

void step_profiles_dev(Engine &e, const double *d_params, int n, int npars, double *d_prof,
                       int *d_status, hipStream_t st) {
  StepArgs *s = e.step;
  if (npars != s->nPT + s->nmolfit)
    throw IoError{"step: npars must be nPT + nmolfit (cloud/scattering/radius parameters are not batched yet)"};
  if (n <= 0) return;
  StepDev a;
  a.L = e.L; a.S = e.S; a.nPT = s->nPT; a.pttype = s->pttype; a.nmolfit = s->nmolfit;
  a.npars = npars; a.tint_thorngren = s->tint_thorngren; a.iH2 = s->iH2; a.iHe = s->iHe;
  a.rstar = s->ptargs[0]; a.tstar = s->ptargs[1]; a.tint = s->ptargs[2];
  a.sma = s->ptargs[3]; a.grav = s->ptargs[4];
  a.tmin = s->tmin; a.tmax = s->tmax;
  a.abund = s->d_abund; a.ratio = s->d_ratio; a.pbar = s->d_pbar;
  a.imol = s->d_imol; a.metal = s->d_metal;
  a.gw = s->d_gw; a.grad = s->grad; a.ptop = s->ptop; a.pbot = s->pbot;
  for (int j = 0; j < 8; j++) a.pnode[j] = s->pnode[j];
  hipLaunchKernelGGL(step_profiles, dim3(n), dim3(128), sizeof(double) * e.L, st, a, d_params,
                     d_prof, d_status);
  HIPCHK(hipGetLastError());
}

void step_bandflux_dev(Engine &e, const double *d_spec_full, int n, int *d_status,
                       double *d_bandflux, hipStream_t st) {
  StepArgs *s = e.step;
  if (n <= 0 || s->nfilters <= 0) return;
  hipLaunchKernelGGL(step_bandflux, dim3(s->nfilters, n), dim3(64), 0, st, s->nfilters, e.Wfull,
                     s->solution, s->rprs, s->ebalance, s->e_in, s->e_fac, s->d_idx0,
                     s->d_npts, s->d_woff, s->d_nifilter, s->d_istarfl, e.d_wn_full, d_spec_full,
                     d_status, d_bandflux);
  HIPCHK(hipGetLastError());
}

void step_run_dev(Engine &e, const double *d_params, int n, int npars, double *d_bandflux,
                  int *d_status, double *d_spec, hipStream_t st) {
  if (e.lo != 0 || e.hi != e.Wfull)
    throw IoError{"step_batch needs the full wavenumber grid on this GPU; sharded runs use "
                  "step_profiles / run_transit_batch_dev / all-gather / step_bandflux"};
  step_ensure(e, n);
  StepArgs *s = e.step;
  double *spec = d_spec ? d_spec : s->d_spec;
  int *status = d_status ? d_status : s->d_status;
  step_profiles_dev(e, d_params, n, npars, s->d_prof, status, st);
  e.run_dev(s->d_prof, n, spec, nullptr, st, false);
  step_bandflux_dev(e, spec, n, status, d_bandflux, st);
}

void step_run_host(Engine &e, const double *params, int n, int npars, double *bandflux,
                   int *status) {
  if (n <= 0) return;
  if (npars > 64) throw IoError{"step_batch: too many parameters"};
  step_ensure(e, n);
  StepArgs *s = e.step;
  HIPCHK(hipMemcpyAsync(s->d_params, params, sizeof(double) * (size_t)n * npars,
                        hipMemcpyHostToDevice, e.stream));
  step_run_dev(e, s->d_params, n, npars, s->d_band, s->d_status, nullptr, e.stream);
  HIPCHK(hipMemcpyAsync(bandflux, s->d_band, sizeof(double) * (size_t)n * s->nfilters,
                        hipMemcpyDeviceToHost, e.stream));
  if (status)
    HIPCHK(hipMemcpyAsync(status, s->d_status, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost,
                          e.stream));
  HIPCHK(hipStreamSynchronize(e.stream));
}

}  // namespace bartrt
