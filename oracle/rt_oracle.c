/*
 * rt_oracle.c -- scalar fp64 CPU restatement of the forward RT path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see rt_oracle.h).
 *
 * Reference evidence followed, function by function:
 *   - profile array layout ............ code/BARTfunc.py:213-222,363
 *   - hydrostatic radii ............... code/makeatm.py:183-263
 *   - mean molecular mass ............. code/makeatm.py:503-506
 *   - Planck function & constants ..... code/cf.py:108-109, code/constants.py:13-16
 *   - tau indexing (0 = top) .......... code/cf.py:68-94,123-131
 *   - opacity grid (tlow/thigh/tempdelt, same layers and wn as the run)
 *                                       examples/demo/BART_eclipse.cfg:142-145,
 *                                       doc/BART_user_manual/BART_user_manual.tex:771-775
 *   - raygrid / toomuch ............... examples/demo/BART_eclipse.cfg:135-136
 *   - output units (erg s-1 cm-2 cm) .. code/BARTfunc.py:375-377, code/wine.py:121-122
 * Everything else (table interpolation rule, CIA scaling, integration rule,
 * angle quadrature, bottom boundary) restates the published Transit
 * algorithm (Cubillos et al. 2022, PSJ 3, 81; Blecic et al. 2022, PSJ 3, 82)
 * and is unverified against source.
 */
#include "rt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
double orc_planck(double wn, double temp) {
  /* cf.py:108-109:  2 h wn^3 c^2 / (exp(h wn c / (k T)) - 1) */
  return (2.0 * ORC_H * wn * wn * wn * ORC_LS * ORC_LS) /
         (exp((ORC_H * wn * ORC_LS) / (ORC_KB * temp)) - 1.0);
}

void orc_meanmass(int L, int S, const double *q, const double *mass, double *mu) {
  for (int l = 0; l < L; l++) {
    double m = 0.0;
    for (int s = 0; s < S; s++) m += q[(size_t)s * L + l] * mass[s];
    mu[l] = m;
  }
}

/* makeatm.py:183-263, restated for layers ordered bottom->top (index 0 has
 * the highest pressure; this is the orientation the reference flips to at
 * makeatm.py:225-227) and cgs units: KB/AMU replaces Avogadro*k because mu
 * is in amu here. */
void orc_radpress(int n, const double *press, const double *temp,
                  const double *mu, double p0, double r0, double g0,
                  double *rad) {
  double *g = (double *)malloc(sizeof(double) * n);
  const double rgas = ORC_KB / ORC_AMU;
  /* closest layer to p0 (np.argmin(|press - p0|), first minimum) */
  int idx = 0;
  double best = fabs(press[0] - p0);
  for (int i = 1; i < n; i++) {
    double d = fabs(press[i] - p0);
    if (d < best) { best = d; idx = i; }
  }
  if (press[idx] != p0) {
    /* temp and mu at p0 by linear interpolation in log10(p)
     * (makeatm.py:214-220) */
    double lp0 = log10(p0), t0 = temp[idx], m0 = mu[idx];
    for (int i = 0; i + 1 < n; i++) {
      double la = log10(press[i]), lb = log10(press[i + 1]);
      if ((lp0 <= la && lp0 >= lb) || (lp0 >= la && lp0 <= lb)) {
        double f = (lp0 - la) / (lb - la);
        t0 = temp[i] + f * (temp[i + 1] - temp[i]);
        m0 = mu[i] + f * (mu[i + 1] - mu[i]);
        break;
      }
    }
    /* makeatm.py:236-243; both branches are the same expression */
    rad[idx] = r0 + 0.5 * (temp[idx] / mu[idx] + t0 / m0) *
                        (rgas * log(p0 / press[idx]) / g0);
    g[idx] = g0 * r0 * r0 / (rad[idx] * rad[idx]);
  } else {
    rad[idx] = r0;
    g[idx] = g0;
  }
  /* below p0 (higher pressure): makeatm.py:250-253 */
  for (int i = idx - 1; i >= 0; i--) {
    rad[i] = rad[i + 1] - 0.5 * (temp[i] / mu[i] + temp[i + 1] / mu[i + 1]) *
                              (rgas * log(press[i] / press[i + 1]) / g[i + 1]);
    g[i] = g[i + 1] * rad[i + 1] * rad[i + 1] / (rad[i] * rad[i]);
  }
  /* above p0: makeatm.py:254-258 */
  for (int i = idx + 1; i < n; i++) {
    rad[i] = rad[i - 1] + 0.5 * (temp[i] / mu[i] + temp[i - 1] / mu[i - 1]) *
                              (rgas * log(press[i - 1] / press[i]) / g[i - 1]);
    g[i] = g[i - 1] * rad[i - 1] * rad[i - 1] / (rad[i] * rad[i]);
  }
  free(g);
}

/* bracket T in grid: largest j with grid[j] <= T, clamped to [0, n-2] */
static int bracket(const double *grid, int n, double t) {
  int j = 0;
  while (j < n - 2 && grid[j + 1] <= t) j++;
  return j;
}

/* Rayleigh cross sections (cm2 per molecule).  Unverified conventions:
 *  flag 1: 10^value * sigma0 * (wn*lambda0)^4 per H2 molecule with
 *          sigma0 = 2.52e-28 cm2 at lambda0 = 750 nm (Lecavelier des Etangs
 *          et al. 2008);
 *  flag 2: (128 pi^5/3) alpha^2 wn^4 for H2 and He polarisabilities. */
static const double RAY_SIGMA0 = 2.52e-28, RAY_LAMBDA0 = 7.5e-5;
static const double POL_H2 = 0.8059e-24, POL_HE = 0.2051e-24;

int orc_extinction(const rt_oracle_cfg *c, const double *prof, double *ext,
                   double *rad_out) {
  const int L = c->nlayers, S = c->nspecies, M = c->nmol, W = c->nwave,
            Nt = c->ntemp;
  const double *temp = prof;
  const double *q = prof + L;
  double *mu = (double *)malloc(sizeof(double) * L);
  double *rad = (double *)malloc(sizeof(double) * L);
  orc_meanmass(L, S, q, c->mass, mu);
  orc_radpress(L, c->press, temp, mu, c->refpress, c->refradius, c->gsurf, rad);
  if (rad_out) memcpy(rad_out, rad, sizeof(double) * L);

  for (int l = 0; l < L; l++) {
    const double T = temp[l];
    const double nd = c->press[l] / (ORC_KB * T); /* molecules cm-3 */
    double *e = ext + (size_t)l * W;
    double grey = 0.0;
    if (c->cloud_ext > 0.0) {
      double r = rad[l];
      grey = r >= c->cloud_rup ? 0.0
             : (r <= c->cloud_rdown ? c->cloud_ext
                                    : c->cloud_ext * (c->cloud_rup - r) / (c->cloud_rup - c->cloud_rdown));
    }
    for (int i = 0; i < W; i++) e[i] = grey + (c->extra_ext ? c->extra_ext[(size_t)l * W + i] : 0.0);
    /* molecular extinction: linear-in-T between the two bracketing planes
     * of the layer's own slab of the grid, times the molecule's mass density */
    if (M > 0) {
      int j = bracket(c->tgrid, Nt, T);
      double f = (T - c->tgrid[j]) / (c->tgrid[j + 1] - c->tgrid[j]);
      for (int m = 0; m < M; m++) {
        int s = c->opmol[m];
        double rho = q[(size_t)s * L + l] * c->mass[s] * ORC_AMU * nd;
        const double *klo = c->kappa + (((size_t)l * Nt + j) * M + m) * W;
        const double *khi = c->kappa + (((size_t)l * Nt + j + 1) * M + m) * W;
        double wlo = rho * (1.0 - f), whi = rho * f;
        for (int i = 0; i < W; i++) e[i] += wlo * klo[i] + whi * khi[i];
      }
    }
    /* CIA: alpha(T, wn) * n1/amagat * n2/amagat; T clamped to the file range */
    size_t toff = 0, aoff = 0;
    for (int k = 0; k < c->ncia; k++) {
      int nt = c->cia_nt[k];
      const double *tg = c->cia_temp + toff;
      const double *al = c->cia_alpha + aoff;
      double Tc = T < tg[0] ? tg[0] : (T > tg[nt - 1] ? tg[nt - 1] : T);
      double n1 = q[(size_t)c->cia_s1[k] * L + l] * nd / ORC_AMAGAT;
      double n2 = q[(size_t)c->cia_s2[k] * L + l] * nd / ORC_AMAGAT;
      if (nt == 1) {
        double w = n1 * n2;
        for (int i = 0; i < W; i++) e[i] += w * al[i];
      } else {
        int j = bracket(tg, nt, Tc);
        double f = (Tc - tg[j]) / (tg[j + 1] - tg[j]);
        double wlo = n1 * n2 * (1.0 - f), whi = n1 * n2 * f;
        const double *alo = al + (size_t)j * W, *ahi = al + (size_t)(j + 1) * W;
        for (int i = 0; i < W; i++) e[i] += wlo * alo[i] + whi * ahi[i];
        if (c->cia_spline && c->cia_y2) {
          /* natural cubic spline through the file's temperatures (Numerical Recipes' splint):
           * + ((a^3 - a) y2_j + (b^3 - b) y2_j+1) h^2 / 6, a = 1 - f, b = f */
          double h = tg[j + 1] - tg[j], sa = 1.0 - f, sb = f;
          double clo = n1 * n2 * (sa * sa * sa - sa) * h * h / 6.0, chi = n1 * n2 * (sb * sb * sb - sb) * h * h / 6.0;
          const double *ylo = c->cia_y2 + aoff + (size_t)j * W, *yhi = c->cia_y2 + aoff + (size_t)(j + 1) * W;
          for (int i = 0; i < W; i++) e[i] += clo * ylo[i] + chi * yhi[i];
        }
      }
      toff += nt;
      aoff += (size_t)nt * W;
    }
    /* Rayleigh */
    if (c->scat_flag == 1 && c->scat_iH2 >= 0) {
      double nh2 = q[(size_t)c->scat_iH2 * L + l] * nd;
      double a = pow(10.0, c->scat_value) * RAY_SIGMA0 * nh2;
      for (int i = 0; i < W; i++) {
        double x = c->wn[i] * RAY_LAMBDA0;
        e[i] += a * (x * x) * (x * x);
      }
    } else if (c->scat_flag == 2) {
      double k0 = 128.0 * pow(ORC_PI, 5) / 3.0, a = 0.0;
      if (c->scat_iH2 >= 0) a += POL_H2 * POL_H2 * q[(size_t)c->scat_iH2 * L + l] * nd;
      if (c->scat_iHe >= 0) a += POL_HE * POL_HE * q[(size_t)c->scat_iHe * L + l] * nd;
      a *= k0;
      for (int i = 0; i < W; i++) {
        double x = c->wn[i];
        e[i] += a * (x * x) * (x * x);
      }
    }
  }
  free(mu);
  free(rad);
  return 0;
}

/* One Simpson panel over the three points (x0, x0 + h0, x0 + h0 + h1); with a
 * zero-width half it degenerates to the trapezoids of its two intervals. */
static double simpson_panel(double h0, double h1, double y0, double y1, double y2) {
  if (h0 == 0.0 || h1 == 0.0)
    return 0.5 * h0 * (y0 + y1) + 0.5 * h1 * (y1 + y2);
  double hs = h0 + h1;
  return hs / 6.0 * (y0 * (2.0 - h1 / h0) + y1 * hs * hs / (h0 * h1) + y2 * (2.0 - h0 / h1));
}

/* Simpson / trapezoid hybrid on n points with interval widths h[0..n-2]
 * (h[i] = x[i+1] - x[i]): an odd number of points is covered by Simpson
 * panels; with an even number the FIRST interval is taken by a trapezoid and
 * the panels cover the rest (the integrator recalled for
 * Transit's tau and intensity integrals; unverified against source). */
static double simps_hybrid(const double *h, const double *y, int n) {
  if (n < 2) return 0.0;
  double res = 0.0;
  int start = 0;
  if ((n & 1) == 0) {
    res += 0.5 * h[0] * (y[0] + y[1]);
    start = 1;
  }
  for (int j = start; j + 2 <= n - 1; j += 2)
    res += simpson_panel(h[j], h[j + 1], y[j], y[j + 1], y[j + 2]);
  return res;
}

/* Per-wavenumber column solve.  e_col[k], r_col[k], t_col[k] are indexed
 * from the TOP layer (k = 0) downwards.  Returns intensities per angle and
 * tau[k]; *last_out = index of the deepest layer included.
 *
 * (The wrapper's default is rule 1, ORC_INTEG_SIMPSON, and the slant cut -- oracle/rt_oracle.py OracleEngine --
 * the product's defaults since round 4.)
 * integ = ORC_INTEG_TRAPZ (0):
 *   tau by trapezoid in radius from the top; I = int B d(exp(-tau/mu)),
 *   trapezoid in the transmittance.
 * integ = ORC_INTEG_SIMPSON (1), App. A-4 as recalled:
 *   tau[k] = integral of the extinction from layer k up to the top by the
 *   hybrid above, the points taken FROM LAYER k UPWARDS: with an even number of
 *   points the trapezoid covers the interval next to layer k and the Simpson
 *   panels end at the top, i.e. panels (0,1,2), (2,3,4), ... in this file's
 *   top-down index, plus the trapezoid of (k-1, k) for odd k.
 *   I(mu) = (1/mu) int B exp(-tau/mu) dtau by the same hybrid over the points
 *   0 .. last taken from the top, padded by ONE point of zero integrand one
 *   unit of tau beyond `last` when a layer exists there (n = min(last + 2, L)
 *   points; no padding when the column ends on the bottom layer or on a cloud
 *   deck reached below toomuch).
 * integ = ORC_INTEG_TRAPZ_TAU (2): tau as in 0; I = (1/mu) int B exp(-tau/mu) dtau
 *   by plain trapezoid over 0 .. last. */
static void column_eclipse(const rt_oracle_cfg *c, double wn, int L,
                           const double *e_col, const double *r_col,
                           const double *t_col, int kcloud, double *tau,
                           double *intens, int *last_out) {
  const int A = c->nangles;
  int last = L - 1;
  double *dr = (double *)malloc(sizeof(double) * L);   /* dr[k] = r[k-1] - r[k] */
  dr[0] = 0.0;
  for (int k = 1; k < L; k++) dr[k] = r_col[k - 1] - r_col[k];
  tau[0] = 0.0;
  int kend = (kcloud >= 0) ? kcloud : L - 1;
  double s_even = 0.0;   /* Simpson: tau at the last even index */
  for (int k = 1; k <= kend; k++) {
    if (c->integ == ORC_INTEG_SIMPSON) {
      if (k & 1) {
        tau[k] = s_even + 0.5 * (e_col[k - 1] + e_col[k]) * dr[k];
      } else {
        s_even += simpson_panel(dr[k - 1], dr[k], e_col[k - 2], e_col[k - 1], e_col[k]);
        tau[k] = s_even;
      }
    } else {
      tau[k] = tau[k - 1] + 0.5 * (e_col[k - 1] + e_col[k]) * dr[k];
    }
    if (tau[k] > c->toomuch) { last = k; break; }
    last = k;
  }
  if (kend == 0) last = 0;
  /* The `toomuch` cut (cfg `cut`).  vertical: the column ends on the
   * first layer whose vertical optical depth passes toomuch, for every ray angle.  slant (the default): App.
   * A-4 read literally -- "slant path ds = dr / cos(theta); tau accumulated from the top; the
   * loop stops where tau > toomuch" -- each ray ends on the first layer whose SLANT depth
   * tau / mu passes toomuch (its own `last`), and rule 1's padded point sits one unit of SLANT
   * depth further.  The optical depths themselves are the same numbers either way.  The slant depth is
   * formed as tau * (1 / mu), the product the kernels compare (RtArgs::invmu, slant_thresholds): tau / mu
   * may round the other way by one ulp on a layer that sits exactly on the cut. */
  int last_v = last;   /* (the slant cuts lie at or above the vertical one: mu <= 1) */
  for (int k = last + 1; k < L; k++) tau[k] = tau[last]; /* not computed deeper */
  double *f = (double *)malloc(sizeof(double) * (L + 1));
  double *bk = (double *)malloc(sizeof(double) * L);
  double *ek = (double *)malloc(sizeof(double) * L);
  double *hx = (double *)malloc(sizeof(double) * (L + 1));
  for (int k = 0; k <= last_v; k++) bk[k] = orc_planck(wn, t_col[k]);
  for (int a = 0; a < A; a++) {
    double mu = cos(c->angles_deg[a] * ORC_PI / 180.0);
    const double invmu = 1.0 / mu;
    int last = last_v;
    if (c->cut_slant) {
      last = kend;
      for (int k = 1; k <= kend; k++)
        if (tau[k] * invmu > c->toomuch) { last = k; break; }
      if (last > last_v) last = last_v;
    }
    const int deck = kcloud >= 0 && last == kcloud &&
                     !((c->cut_slant ? tau[last] * invmu : tau[last]) > c->toomuch);
    for (int k = 0; k <= last; k++) {
      ek[k] = exp(-tau[k] / mu);
      f[k] = bk[k] * ek[k];
    }
    double I;
    if (c->integ == ORC_INTEG_SIMPSON) {
      int n = last + 1;
      for (int k = 0; k < last; k++) hx[k] = tau[k + 1] - tau[k];
      if (!deck && last + 1 < L) { /* one padded point: integrand 0, one unit of (vertical / slant) tau further */
        hx[last] = c->cut_slant ? mu : 1.0;
        f[last + 1] = 0.0;
        n = last + 2;
      }
      I = simps_hybrid(hx, f, n) / mu;
    } else if (c->integ == ORC_INTEG_TRAPZ_TAU) {
      I = 0.0;
      for (int k = 1; k <= last; k++) I += 0.5 * (f[k - 1] + f[k]) * (tau[k] - tau[k - 1]);
      I /= mu;
    } else {
      /* I = int B d(exp(-tau/mu)), trapezoid in the transmittance.
       * Exact for an isothermal column, never exceeds the hottest layer's
       * Planck function (a trapezoid in tau does when tau jumps by >> 1
       * across one layer), and is the discretisation BART's own
       * contribution functions use (code/cf.py:123-131). */
      I = 0.0;
      for (int k = 1; k <= last; k++) I += 0.5 * (bk[k - 1] + bk[k]) * (ek[k - 1] - ek[k]);
    }
    /* opaque cloud deck reached before toomuch: it emits as a surface */
    if (deck) I += f[last];
    intens[a] = I;
  }
  last = last_v;
  free(bk);
  free(ek);
  free(f);
  free(hx);
  free(dr);
  *last_out = last;
}

/* Transit (transmission) geometry.  For the ray with impact parameter b = r_k
 * (each layer radius in turn, from the top) the optical depth along the chord is
 *   tau_k = 2 int_b^{r_top} e(r) ds,  s = sqrt(r^2 - b^2),
 * by trapezoid in s over the layers above; the chord with tau > toomuch and
 * everything below it is opaque.  Modulation (what BARTfunc compares with the
 * transit depths, examples/demo/BART_transit.cfg:45-47):
 *   M = (r_top^2 - 2 int_{r_last}^{r_top} exp(-tau(b)) b db) / R_star^2,
 * trapezoid in b.  With no absorber M = (r_bottom / R_star)^2: the lowest layer
 * is an opaque surface.  Published Transit algorithm; unverified vs source. */
static void column_transit(const rt_oracle_cfg *c, int L, const double *e_col,
                           const double *r_col, int kcloud, double *tau, double *mod,
                           int *last_out) {
  int kend = (kcloud >= 0) ? kcloud : L - 1;
  int last = kend;
  tau[0] = 0.0;
  for (int k = 1; k <= kend; k++) {
    double t = 0.0, sprev = sqrt((r_col[0] - r_col[k]) * (r_col[0] + r_col[k]));
    for (int j = 1; j <= k; j++) {
      double s = (j == k) ? 0.0 : sqrt((r_col[j] - r_col[k]) * (r_col[j] + r_col[k]));
      t += (e_col[j - 1] + e_col[j]) * (sprev - s);
      sprev = s;
    }
    tau[k] = t;
    if (t > c->toomuch) { last = k; break; }
  }
  for (int k = last + 1; k < L; k++) tau[k] = tau[last];
  double integ = 0.0, gprev = r_col[0];
  for (int k = 1; k <= last; k++) {
    double g = exp(-tau[k]) * r_col[k];
    integ += 0.5 * (gprev + g) * (r_col[k - 1] - r_col[k]);
    gprev = g;
  }
  /* below r_last the planet is opaque: subtract the transmitted light of the
   * annulus between r_last and the top only */
  double rl = r_col[last];
  double area = rl * rl + 2.0 * (0.5 * (r_col[0] * r_col[0] - rl * rl) - integ);
  /* `transparent`: no opaque core, the rays below the last chord keep its transmission */
  if (c->transparent) area -= exp(-tau[last]) * rl * rl;
  *mod = area / (c->starrad * c->starrad);
  *last_out = last;
}

static int solve_transit(const rt_oracle_cfg *c, const double *prof, double *spec,
                         double *tau_out, int *last_out) {
  const int L = c->nlayers, W = c->nwave;
  double *ext = (double *)malloc(sizeof(double) * (size_t)L * W);
  double *rad = (double *)malloc(sizeof(double) * L);
  orc_extinction(c, prof, ext, rad);
  double *e_col = (double *)malloc(sizeof(double) * L);
  double *r_col = (double *)malloc(sizeof(double) * L);
  double *tau = (double *)malloc(sizeof(double) * L);
  int kcloud = -1;
  for (int k = 0; k < L; k++) {
    r_col[k] = rad[L - 1 - k];
    if (c->has_cloud && kcloud < 0 && c->press[L - 1 - k] >= c->cloudtop) kcloud = k;
  }
  for (int i = 0; i < W; i++) {
    for (int k = 0; k < L; k++) e_col[k] = ext[(size_t)(L - 1 - k) * W + i];
    int last;
    column_transit(c, L, e_col, r_col, kcloud, tau, &spec[i], &last);
    if (tau_out) memcpy(tau_out + (size_t)i * L, tau, sizeof(double) * L);
    if (last_out) last_out[i] = last;
  }
  free(ext); free(rad); free(e_col); free(r_col); free(tau);
  return 0;
}

static int solve(const rt_oracle_cfg *c, const double *prof, double *spec,
                 double *tau_out, int *last_out, double *intens_out) {
  const int L = c->nlayers, W = c->nwave, A = c->nangles;
  double *ext = (double *)malloc(sizeof(double) * (size_t)L * W);
  double *rad = (double *)malloc(sizeof(double) * L);
  orc_extinction(c, prof, ext, rad);
  double *e_col = (double *)malloc(sizeof(double) * L);
  double *r_col = (double *)malloc(sizeof(double) * L);
  double *t_col = (double *)malloc(sizeof(double) * L);
  double *tau = (double *)malloc(sizeof(double) * L);
  double *intens = (double *)malloc(sizeof(double) * A);
  double *wgt = (double *)malloc(sizeof(double) * A);
  /* angle quadrature: pi * sum_a I_a (sin^2 hi - sin^2 lo), bin edges midway
   * between raygrid angles, 0 and 90 deg at the ends */
  for (int a = 0; a < A; a++) {
    double lo = (a == 0) ? 0.0 : 0.5 * (c->angles_deg[a - 1] + c->angles_deg[a]);
    double hi = (a == A - 1) ? 90.0 : 0.5 * (c->angles_deg[a] + c->angles_deg[a + 1]);
    double sl = sin(lo * ORC_PI / 180.0), sh = sin(hi * ORC_PI / 180.0);
    wgt[a] = ORC_PI * (sh * sh - sl * sl);
  }
  int kcloud = -1;
  for (int k = 0; k < L; k++) {
    r_col[k] = rad[L - 1 - k];
    t_col[k] = prof[L - 1 - k];
    if (c->has_cloud && kcloud < 0 && c->press[L - 1 - k] >= c->cloudtop) kcloud = k;
  }
  for (int i = 0; i < W; i++) {
    for (int k = 0; k < L; k++) e_col[k] = ext[(size_t)(L - 1 - k) * W + i];
    int last;
    column_eclipse(c, c->wn[i], L, e_col, r_col, t_col, kcloud, tau, intens, &last);
    if (spec) {
      double F = 0.0;
      for (int a = 0; a < A; a++) F += wgt[a] * intens[a];
      spec[i] = F;
    }
    if (intens_out)
      for (int a = 0; a < A; a++) intens_out[(size_t)a * W + i] = intens[a];
    if (tau_out) memcpy(tau_out + (size_t)i * L, tau, sizeof(double) * L);
    if (last_out) last_out[i] = last;
  }
  free(ext); free(rad); free(e_col); free(r_col); free(t_col);
  free(tau); free(intens); free(wgt);
  return 0;
}

int orc_run_transit(const rt_oracle_cfg *c, const double *prof, double *spec,
                    double *tau_out, int *last_out) {
  if (c->solution == ORC_SOL_TRANSIT) return solve_transit(c, prof, spec, tau_out, last_out);
  return solve(c, prof, spec, tau_out, last_out, NULL);
}

int orc_intensity(const rt_oracle_cfg *c, const double *prof, double *intens) {
  if (c->solution != ORC_SOL_ECLIPSE) return -1;
  return solve(c, prof, NULL, NULL, NULL, intens);
}
