/*
 * rt_oracle.h -- CPU restatement (fp64, scalar C) of the forward
 * radiative-transfer path that BART's per-step callable reaches through
 * `trm.run_transit(profiles.flatten(), nwave)` (reference call site:
 * code/BARTfunc.py:363).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED: the reference implementation of this path lives in the
 * un-vendored git submodule exosports/transit (.gitmodules:8-10; the
 * directory modules/transit is empty in /root/reference and the pinned
 * commit is unrecoverable), and the reference holds no test, fixture or
 * golden spectrum for it.  This file restates the algorithm from the
 * in-tree evidence (cited per function) plus the published description of
 * Transit; every convention that could not be checked against source is a
 * named switch in rt_oracle_cfg (see DESIGN.md "Conventions").  It is pinned
 * only by the analytic known-answer tests in tests/test_oracle_kat.py.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Physical constants, cgs.  H, LS, KB are the values BART copies from
 * transit's constants header (code/constants.py:13-16).  AMU and AMAGAT are
 * the CODATA-2002 / standard values transit is recalled to use (unverified). */
#define ORC_H      6.6260755e-27
#define ORC_LS     2.99792458e10
#define ORC_KB     1.380658e-16
#define ORC_AMU    1.66053886e-24
#define ORC_AMAGAT 2.68679e19
#define ORC_PI     3.141592653589793

/* intensity integration rule (rt_oracle.c, column_eclipse): 0 (default)
 * trapezoid in the transmittance exp(-tau/mu); 1 the Simpson / trapezoid hybrid
 * of SURVEY.md App. A-4 for tau (over radius) and for B exp(-tau/mu) (over tau,
 * zero-padded past `last`); 2 plain trapezoid in tau of B exp(-tau/mu).
 * The product carries the same switch (cfg key `integ`, BARTRT_INTEG). */
enum { ORC_INTEG_TRAPZ = 0, ORC_INTEG_SIMPSON = 1, ORC_INTEG_TRAPZ_TAU = 2 };
enum { ORC_SOL_ECLIPSE = 0, ORC_SOL_TRANSIT = 1 };

typedef struct {
  /* sizes */
  int nlayers;              /* L */
  int nspecies;             /* S: rows 1..S of the profile array */
  int nmol;                 /* M: molecules in the opacity table */
  int ntemp;                /* Nt: table temperatures */
  int nwave;                /* W */
  int nangles;              /* A: raygrid entries */
  int ncia;                 /* number of CIA (cross-section) tables */
  int integ;                /* ORC_INTEG_* */
  int solution;             /* ORC_SOL_* */
  int scat_flag;            /* 0 none, 1 scaled lambda^-4, 2 polarisability */
  int has_cloud;            /* grey opaque deck below cloudtop */
  int cut_slant;            /* the `toomuch` cut (column_eclipse): 0 on the vertical optical depth, one `last`
                             * for every ray angle; 1 on each ray's SLANT depth tau / mu, its own `last` per angle */
  /* atmosphere (atm-file order: layer 0 = bottom, L-1 = top) */
  const double *press;      /* [L] barye */
  const double *mass;       /* [S] amu, species order of the atm file */
  /* opacity grid, layout of the --opacityfile: o[L][Nt][M][W], cm2/g */
  const int    *opmol;      /* [M] index into species for each table molecule */
  const double *tgrid;      /* [Nt] K, ascending */
  const double *kappa;      /* [L*Nt*M*W] */
  const double *wn;         /* [W] cm-1 ascending */
  /* CIA tables already resampled on wn: alpha[ncia][cia_nt[c]][W] concatenated */
  const int    *cia_s1;     /* [ncia] species index */
  const int    *cia_s2;     /* [ncia] */
  const int    *cia_nt;     /* [ncia] */
  const double *cia_temp;   /* concatenated temperatures */
  const double *cia_alpha;  /* concatenated [nt][W], cm-1 amagat-2 */
  /* geometry */
  const double *angles_deg; /* [A] */
  double toomuch;
  double gsurf;             /* cm s-2 at refradius */
  double refpress;          /* barye */
  double refradius;         /* cm */
  double cloudtop;          /* barye (used when has_cloud) */
  double scat_value;        /* log10 scale for scat_flag==1 */
  int    scat_iH2;          /* species index of H2 (or -1) */
  int    scat_iHe;          /* species index of He (or -1) */
  double starrad;           /* cm (transit geometry only) */
  const double *extra_ext;  /* optional [L][W] extinction added as is (line-by-line), or NULL */
  /* radius-ramp cloud (cfg cloudrad / cloudfct / cloudext, code/makecfg.py:46-47):
   * grey extinction 0 above cloud_rup, linear up to cloud_ext at cloud_rdown,
   * cloud_ext below; cloud_ext = 0: none.  Unverified against transit's source. */
  double cloud_rup, cloud_rdown, cloud_ext;
  int transparent;          /* transit geometry: rays below the last chord keep its transmission */
  int cia_spline;           /* CIA interpolation (cfg `cia_interp`): 0 linear in wavenumber (at init)
                             * and temperature; 1 natural cubic splines in both.  Unverified either way. */
  const double *cia_y2;     /* cia_spline: second derivatives in T of cia_alpha, same layout; else NULL */
} rt_oracle_cfg;

/* Hydrostatic radii.  Follows code/makeatm.py:183-263 (radpress), in cgs and
 * for layers given bottom->top.  temp[L], mu[L] (amu), press[L] barye -> rad[L] cm. */
void orc_radpress(int n, const double *press, const double *temp,
                  const double *mu, double p0, double r0, double g0,
                  double *rad);

/* Planck function per wavenumber, erg s-1 sr-1 cm-2 cm.  code/cf.py:108-109. */
double orc_planck(double wn, double temp);

/* Mean molecular mass per layer: sum_s q_s * mass_s (code/makeatm.py:503-506). */
void orc_meanmass(int L, int S, const double *q /*[S][L]*/, const double *mass,
                  double *mu);

/* Total extinction e[L][W] (cm-1) for one atmosphere.  prof = [(S+1)][L]. */
int orc_extinction(const rt_oracle_cfg *c, const double *prof, double *ext,
                   double *rad_out /* [L] or NULL */);

/* One forward spectrum.  prof = [(S+1)*L] exactly as BARTfunc.py:213-222,363
 * builds it.  spec[W].  Optional: tau[W][L] (index 0 = top layer, the
 * tau.dat convention of code/cf.py:68-94,123-131), last[W]. Returns 0. */
int orc_run_transit(const rt_oracle_cfg *c, const double *prof, double *spec,
                    double *tau_out, int *last_out);

/* Intensities per angle, [A][W] (outintens), same call otherwise. */
int orc_intensity(const rt_oracle_cfg *c, const double *prof, double *intens);

#ifdef __cplusplus
}
#endif
#endif
