// Input/output converters of the per-step callable (reference
// code/BARTfunc.py:309-399), batched over walkers on the device.
#pragma once
#include "engine.hpp"

namespace bartrt {

enum { PT_LINE = 0, PT_ISO = 1, PT_MADHU_NOINV = 2, PT_MADHU_INV = 3, PT_ADIABATIC = 4, PT_PIETTE = 5 };

struct StepArgs {
  int pttype = PT_LINE, nPT = 5;
  double ptargs[5] = {0, 0, 0, 0, 0};  // R_star[m], T_star[K], T_int[K], sma[m], g[cm s-2]
  int tint_thorngren = 0;
  double tmin = 400, tmax = 3000;
  int nmolfit = 0, nfilters = 0, solution = 0;
  int nrad = 0, ncloud = 0, nray = 0;   // radius / cloud-top / scattering parameters after the T(p) ones
  double rprs = 0;
  int ebalance = 0;
  double e_in = 0, e_fac = 0;  // reject when trapz(spec) * e_fac > e_in
  int iH2 = -1, iHe = -1;
  int nwin = 0;                // total filter samples
  int grad = 0;                // smoothing radius (0 = none)
  double ptop = 0, pbot = 0;
  int pnode[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double *d_gw = nullptr;
  // device
  // abund[L][S] base abundances, ratio[L] H2/He of the base abundances, pbar[L]
  // pressure in bar (atm order), in one block (step_profiles stages it as it lies)
  double *d_consts = nullptr;
  int imol[16] = {0};                // [nmolfit] species index of each fitted molecule
  unsigned long long metal_mask = 0; // bit s: species s is a metal
  int *d_idx0 = nullptr, *d_npts = nullptr, *d_woff = nullptr;  // [F]
  // [nwin] per window sample: filter weight (x rprs^2 / stellar flux for eclipse)
  double *d_gwt = nullptr;
  // workspaces
  int cap = 0;
  double *d_prof = nullptr, *d_spec = nullptr;
  double *d_over = nullptr;             // [cap][3] per-walker overrides for prep (unfused path)
  // carry-over of the reference's worker (BARTfunc.py:318-324): a T(p) model that raises
  // ValueError leaves the chain's previous temperature profile in place.  Off: such a
  // walker is rejected.  On: walker w of a call is chain w; its last generated profile
  // is kept here ([cap][L], zeros before the first one, like the reference's array)
  int carry = 0;
  double *d_prevT = nullptr;
  int *d_status = nullptr;
  ~StepArgs();
};

void step_setup(Engine &e, const double *ptargs5, int tint_thorngren, int pttype,
                double tmin, double tmax, const double *abund, int nmolfit,
                const int *imol, int nfilters, const int *idx0, const int *npts,
                const double *nifilter, const double *istarfl, double rprs, int solution);
void step_set_ebalance(Engine &e, int on, double e_in, double e_fac);
void step_set_extras(Engine &e, int nrad, int ncloud, int nray);
void step_set_carry(Engine &e, int on);
void step_ensure(Engine &e, int n);
// params[n][npars] -> prof[n][(S+1)][L], status[n]
void step_profiles_dev(Engine &e, const double *d_params, int n, int npars, double *d_prof,
                       int *d_status, hipStream_t st);
// full-grid spectra [n][Wfull] -> bandflux[n][F]; may flip status to 3 (energy)
// status_out (optional, device-accessible): final status per walker
void step_bandflux_dev(Engine &e, const double *d_spec_full, int n, int *d_status,
                       double *d_bandflux, hipStream_t st, int *status_out = nullptr);
void step_run_dev(Engine &e, const double *d_params, int n, int npars, double *d_bandflux,
                  int *d_status, double *d_spec, hipStream_t st, int *status_out = nullptr);
void step_run_host(Engine &e, const double *params, int n, int npars, double *bandflux,
                   int *status);
// DEMC / snooker over all chains, one step_run_host per iteration (mcmc.hip)
void mcmc_run(Engine &e, int nchains, int npars, long nsteps, const double *params, const double *pmin,
              const double *pmax, const double *stepsize, int ndata, const double *data,
              const double *uncert, int snooker, unsigned long long seed, double *chain,
              double *chisq, long *naccept, long *nbad);

}  // namespace bartrt
