// Line-by-line (Voigt) extinction from a TLI line list: the on-the-fly path
// the reference takes when no opacity file is configured
// (doc/BART_user_manual/BART_user_manual.tex:776-777) and the generator of the
// opacity grid (`transit --justOpacity`, BART.py:561-565).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "io.hpp"
#include "kernels.hpp"

namespace bartrt {

struct Engine;

constexpr int kMaxIso = 64;    // isotopes over all databases
constexpr int kMaxGroup = 16;  // databases (molecules) in the TLI

struct LblDev {
  // line arrays, all groups concatenated, ascending wavenumber inside a group
  const double *nu0, *elow, *gf;
  const int *liso;               // global isotope index per line
  int ngroup, niso;
  long gstart[kMaxGroup], gend[kMaxGroup];
  int gspecies[kMaxGroup];       // species index of the group's molecule
  double gdiam[kMaxGroup];       // collision diameter of the molecule, cm
  int iso_group[kMaxIso];
  double iso_mass[kMaxIso], iso_ratio[kMaxIso];
  int iso_zoff[kMaxIso], iso_nt[kMaxIso], iso_toff[kMaxIso];
  const double *ztab, *ztemp;    // concatenated partition functions / their temperatures
  double nwidth, ethresh;
  // Sampling (cfg `wnosamp`): a layer's line sums are evaluated dv times
  // finer than the output grid and reduced to it; dv = the smallest divisor of osamp
  // whose spacing wndelt / dv is at most half the layer's narrowest line half-width
  // (osamp_rule 1: dv = osamp everywhere).  Fine point k of the FULL grid sits at
  // wn_first + k * wndelt / dv; this engine's block starts at output index i_off.
  int osamp, osamp_rule, ndiv, i_off, wfull;
  int odiv[64];                  // divisors of osamp, ascending
  double wn_first, wndelt;
  // Voigt evaluation (cfg `voigt`; SURVEY.md App. A-5 as recalled): 0 = the
  // Faddeeva function per (line, point) at the line's own widths; 1 = WIDTH GRID: profiles are
  // tabulated once per state on a grid of ndop Doppler x nlor Lorentz half-widths (log spaced,
  // dgrid / lgrid), a line takes the profile of the nearest grid widths, centred on the sampling
  // point nearest to its centre and reaching floor(nwidth max(aD, aL) / step) points either side.
  int voigt_grid, ndop, nlor;
  int dspan;                     // most Doppler-grid indices the lines of one isotope can span at one state
  const double *dgrid, *lgrid;   // [ndop], [nlor] half-widths, cm-1
  double dop_ln0, dop_dln, lor_ln0, lor_dln;   // ln of the first grid width and ln spacing (0: one width)
  double nu_lo, nu_hi;           // range of the line centres kept (Doppler widths scale with the centre)
  // coarse index of the sorted lists: bucket[boff[g] + b] = first line of group g
  // with nu0 >= bmin + b * bstep (b = 0..nbucket; entry nbucket = gend[g])
  const long *bucket;
  int boff[kMaxGroup];
  int nbucket;
  double bmin, bstep;
};

struct Lbl {
  Tli tli;
  LblDev dev{};
  long nlines = 0;
  double *d_nu0 = nullptr, *d_elow = nullptr, *d_gf = nullptr, *d_ztab = nullptr, *d_ztemp = nullptr;
  int *d_liso = nullptr;
  long *d_bucket = nullptr;
  // per-call workspaces
  double *d_state = nullptr;     // [nstate][3*niso + 2]
  double *d_smax = nullptr;      // [nstate][ngroup]
  double *d_ext = nullptr;       // [nstate][W] (extinction mode)
  int *d_dvmax = nullptr;        // largest oversampling factor among the states of a call
  long cap_state = 0;
  // width-grid mode (LblDev::voigt_grid): the grids, and per call the profile tables
  double *d_dgrid = nullptr, *d_lgrid = nullptr;
  int *d_ginfo = nullptr;        // [nstate][niso][3]: Lorentz index, first Doppler index, Doppler indices used
  int *d_gK = nullptr;           // [nstate][niso][dspan]: reach of profile (isotope, Doppler index), points either side
  long *d_goff = nullptr;        // [nstate][niso][dspan]: its offset in d_ptab
  long *d_gsize = nullptr;       // [nstate + 1]: profile doubles per state, then their total
  double *d_ptab = nullptr;      // the profiles of the call, state after state
  long cap_gstate = 0, cap_ptab = 0;
  ~Lbl();
};

// Reads the TLI file(s) named by the cfg's `linedb` (comma / blank separated), merges
// their databases, uploads the lines.
void lbl_init(Engine &e, const std::string &paths);
// ext[w][l][W_local] (atm layer order) for nwalkers profiles, into e.lbl->d_ext.
void lbl_extinction(Engine &e, const double *d_prof, int nwalkers, hipStream_t st);
// Fused lazy form for eclipse spectra: per wavenumber tile the layers are
// visited from the top, each layer's line extinction is summed when it is
// reached, and the tile stops once every sample passed `toomuch` -- the deep,
// pressure-broadened layers (where almost all line-wing work sits) are never
// evaluated for opaque tiles.  `r` carries the coefficient records of
// prep_profiles (CIA, Rayleigh, path lengths) and the output pointer.
void lbl_rt_eclipse(Engine &e, const double *d_prof, int nwalkers, const RtArgs &r, hipStream_t st);
// Computes o[L][Nt][M][W] for the cfg's tlow/thigh/tempdelt and writes the
// opacity file (molecule order = TLI database order).
void lbl_write_opacity(Engine &e, const std::string &path, const std::vector<double> &tgrid);

// Diagnostics: Re w(x + i y) as the accumulation kernels evaluate it (host arrays).
void lbl_voigt_probe(const double *x, const double *y, double *k, long n);

}  // namespace bartrt
