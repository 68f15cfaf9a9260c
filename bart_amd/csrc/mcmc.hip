// Native driver loop of a retrieval: differential-evolution MCMC (ter Braak
// 2006; the reference's `walk = demc`) and its snooker variant (ter Braak &
// Vrugt 2008; `walk = snooker`) over all chains per iteration, one batched model
// call (step_run_host) per iteration.  Same moves, bounds and rejection rules as
// bart_amd/sampler.py (the keys of the reference's [MCMC] section,
// examples/demo/BART_eclipse.cfg:43-102); the random streams differ, so the two
// agree statistically, not draw for draw.  Host code: an iteration is a few
// hundred flops around three kernel launches, and in Python its bookkeeping
// costs as much as the launches.
#include <algorithm>
#include <cmath>
#include <limits>
#include <random>
#include <vector>

#include "engine.hpp"
#include "step.hpp"

namespace bartrt {

void mcmc_run(Engine &e, int nch, int npars, long nsteps, const double *params, const double *pmin,
              const double *pmax, const double *stepsize, int ndata, const double *data,
              const double *uncert, int snooker, unsigned long long seed, double *chain,
              double *chisq, long *naccept_out, long *nbad) {
  if (!e.step) throw IoError{"mcmc_run: call step_setup first"};
  if (nch < 1 || npars < 1 || nsteps < 1) throw IoError{"mcmc_run: bad sizes"};
  if (ndata != e.step->nfilters) throw IoError{"mcmc_run: data length must equal the number of filters"};
  std::vector<int> free_;
  for (int j = 0; j < npars; j++) {
    if (stepsize[j] < 0)
      throw IoError{"mcmc_run: stepsize < 0 (a shared parameter in MC3's convention) is not supported"};
    if (stepsize[j] > 0) free_.push_back(j);
  }
  const int nfree = (int)free_.size();
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> normal(0.0, 1.0);
  std::uniform_real_distribution<double> unif(0.0, 1.0);
  const double inf = std::numeric_limits<double>::infinity();

  std::vector<double> band((size_t)nch * ndata), packed((size_t)nch * npars);
  std::vector<int> rows(nch), status(nch);
  if (nbad) std::fill(nbad, nbad + 4, 0L);
  // chi-square of the rows of p flagged in `use` (others: inf); a row of -1 is
  // the worker's rejection sentinel
  auto chisq_of = [&](const std::vector<double> &p, const std::vector<char> &use, std::vector<double> &out) {
    int m = 0;
    for (int i = 0; i < nch; i++) {
      out[i] = inf;
      if (!use[i]) continue;
      std::copy(p.begin() + (size_t)i * npars, p.begin() + (size_t)(i + 1) * npars,
                packed.begin() + (size_t)m * npars);
      rows[m++] = i;
    }
    if (m == 0) return;
    step_run_host(e, packed.data(), m, npars, band.data(), status.data());
    if (nbad)
      for (int k = 0; k < m; k++)
        if (status[k] >= 1 && status[k] <= 3) nbad[status[k]]++;
    for (int k = 0; k < m; k++) {
      const double *b = band.data() + (size_t)k * ndata;
      bool sentinel = true;
      double c = 0.0;
      for (int f = 0; f < ndata; f++) {
        sentinel = sentinel && b[f] == -1.0;
        const double r = (b[f] - data[f]) / uncert[f];
        c += r * r;
      }
      out[rows[k]] = sentinel ? inf : c;
    }
  };
  // the box applies to the free parameters; a fixed one keeps its configured value
  auto clip = [&](std::vector<double> &p) {
    for (int i = 0; i < nch; i++)
      for (int j : free_) {
        double &v = p[(size_t)i * npars + j];
        v = std::min(std::max(v, pmin[j]), pmax[j]);
      }
  };

  // start: the configured point jittered by the stepsizes, inside the box
  std::vector<double> x((size_t)nch * npars), prop((size_t)nch * npars), c(nch), cp(nch);
  std::vector<char> all(nch, 1), inside(nch);
  for (int i = 0; i < nch; i++)
    for (int j = 0; j < npars; j++)
      x[(size_t)i * npars + j] = params[j] + (i > 0 && stepsize[j] > 0 ? stepsize[j] * normal(rng) : 0.0);
  clip(x);
  chisq_of(x, all, c);
  for (int round = 0; round < 20; round++) {  // re-draw chains that start on a rejected model
    bool any_bad = false;
    for (int i = 0; i < nch; i++)
      if (!std::isfinite(c[i])) {
        any_bad = true;
        for (int j = 0; j < npars; j++)
          x[(size_t)i * npars + j] = params[j] + (stepsize[j] > 0 ? 0.1 * stepsize[j] * normal(rng) : 0.0);
      }
    if (!any_bad) break;
    clip(x);
    chisq_of(x, all, c);
  }
  if (std::none_of(c.begin(), c.end(), [](double v) { return std::isfinite(v); }))
    throw IoError{"mcmc_run: no chain starts on a physical model: check params/pmin/pmax"};

  auto other = [&](int i, int a, int b) {  // uniform over the chains other than i, a, b (a, b < 0: unused)
    int ex[3] = {i, a, b}, k = 1 + (a >= 0) + (b >= 0);
    std::sort(ex, ex + 3);                 // unused entries (-1) sort first
    int draw = (int)(unif(rng) * (nch - k));
    if (draw >= nch - k) draw = nch - k - 1;
    for (int q = 3 - k; q < 3; q++) draw += draw >= ex[q];
    return draw;
  };
  const double gamma0 = 2.38 / std::sqrt(2.0 * std::max(nfree, 1));
  const bool do_snooker = snooker && nch > 3;
  long naccept = 0;
  std::vector<double> logjac(nch);
  for (long t = 0; t < nsteps; t++) {
    prop = x;
    for (int i = 0; i < nch; i++) {
      const int r1 = nch > 1 ? other(i, -1, -1) : i;
      const int r2 = nch > 2 ? other(i, r1, -1) : r1;
      double *pi = prop.data() + (size_t)i * npars;
      const double *xi = x.data() + (size_t)i * npars;
      const double *x1 = x.data() + (size_t)r1 * npars, *x2 = x.data() + (size_t)r2 * npars;
      logjac[i] = 0.0;
      if (do_snooker && t % 10 != 0) {
        // snooker update: move along the line through a third chain
        const int z = other(i, r1, r2);
        const double *xz = x.data() + (size_t)z * npars;
        double nd = 0.0, proj = 0.0;
        for (int j : free_) nd += (xi[j] - xz[j]) * (xi[j] - xz[j]);
        nd = std::sqrt(nd);
        if (nd == 0.0) nd = 1.0;
        for (int j : free_) proj += (x1[j] - x2[j]) * (xi[j] - xz[j]) / nd;
        const double g = 1.2 + unif(rng);
        double ndn = 0.0;
        for (int j : free_) {
          pi[j] = xi[j] + g * proj * (xi[j] - xz[j]) / nd;
          ndn += (pi[j] - xz[j]) * (pi[j] - xz[j]);
        }
        logjac[i] = (nfree - 1) * (std::log(std::max(std::sqrt(ndn), 1e-300)) - std::log(nd));
      } else {
        const double gam = t % 10 == 0 ? 1.0 : gamma0;
        for (int j : free_) pi[j] = xi[j] + gam * (x1[j] - x2[j]) + 1e-3 * stepsize[j] * normal(rng);
      }
      inside[i] = 1;
      for (int j : free_) inside[i] = inside[i] && pi[j] >= pmin[j] && pi[j] <= pmax[j];
    }
    chisq_of(prop, inside, cp);
    for (int i = 0; i < nch; i++) {
      const double loga = -0.5 * (cp[i] - c[i]) + logjac[i];
      if (std::isfinite(cp[i]) && std::log(unif(rng)) < loga) {
        std::copy(prop.begin() + (size_t)i * npars, prop.begin() + (size_t)(i + 1) * npars,
                  x.begin() + (size_t)i * npars);
        c[i] = cp[i];
        naccept++;
      }
      std::copy(x.begin() + (size_t)i * npars, x.begin() + (size_t)(i + 1) * npars,
                chain + ((size_t)i * nsteps + t) * npars);
      chisq[(size_t)i * nsteps + t] = c[i];
    }
  }
  if (naccept_out) *naccept_out = naccept;
}

}  // namespace bartrt
