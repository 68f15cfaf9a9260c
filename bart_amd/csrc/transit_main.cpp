// Standalone run of the engine: the replacement of the `transit` executable
// BART calls outside the MCMC loop,
//   transit -c <cfg> --justOpacity      (BART.py:561-565, examples/demo/README:12)
//   transit -c bestFit_tconfig.cfg      (code/bestFit.py:421-427) -> outspec
//   transit -c cf_tconfig.cfg           (code/cf.py:46-64, savefiles yes) -> tau.dat
// through the public C ABI only (include/bartrt.h).  The atmosphere file's own
// temperature and abundances are the model.  Output files:
//   outspec    1 header line; wavelength[um]  value      (read by code/readtransit.py:23-64)
//   outtoomuch 1 header line; wavelength[um]  radius[km] where tau passed toomuch (0: never)
//   outintens  1 header line; wavelength[um]  I(angle_1) ... I(angle_A)
//   outsample  the wavenumber and radius samplings
//   tau.dat    (savefiles yes) per wavenumber three lines: "wavenumber[cm-1]: <wn>",
//              the optical depths of the L layers from the top, "last: <k>"
//              (read by code/cf.py:68-94)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/bartrt.h"

static std::map<std::string, std::string> read_cfg(const std::string &path) {
  std::map<std::string, std::string> c;
  std::ifstream f(path);
  std::string line;
  while (std::getline(f, line)) {
    size_t a = line.find_first_not_of(" \t\r\n");
    if (a == std::string::npos || line[a] == '#' || line[a] == ';') continue;
    std::istringstream is(line);
    std::string k, v;
    is >> k;
    std::getline(is, v);
    size_t b = v.find_first_not_of(" \t");
    c[k] = b == std::string::npos ? "" : v.substr(b, v.find_last_not_of(" \t\r\n") - b + 1);
  }
  return c;
}

static void die(const char *what) {
  std::fprintf(stderr, "transit: %s: %s\n", what, bartrt_last_error());
  std::exit(1);
}

int main(int argc, char **argv) {
  std::string cfg;
  bool just_opacity = false;
  std::vector<const char *> pass = {"transit"};
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if ((a == "-c" || a == "--config_file") && i + 1 < argc) { cfg = argv[++i]; pass.push_back("-c"); pass.push_back(argv[i]); }
    else if (a == "--justOpacity") just_opacity = true;
    else pass.push_back(argv[i]);
  }
  if (cfg.empty()) {
    std::fprintf(stderr, "usage: transit -c <configuration file> [--justOpacity]\n");
    return 2;
  }
  auto c = read_cfg(cfg);
  if (just_opacity && !c.count("opacityfile")) {
    std::fprintf(stderr, "transit: --justOpacity needs an 'opacityfile' in the configuration\n");
    return 2;
  }
  if (bartrt_init((int)pass.size(), pass.data()) < 0) die("initialisation failed");
  if (just_opacity) {
    std::printf("Opacity grid ready: %s\n", c["opacityfile"].c_str());
    bartrt_free_memory();
    return 0;
  }
  const int nwave = bartrt_get_no_samples(), nprof = bartrt_get_nprof(), L = bartrt_get_nlayers();
  int lo = 0, hi = 0;
  bartrt_get_local_range(&lo, &hi);
  if (hi - lo != nwave) { std::fprintf(stderr, "transit: standalone runs are unsharded\n"); return 2; }
  std::vector<double> wn(nwave), prof(nprof), spec(nwave), rad(L), tau((size_t)nwave * L);
  std::vector<int> last(nwave);
  if (bartrt_get_waveno_arr(wn.data(), nwave) < 0) die("get_waveno_arr");
  if (bartrt_get_atm_profile(prof.data(), nprof) < 0) die("get_atm_profile");
  if (bartrt_run_transit(prof.data(), nprof, spec.data(), nwave) < 0) die("run_transit");
  if (bartrt_get_radius(rad.data(), L) < 0) die("get_radius");
  if (bartrt_get_tau(tau.data(), last.data(), nwave, L) < 0) die("get_tau");
  const bool eclipse = !c.count("solution") || c["solution"] == "eclipse";
  if (eclipse) {   // runs under different rules are not to be confused (the files keep the reference's layouts)
    int rule = -1;
    static const char *names[] = {"transmittance trapezoid", "Simpson hybrid, SURVEY App. A-4", "trapezoid in tau"};
    if (bartrt_get_integ(&rule) == 0 && rule >= 0 && rule < 3)
      std::printf("Integration rule of the eclipse geometry: integ %d (%s)\n", rule, names[rule]);
    int slant = -1, spline = -1;
    if (bartrt_get_cut(&slant) == 0 && bartrt_get_cia_interp(&spline) == 0)
      std::printf("`toomuch` cuts each ray's %s optical depth (cut %s); CIA interpolation: %s\n", slant ? "slant" : "the column's vertical",
                  slant ? "slant" : "vertical", spline ? "natural cubic splines (cia_interp spline)" : "linear (cia_interp linear)");
  }
  const double toomuch = c.count("toomuch") ? std::atof(c["toomuch"].c_str()) : 20.0;

  if (c.count("outspec")) {
    FILE *f = std::fopen(c["outspec"].c_str(), "w");
    if (!f) { std::perror(c["outspec"].c_str()); return 1; }
    std::fprintf(f, eclipse ? "#wvl [um]            flux [erg/s/cm]\n" : "#wvl [um]            modulation\n");
    for (int i = 0; i < nwave; i++) std::fprintf(f, "%-20.12g %.12g\n", 1e4 / wn[i], spec[i]);
    std::fclose(f);
  }
  if (c.count("outtoomuch")) {
    FILE *f = std::fopen(c["outtoomuch"].c_str(), "w");
    if (!f) { std::perror(c["outtoomuch"].c_str()); return 1; }
    std::fprintf(f, "#wvl [um]            radius [km] where the optical depth passes %g (0: nowhere)\n", toomuch);
    for (int i = 0; i < nwave; i++) {
      const int k = last[i];
      const bool reached = tau[(size_t)i * L + k] > toomuch;
      std::fprintf(f, "%-20.12g %.12g\n", 1e4 / wn[i], reached ? rad[L - 1 - k] / 1e5 : 0.0);
    }
    std::fclose(f);
  }
  if (c.count("outintens") && eclipse) {
    const int A = bartrt_get_nangles();
    std::vector<double> ang(A), in((size_t)A * nwave);
    bartrt_get_angles(ang.data(), A);
    if (bartrt_get_intensity(in.data(), A, nwave) < 0) die("get_intensity");
    FILE *f = std::fopen(c["outintens"].c_str(), "w");
    if (!f) { std::perror(c["outintens"].c_str()); return 1; }
    std::fprintf(f, "#wvl [um]            intensity [erg/s/cm/sr] at");
    for (int a = 0; a < A; a++) std::fprintf(f, " %g", ang[a]);
    std::fprintf(f, " deg\n");
    for (int i = 0; i < nwave; i++) {
      std::fprintf(f, "%-20.12g", 1e4 / wn[i]);
      for (int a = 0; a < A; a++) std::fprintf(f, " %.12g", in[(size_t)a * nwave + i]);
      std::fprintf(f, "\n");
    }
    std::fclose(f);
  }
  if (c.count("outsample")) {
    FILE *f = std::fopen(c["outsample"].c_str(), "w");
    if (!f) { std::perror(c["outsample"].c_str()); return 1; }
    std::fprintf(f, "# wavenumber sampling [cm-1]: n = %d, first = %.9g, last = %.9g, step = %.9g\n", nwave,
                 wn[0], wn[nwave - 1], nwave > 1 ? wn[1] - wn[0] : 0.0);
    std::fprintf(f, "# radius sampling [km], %d layers bottom -> top (hydrostatic)\n", L);
    for (int l = 0; l < L; l++) std::fprintf(f, "%.9g\n", rad[l] / 1e5);
    std::fclose(f);
  }
  if (c.count("savefiles") && c["savefiles"] == "yes") {
    FILE *f = std::fopen("tau.dat", "w");
    if (!f) { std::perror("tau.dat"); return 1; }
    std::fprintf(f, "# optical depth per wavenumber; layers from the top of the atmosphere\n");
    for (int i = 0; i < nwave; i++) {
      std::fprintf(f, "wavenumber[cm-1]: %.12g\n", wn[i]);
      for (int k = 0; k < L; k++) std::fprintf(f, "%.11e ", tau[(size_t)i * L + k]);
      std::fprintf(f, "\nlast: %d\n", last[i]);
    }
    std::fclose(f);
  }
  bartrt_free_memory();
  return 0;
}
