// Host-side engine: reads the transit inputs, places the tables in HBM and
// drives the kernels.  One engine per process (one process per GPU).
#include "engine.hpp"
#include "lbl.hpp"
#include "share.hpp"
#include "step.hpp"
#include "svc.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <sys/stat.h>
#include <unistd.h>

namespace bartrt {

template <class T>
static T *dev_upload(const std::vector<T> &v) {
  T *d = nullptr;
  size_t n = std::max<size_t>(v.size(), 1);
  HIPCHK(hipMalloc(&d, n * sizeof(T)));
  if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

Engine::~Engine() {
  delete step;
  delete lbl;
  auto fr = [](void *p) { if (p) (void)hipFree(p); };
  if (kappa_share) { kappa_share->release(); kappa_share = nullptr; d_kappa = nullptr; }
  fr(d_kappa); fr(d_cia); fr(d_wn); fr(d_wn_full); fr(d_press); fr(d_mass);
  fr(d_prep_consts); fr(d_diam); fr(d_prof); fr(d_coef); fr(d_spec);
  fr(d_idx); fr(d_kstop); fr(d_rtop); fr(d_ds); fr(d_rad); fr(d_intens); fr(d_ok); fr(d_tau); fr(d_last);
  fr(d_walked); fr(d_coef2); fr(d_idx2); fr(d_kstop2); fr(d_ok2); fr(d_slog);
  if (h_pin) (void)hipHostFree(h_pin);
  if (h_flag) (void)hipHostFree(h_flag);
  for (auto e : ev) (void)hipEventDestroy(e);
  if (stream) (void)hipStreamDestroy(stream);
}

void Engine::init(int argc, const char **argv) {
  std::string cfile;
  int shard_rank = 0, shard_n = 1;
  bool no_service = false;
  device = -1;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if ((a == "-c" || a == "--config_file") && i + 1 < argc) cfile = argv[++i];
    else if (a == "--shard" && i + 2 < argc) { shard_rank = std::atoi(argv[++i]); shard_n = std::atoi(argv[++i]); }
    else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (a == "--no-service") no_service = true;
  }
  if (cfile.empty()) throw IoError{"transit_init: no '-c <configuration file>' in argv"};
  if (shard_n < 1 || shard_rank < 0 || shard_rank >= shard_n)
    throw IoError{"transit_init: bad --shard rank/nranks"};
  const TCfg c = read_tcfg(cfile);
  if (share_mode < 0) share_mode = resolve_share_mode(c, no_service);
  if (share_mode == kShareService) share_mode = kShareIpc;   // (an engine of its own was asked for: capi.hip runs the service)
  setup(c, shard_rank, shard_n);
}

int resolve_share_mode(const TCfg &cfg, bool no_service) {
  auto truthy = [](const std::string &v) { return v != "0" && v != "no" && v != "false" && v != "False"; };
  // (makeTransit writes the key bare -- code/makecfg.py:106-107: present without a value means yes)
  bool on = false;
  if (auto it = cfg.find("shareOpacity"); it != cfg.end()) on = truthy(it->second);
  if (const char *e = std::getenv("BARTRT_SHARE_OPACITY")) if (*e) on = std::string(e) != "0";
  int mode = kShareService;
  // Fewer than five worker processes: separate HIP contexts overlap their one-walker launches and beat the service's
  // one batched launch (measured, headline grid: three processes 63 against 85 us per call, four 86 / 90, five 89 / 92
  // per call but 2.8e4 / 5.4e4 spectra/s) -- when the launcher says how many chains there are (BARTRT_NCHAINS, set by
  // bart_amd.BARTfunc from the communicator's size; else the MPI world size of the spawned workers) and nothing names a
  // mode, such runs take the IPC reading.
  for (const char *name : {"BARTRT_NCHAINS", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "MV2_COMM_WORLD_SIZE"}) {
    const char *e = std::getenv(name);
    if (e && *e) {
      const int n = std::atoi(e);
      if (n >= 1 && n < 5) mode = kShareIpc;
      break;
    }
  }
  if (const char *e = std::getenv("BARTRT_SHARE_MODE")) if (*e) {
    const std::string v = e;
    if (v == "service") mode = kShareService;
    else if (v == "ipc") mode = kShareIpc;
    else if (v == "off" || v == "0" || v == "none") mode = kShareOff;
    else throw IoError{"BARTRT_SHARE_MODE: '" + v + "' is none of service, ipc, off"};
  }
  if (const char *e = std::getenv("BARTRT_SERVICE")) if (*e && std::string(e) != "0") { on = true; mode = kShareService; }
  if (!on) return kShareOff;
  if (mode == kShareService && no_service) mode = kShareIpc;
  return mode;
}

int parse_integ(const std::string &v) {
  if (v == "0" || v == "transmittance") return 0;
  if (v == "1" || v == "simpson") return 1;
  if (v == "2" || v == "trapz_tau" || v == "trapz") return 2;
  throw IoError{"integ: '" + v + "' is not an integration rule (0 transmittance, 1 simpson, 2 trapz_tau)"};
}

static bool file_exists(const std::string &p) {
  FILE *f = std::fopen(p.c_str(), "rb");
  if (f) std::fclose(f);
  return f != nullptr;
}

void Engine::setup(const TCfg &cfg_in, int shard_rank, int shard_n) {
  cfg = cfg_in;
  // Keys of the reference's whitelist (code/makecfg.py:36-52) accepted without
  // effect: verb, allowq, rad*, orbpars*, tauiso, outtau, taulevel, modlevel --
  // sampling and diagnostic controls of the CPU engine.
  // Radius-ramp cloud (makecfg.py:46-47): `cloudrad <up> <down>` (blank or comma
  // separated, in units of `cloudfct`, default `radfct`, default km) and `cloudext`
  // (cm-1): grey extinction 0 above <up>, rising linearly to cloudext at <down>,
  // cloudext below -- on the radii of each call's own hydrostatic solution.
  if (cfg_has(cfg, "cloudext") && cfg_num(cfg, "cloudext", 0.0) != 0.0) {
    std::string cr = cfg_has(cfg, "cloudrad") ? cfg["cloudrad"] : "";
    for (char &ch : cr) if (ch == ',') ch = ' ';
    TCfg tmp;
    tmp["cloudrad"] = cr;
    const auto v = cfg_list(tmp, "cloudrad");
    const double fct = cfg_num(cfg, "cloudfct", cfg_num(cfg, "radfct", 1e5));
    if (v.size() != 2 || !(v[0] > v[1]) || !(v[1] > 0) || !(fct > 0))
      throw IoError{"transit cfg: cloudext needs `cloudrad <up> <down>` with up > down > 0 (units of cloudfct)"};
    cloud_rup = v[0] * fct;
    cloud_rdown = v[1] * fct;
    cloud_ext = cfg_num(cfg, "cloudext", 0.0);
    if (!(cloud_ext > 0)) throw IoError{"transit cfg: cloudext must be positive"};
  }
  // `transparent` (makecfg.py:44): transit geometry without an opaque core below the
  // last chord; no effect on the eclipse geometry
  transparent = cfg_has(cfg, "transparent") && cfg["transparent"] != "0" && cfg["transparent"] != "no" &&
                cfg["transparent"] != "false";
  // An opacity file that does not exist yet is generated from the line list
  // first (what `transit --justOpacity` does, BART.py:561-565), by a
  // temporary line-by-line engine on the same configuration.
  if (cfg_has(cfg, "opacityfile") && !file_exists(cfg["opacityfile"])) {
    if (!cfg_has(cfg, "linedb"))
      throw IoError{"cannot open opacity file '" + cfg["opacityfile"] + "' (and no 'linedb' to build it from)"};
    if (shard_rank == 0) {
      TCfg gcfg = cfg;
      gcfg.erase("opacityfile");
      Engine gen;
      gen.device = device;
      gen.share_mode = kShareOff;
      gen.setup(gcfg, 0, 1);
      std::vector<double> tg;
      const double tlow = cfg_num(cfg, "tlow", 500.0), thigh = cfg_num(cfg, "thigh", 3000.0),
                   dt = cfg_num(cfg, "tempdelt", 100.0);
      if (!(dt > 0) || !(thigh > tlow)) throw IoError{"transit cfg: bad tlow/thigh/tempdelt"};
      for (int k = 0; tlow + k * dt <= thigh + 1e-9 * dt; k++) tg.push_back(tlow + k * dt);
      lbl_write_opacity(gen, cfg["opacityfile"], tg);
    } else {
      throw IoError{"opacity file missing: generate it on an unsharded engine first"};
    }
  }
  if (!cfg_has(cfg, "atm")) throw IoError{"transit cfg: missing 'atm'"};
  if (!cfg_has(cfg, "molfile")) throw IoError{"transit cfg: missing 'molfile'"};
  std::string sol = cfg_has(cfg, "solution") ? cfg["solution"] : "eclipse";
  if (sol == "transit") {
    solution = 1;
    if (!cfg_has(cfg, "starrad"))
      throw IoError{"solution 'transit' needs 'starrad' (stellar radius in solar radii) in the transit cfg"};
    starrad = cfg_num(cfg, "starrad", 0) * 6.96e10;  // Rsun of code/constants.py:11
    if (!(starrad > 0)) throw IoError{"transit cfg: bad 'starrad'"};
  } else if (sol != "eclipse") {
    throw IoError{"unknown solution '" + sol + "' (eclipse or transit)"};
  }
  // integration rule of the eclipse geometry (integ.hpp): `integ` in the cfg (this
  // engine's own key: the reference's source, which would settle the rule, is
  // absent), overridden by BARTRT_INTEG; number or name
  {
    // default: rule 1, the integrator SURVEY.md App. A-4 recalls for the reference's engine
    std::string v = cfg_has(cfg, "integ") ? cfg["integ"] : "1";
    if (const char *ev = std::getenv("BARTRT_INTEG")) if (*ev) v = ev;
    integ = parse_integ(v);
  }
  // `voigt exact | grid` (line-by-line evaluation, lbl.hpp / DESIGN.md C18) is checked whether or
  // not this engine ends up reading the lines
  if (cfg_has(cfg, "voigt") && cfg["voigt"] != "exact" && cfg["voigt"] != "grid")
    throw IoError{"voigt: '" + cfg["voigt"] + "' is neither exact nor grid"};
  // `cut vertical | slant`: which optical depth `toomuch` is compared with
  {
    std::string v = cfg_has(cfg, "cut") ? cfg["cut"] : "slant";
    if (const char *ev = std::getenv("BARTRT_CUT")) if (*ev) v = ev;
    if (v != "vertical" && v != "slant") throw IoError{"cut: '" + v + "' is neither vertical nor slant"};
    cut_slant = v == "slant";
  }
  {
    std::string v = cfg_has(cfg, "kernel_by") ? cfg["kernel_by"] : "local";
    if (const char *ev = std::getenv("BARTRT_KERNEL_BY")) if (*ev) v = ev;
    if (v != "whole" && v != "local") throw IoError{"kernel_by: '" + v + "' is neither whole nor local"};
    kernel_by_local = v == "local";
  }
  atm = read_atm(cfg["atm"]);
  mol = read_molfile(cfg["molfile"]);
  L = (int)atm.press.size();
  S = (int)atm.species.size();
  mass.resize(S);
  for (int s = 0; s < S; s++) {
    int j = mol.find_name(atm.species[s]);
    if (j < 0) throw IoError{"species '" + atm.species[s] + "' is not in the molecule file"};
    mass[s] = mol.mass[j];
    if (atm.species[s] == "H2") iH2 = s;
    if (atm.species[s] == "He") iHe = s;
  }
  for (int l = 0; l + 1 < L; l++)
    if (!(atm.press[l] > atm.press[l + 1]))
      throw IoError{"atmosphere file: layers must run bottom -> top (decreasing pressure)"};

  // ---- wavenumber grid and opacity table
  OpacityHeader oh;
  const bool have_table = cfg_has(cfg, "opacityfile");
  if (have_table) {
    oh = read_opacity_header(cfg["opacityfile"]);
    if (oh.nlayer != L) throw IoError{"opacity file: layer count differs from the atmosphere file"};
    for (int l = 0; l < L; l++)
      if (std::fabs(oh.press[l] - atm.press[l]) > 1e-9 * atm.press[l])
        throw IoError{"opacity file: pressure layers differ from the atmosphere file"};
    wn_full = oh.wn;
    tgrid = oh.temp;
    M = (int)oh.nmol;
    Nt = (int)oh.ntemp;
    if (M > kMaxMol) throw IoError{"opacity file: too many molecules"};
    opmol.resize(M);
    for (int m = 0; m < M; m++) {
      int j = mol.find_id(oh.molid[m]);
      if (j < 0) throw IoError{"opacity file: molecule ID not in the molecule file"};
      auto it = std::find(atm.species.begin(), atm.species.end(), mol.name[j]);
      if (it == atm.species.end())
        throw IoError{"opacity file: molecule '" + mol.name[j] + "' is not in the atmosphere file"};
      opmol[m] = (int)(it - atm.species.begin());
    }
  } else {
    double lo_wn, hi_wn;
    double wnfct = cfg_num(cfg, "wnfct", 1.0), wlfct = cfg_num(cfg, "wlfct", 1e-4);
    if (cfg_has(cfg, "wnlow") && cfg_has(cfg, "wnhigh")) {
      lo_wn = cfg_num(cfg, "wnlow", 0) * wnfct;
      hi_wn = cfg_num(cfg, "wnhigh", 0) * wnfct;
    } else if (cfg_has(cfg, "wllow") && cfg_has(cfg, "wlhigh")) {
      lo_wn = 1.0 / (cfg_num(cfg, "wlhigh", 0) * wlfct);
      hi_wn = 1.0 / (cfg_num(cfg, "wllow", 0) * wlfct);
    } else {
      throw IoError{"transit cfg: no spectral range (wnlow/wnhigh or wllow/wlhigh)"};
    }
    double d = cfg_num(cfg, "wndelt", 1.0) * wnfct;
    if (!(d > 0) || !(hi_wn > lo_wn)) throw IoError{"transit cfg: bad spectral sampling"};
    long n = (long)std::floor((hi_wn - lo_wn) / d + 1e-9) + 1;
    wn_full.resize(n);
    for (long i = 0; i < n; i++) wn_full[i] = lo_wn + d * (double)i;
    M = 0; Nt = 2; tgrid = {0.0, 1.0};
  }
  Wfull = (int)wn_full.size();
  lo = (int)((long)Wfull * shard_rank / shard_n);
  hi = (int)((long)Wfull * (shard_rank + 1) / shard_n);
  if (hi <= lo) throw IoError{"--shard leaves this rank without wavenumber samples"};
  const int Wl = W();

  // ---- geometry, hydrostatic reference
  angles = cfg_list(cfg, "raygrid");
  if (angles.empty()) angles = {0, 20, 40, 60, 80};
  A = (int)angles.size();
  if (A > kMaxAngles) throw IoError{"raygrid: too many angles"};
  for (int a = 0; a < A; a++)
    if (angles[a] < 0 || angles[a] >= 90 || (a && angles[a] <= angles[a - 1]))
      throw IoError{"raygrid: angles must increase within [0, 90)"};
  toomuch = cfg_num(cfg, "toomuch", 20.0);
  if (!cfg_has(cfg, "gsurf") || !cfg_has(cfg, "refpress") || !cfg_has(cfg, "refradius"))
    throw IoError{"transit cfg: gsurf, refpress and refradius are required"};
  gsurf = cfg_num(cfg, "gsurf", 0);
  refpress = cfg_num(cfg, "refpress", 0) * 1e6;      // bar -> barye
  refradius = cfg_num(cfg, "refradius", 0) * 1e5;    // km -> cm
  if (cfg_has(cfg, "cloudtop")) { has_cloud = 1; cloudtop = std::pow(10.0, cfg_num(cfg, "cloudtop", 0)) * 1e6; }
  if (cfg_has(cfg, "scattering")) { scat_flag = 1; scat_value = cfg_num(cfg, "scattering", 0); }

  // ---- device
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    throw HipError{hipErrorNoDevice, "no HIP device: libbartrt computes on the GPU only"};
  if (device < 0) {
    const char *lr = std::getenv("LOCAL_RANK");
    device = lr ? std::atoi(lr) % ndev : 0;
  }
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  {
    const char *e = std::getenv("BARTRT_SYNC");
    sync_poll = !(e && std::string(e) == "stream");
    if (sync_poll) {
      void *dv = nullptr;
      if (hipHostMalloc(reinterpret_cast<void **>(&h_flag), 64, hipHostMallocDefault) == hipSuccess &&
          hipHostGetDevicePointer(&dv, h_flag, 0) == hipSuccess) {
        *h_flag = 0;
        d_flag = static_cast<unsigned int *>(dv);
      } else {
        (void)hipGetLastError();
        sync_poll = false;
      }
    }
  }

  // ---- tables to HBM
  std::vector<double> wn_loc(wn_full.begin() + lo, wn_full.begin() + hi);
  d_wn = dev_upload(wn_loc);
  d_wn_full = dev_upload(wn_full);
  if (M > 0) {
    // The file's order o[L][Nt][M][W] goes up slab by slab -- a bounded number of (layer,
    // temperature) planes through one pinned host buffer and one device staging buffer
    // (256 MB, BARTRT_INIT_SLAB_BYTES) -- and is re-laid out on the device with each
    // wavenumber's molecules contiguous (kernels.hpp, "Table layout"): peak memory during
    // init is the table + one slab, on the device and on the host.
    const size_t n = (size_t)L * Nt * M * Wl;
    const long planes = (long)L * Nt;
    const size_t plane_doubles = (size_t)M * Wl;
    size_t slab_bytes = (size_t)256 << 20;
    if (const char *e = std::getenv("BARTRT_INIT_SLAB_BYTES")) slab_bytes = std::max<size_t>(1, std::strtoull(e, nullptr, 10));
    const long per = std::max<long>(1, std::min<long>(planes, (long)(slab_bytes / (plane_doubles * sizeof(double)))));
    auto upload = [&](double *dst) {
      double *h = nullptr, *d_stage = nullptr;
      hipError_t er = hipHostMalloc(&h, (size_t)per * plane_doubles * sizeof(double), hipHostMallocDefault);
      if (er == hipSuccess) er = hipMalloc(&d_stage, (size_t)per * plane_doubles * sizeof(double));
      try {
        for (long p0 = 0; p0 < planes && er == hipSuccess; p0 += per) {
          const long np = std::min(per, planes - p0);
          read_opacity_rows(cfg["opacityfile"], oh, lo, hi, p0 * M, np * M, h);
          er = hipMemcpyAsync(d_stage, h, (size_t)np * plane_doubles * sizeof(double), hipMemcpyHostToDevice, stream);
          if (er == hipSuccess) er = launch_grid_transpose(d_stage, dst + (size_t)p0 * plane_doubles, np, M, Wl, stream);
          if (er == hipSuccess) er = hipStreamSynchronize(stream);   // the slab buffers are reused
        }
      } catch (...) {
        if (h) (void)hipHostFree(h);
        if (d_stage) (void)hipFree(d_stage);
        throw;
      }
      if (h) (void)hipHostFree(h);
      if (d_stage) (void)hipFree(d_stage);
      HIPCHK(er);
    };
    // `shareOpacity` (code/makecfg.py:106-107: BART's worker processes share ONE opacity grid; a bare key in
    // the cfg makeTransit writes) or BARTRT_SHARE_OPACITY=1: the first process of this (file, device, block)
    // uploads the grid, the others map its HBM allocation through an IPC handle (share.hpp)
    if (share_mode < 0) share_mode = resolve_share_mode(cfg, true);
    if (share_mode == kShareIpc) {
      struct stat fst;
      if (stat(cfg["opacityfile"].c_str(), &fst) != 0) throw IoError{"opacity file: cannot stat " + cfg["opacityfile"]};
      char rp[PATH_MAX];
      const std::string real = realpath(cfg["opacityfile"].c_str(), rp) ? std::string(rp) : cfg["opacityfile"];
      int bus = 0;
      hipDeviceProp_t prop;
      std::string devid = std::to_string(device);
      if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        bus = prop.pciBusID;
        devid = std::to_string(prop.pciDomainID) + ":" + std::to_string(bus) + ":" + std::to_string(prop.pciDeviceID);
      }
      const std::string key = real + "|" + std::to_string((long long)fst.st_size) + "|" + std::to_string((long long)fst.st_mtime) +
                              "|" + std::to_string((long long)getuid()) + "|dev " + devid + "|wn " + std::to_string(lo) + ":" +
                              std::to_string(hi) + "|layout LTWM v1";
      kappa_share = TableShare::attach(key, n * sizeof(double), upload);
      d_kappa = kappa_share->ptr;
    } else {
      HIPCHK(hipMalloc(&d_kappa, n * sizeof(double)));
      upload(d_kappa);
    }
  }
  // CIA: resample on the local grid (zero outside the file) and lay out, per table, nt-1
  // pair planes [W][2] = (alpha_j, alpha_j+1) per wavenumber: one 16-byte load per table and
  // layer (kernels.hpp, "Table layout").  `cia_interp` (BARTRT_CIA_INTERP):
  // spline (default: the reading of the reference believed in) -- natural cubic splines in wavenumber
  // and temperature -- or linear in both;
  // the temperature spline's second derivatives then ride as one more table per file, whose
  // weights prep_body fills with the spline's curvature terms, so every RT kernel serves it.
  std::vector<double> cia_planes, cia_temp;
  PrepArgs &pa = prep;
  {
    std::string v = cfg_has(cfg, "cia_interp") ? cfg["cia_interp"] : "spline";
    if (const char *ev = std::getenv("BARTRT_CIA_INTERP")) if (*ev) v = ev;
    if (v != "linear" && v != "spline") throw IoError{"cia_interp: '" + v + "' is neither linear nor spline"};
    cia_spline = v == "spline";
  }
  // second derivatives of the natural cubic spline through (x, y), n points (n < 3: zero)
  auto spline_y2 = [](const double *x, const double *y, size_t n, size_t stride, double *y2) {
    for (size_t i = 0; i < n; i++) y2[i * stride] = 0.0;
    if (n < 3) return;
    std::vector<double> u(n, 0.0);
    for (size_t i = 1; i + 1 < n; i++) {
      const double sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
      const double p = sig * y2[(i - 1) * stride] + 2.0;
      y2[i * stride] = (sig - 1.0) / p;
      const double d = (y[(i + 1) * stride] - y[i * stride]) / (x[i + 1] - x[i]) -
                       (y[i * stride] - y[(i - 1) * stride]) / (x[i] - x[i - 1]);
      u[i] = (6.0 * d / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p;
    }
    for (size_t k = n - 1; k-- > 1;) y2[k * stride] = y2[k * stride] * y2[(k + 1) * stride] + u[k];
  };
  if (cfg_has(cfg, "csfile")) {
    auto files = split_file_list(cfg["csfile"]);
    if ((int)files.size() * (cia_spline ? 2 : 1) > kMaxCia)
      throw IoError{cia_spline ? "csfile: too many cross-section files for cia_interp spline (two table slots each)"
                               : "csfile: too many cross-section files"};
    for (auto &fn : files) {
      Cia c = read_cia(fn);
      int cc = C++;
      auto f1 = std::find(atm.species.begin(), atm.species.end(), c.s1);
      auto f2 = std::find(atm.species.begin(), atm.species.end(), c.s2);
      if (f1 == atm.species.end() || f2 == atm.species.end())
        throw IoError{"cross-section file '" + fn + "': species not in the atmosphere file"};
      pa.cia_s1[cc] = (int)(f1 - atm.species.begin());
      pa.cia_s2[cc] = (int)(f2 - atm.species.begin());
      pa.cia_nt[cc] = (int)c.temp.size();
      pa.cia_toff[cc] = (int)cia_temp.size();
      pa.cia_kind[cc] = 0;
      const size_t nw = c.wn.size(), ntc = c.temp.size();
      // resampled planes [nt][Wl] first, then the (lower, upper) pair planes
      std::vector<double> planes(ntc * (size_t)Wl), y2w(cia_spline ? nw : 0);
      for (size_t t = 0; t < ntc; t++) {
        const double *al = c.alpha.data() + t * nw;
        if (cia_spline) spline_y2(c.wn.data(), al, nw, 1, y2w.data());
        for (int i = 0; i < Wl; i++) {
          double x = wn_loc[i], v = 0.0;
          if (x >= c.wn.front() && x <= c.wn.back() && nw > 1) {
            size_t j = std::upper_bound(c.wn.begin(), c.wn.end(), x) - c.wn.begin();
            if (j >= nw) j = nw - 1;
            if (j == 0) j = 1;
            double x0 = c.wn[j - 1], x1 = c.wn[j];
            // np.interp form: slope * (x - x0) + y0
            v = (al[j] - al[j - 1]) / (x1 - x0) * (x - x0) + al[j - 1];
            if (x == x1) v = al[j];
            if (cia_spline) {
              const double h = x1 - x0, a = (x1 - x) / h, b = (x - x0) / h;
              v = a * al[j - 1] + b * al[j] + ((a * a * a - a) * y2w[j - 1] + (b * b * b - b) * y2w[j]) * (h * h) / 6.0;
            }
          } else if (nw == 1 && x == c.wn.front()) {
            v = al[0];
          }
          planes[t * Wl + i] = v;
        }
      }
      auto push_pairs = [&](const std::vector<double> &pl) {
        const size_t npair = std::max<size_t>(ntc - 1, 1);
        for (size_t t = 0; t < npair; t++) {
          const size_t hi_t = std::min(t + 1, ntc - 1);
          for (int i = 0; i < Wl; i++) {
            cia_planes.push_back(pl[t * Wl + i]);
            cia_planes.push_back(pl[hi_t * Wl + i]);
          }
        }
      };
      pa.cia_poff[cc] = (int)(cia_planes.size() / ((size_t)2 * Wl));
      push_pairs(planes);
      cia_temp.insert(cia_temp.end(), c.temp.begin(), c.temp.end());
      if (cia_spline) {
        // the second table of the file: second derivatives in T of the resampled planes
        const int c2 = C++;
        pa.cia_s1[c2] = pa.cia_s1[cc]; pa.cia_s2[c2] = pa.cia_s2[cc];
        pa.cia_nt[c2] = pa.cia_nt[cc];
        pa.cia_toff[c2] = (int)cia_temp.size();
        pa.cia_kind[c2] = 1;
        std::vector<double> y2t(planes.size(), 0.0);
        for (int i = 0; i < Wl; i++) spline_y2(c.temp.data(), planes.data() + i, ntc, (size_t)Wl, y2t.data() + i);
        pa.cia_poff[c2] = (int)(cia_planes.size() / ((size_t)2 * Wl));
        push_pairs(y2t);
        cia_temp.insert(cia_temp.end(), c.temp.begin(), c.temp.end());
      }
    }
  }
  d_cia = dev_upload(cia_planes);

  std::vector<double> dlnp(std::max(L - 1, 1), 0.0);
  for (int l = 0; l + 1 < L; l++) dlnp[l] = std::log(atm.press[l] / atm.press[l + 1]);
  d_press = dev_upload(atm.press);
  d_mass = dev_upload(mass);
  {
    std::vector<double> diam(S);
    for (int s = 0; s < S; s++) diam[s] = mol.diam[mol.find_name(atm.species[s])] * 1e-8;
    d_diam = dev_upload(diam);
  }

  // hydrostatic reference layer (makeatm.py:229-247)
  {
    int ix = 0;
    double best = std::fabs(atm.press[0] - refpress);
    for (int i = 1; i < L; i++) {
      double d = std::fabs(atm.press[i] - refpress);
      if (d < best) { best = d; ix = i; }
    }
    pa.ref_idx = ix;
    pa.ref_exact = atm.press[ix] == refpress;
    pa.ref_ib = ix < L - 1 ? ix : ix - 1;
    pa.ref_f = ix < L - 1 ? 0.0 : 1.0;
    if (L == 1) { pa.ref_ib = 0; pa.ref_f = 0.0; }
    double lp0 = std::log10(refpress);
    for (int i = 0; i + 1 < L; i++) {
      double la = std::log10(atm.press[i]), lb = std::log10(atm.press[i + 1]);
      if ((lp0 <= la && lp0 >= lb) || (lp0 >= la && lp0 <= lb)) {
        pa.ref_ib = i;
        pa.ref_f = (lp0 - la) / (lb - la);
        break;
      }
    }
    pa.ref_lnp = std::log(refpress / atm.press[ix]);
  }
  pa.L = L; pa.S = S; pa.M = M; pa.Nt = Nt; pa.C = C; pa.W = Wl;
  {
    // prep_profiles' constants in one block (layout: kernels.hpp, PrepArgs::consts)
    std::vector<double> blob(atm.press);
    blob.insert(blob.end(), dlnp.begin(), dlnp.end());
    blob.resize(2 * (size_t)L, 0.0);
    blob.insert(blob.end(), mass.begin(), mass.end());
    auto with_inverse_spacing = [&](const double *g, int n) {
      blob.insert(blob.end(), g, g + n);
      for (int j = 0; j < n; j++) blob.push_back(j + 1 < n ? 1.0 / (g[j + 1] - g[j]) : 0.0);
    };
    with_inverse_spacing(tgrid.data(), Nt);
    const size_t at = blob.size();
    blob.insert(blob.end(), cia_temp.begin(), cia_temp.end());
    blob.resize(at + 2 * cia_temp.size(), 0.0);
    for (int c = 0; c < C; c++)
      for (int j = 0; j + 1 < pa.cia_nt[c]; j++) {
        const double *g = cia_temp.data() + pa.cia_toff[c];
        blob[at + cia_temp.size() + pa.cia_toff[c] + j] = 1.0 / (g[j + 1] - g[j]);
      }
    d_prep_consts = dev_upload(blob);
  }
  pa.consts = d_prep_consts;
  if (M > kMaxMol) throw IoError{"more opacity-table molecules than the kernels are built for (16)"};
  for (int m = 0; m < M; m++) pa.opmol[m] = opmol[m];
  pa.ncia_temps = (int)cia_temp.size();
  pa.iH2 = iH2; pa.iHe = iHe;

  RtArgs &r = rt;
  r.L = L; r.M = M; r.Nt = Nt; r.C = C; r.A = A; r.W = Wl; r.Wfull = Wfull;
  r.kappa = d_kappa; r.cia = d_cia; r.wn = d_wn;
  r.kappa_bytes = (unsigned long long)L * Nt * M * Wl * 8ull;
  r.cia_bytes = (unsigned long long)cia_planes.size() * 8ull;
  for (int a = 0; a < A; a++) {
    double lo_a = a == 0 ? 0.0 : 0.5 * (angles[a - 1] + angles[a]);
    double hi_a = a == A - 1 ? 90.0 : 0.5 * (angles[a] + angles[a + 1]);
    double sl = std::sin(lo_a * kPI / 180.0), sh = std::sin(hi_a * kPI / 180.0);
    r.wgt[a] = kPI * (sh * sh - sl * sl);
    r.invmu[a] = 1.0 / std::cos(angles[a] * kPI / 180.0);
    r.wq[a] = r.wgt[a] * r.invmu[a];
    r.mu[a] = std::cos(angles[a] * kPI / 180.0);
  }
  if (!have_table && cfg_has(cfg, "linedb")) {
    lbl_init(*this, cfg["linedb"]);
    const char *m = std::getenv("BARTRT_LBL");
    // measured on the config-5 shape (tools/lbl_bench.py): tiles rarely turn opaque
    // as a whole, so the lazy fused kernel (55 ms) loses to the eager two-pass form
    // (43 ms) that exposes all layers as parallel work; BARTRT_LBL=lazy selects it
    // (the fused kernel evaluates the line sums on the output points: no oversampling)
    lbl_eager = !(m && std::string(m) == "lazy") || lbl->dev.osamp > 1 || lbl->dev.voigt_grid;
  }
  HIPCHK(hipMalloc(&d_tau, sizeof(double) * (size_t)Wl * L));
  HIPCHK(hipMalloc(&d_last, sizeof(int) * (size_t)Wl));
  ensure_walkers(16);
  // The first launch of a kernel loads its code object (the single-wave kernels' translation unit is 10 MB: 20 ms,
  // seen as ONE 20 ms call among the first of a run -- round 5, three worker processes on the chain service).  The
  // atmosphere file's own profile goes through a one-walker and a twelve-walker launch here, once, so that the
  // first MCMC step does not pay it (BARTRT_WARMUP=0: off).  Table path of the eclipse geometry.
  {
    const char *wv = std::getenv("BARTRT_WARMUP");
    if (!(wv && wv[0] == '0') && solution == 0 && !lbl && M > 0) {
      const int nprof = (S + 1) * L, nw = 12;
      std::vector<double> hp((size_t)nw * nprof);
      for (int w = 0; w < nw; w++)
        for (int l = 0; l < L; l++) {
          hp[(size_t)w * nprof + l] = atm.temp[l];
          for (int k = 0; k < S; k++) hp[(size_t)w * nprof + (size_t)(k + 1) * L + l] = atm.abund[(size_t)l * S + k];
        }
      HIPCHK(hipMemcpy(d_prof, hp.data(), sizeof(double) * hp.size(), hipMemcpyHostToDevice));
      run_dev(d_prof, 1, d_spec, d_ok, stream, false);
      run_dev(d_prof, nw, d_spec, d_ok, stream, false);
      HIPCHK(hipStreamSynchronize(stream));
      last_prof = nullptr;
      last_n = 0;
    }
  }
}

void Engine::ensure_walkers(int n) {
  if (n <= cap_walkers) return;
  int cap = std::max(n, cap_walkers * 2);
  HIPCHK(hipDeviceSynchronize());
  auto re = [&](auto *&p, size_t count) {
    if (p) HIPCHK(hipFree(p));
    p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(*p)));
  };
  if (last_prof == d_prof) last_prof = nullptr;
  pf_have_prof = nullptr;   // (prefetched records, if any, are dropped with their buffers' sizes)
  if (cap2) {
    re(d_coef2, (size_t)cap * L * coef_stride(M, C));
    re(d_idx2, (size_t)cap * L * idx_stride(C));
    re(d_kstop2, (size_t)cap);
    re(d_ok2, (size_t)cap);
    cap2 = cap;
  }
  re(d_prof, (size_t)cap * (S + 1) * L);
  re(d_coef, (size_t)cap * L * coef_stride(M, C));
  re(d_idx, (size_t)cap * L * idx_stride(C));
  re(d_kstop, (size_t)cap);
  re(d_ok, (size_t)cap);
  re(d_spec, (size_t)cap * W());
  re(d_rad, (size_t)cap * L);
  if (solution == 1) {
    re(d_rtop, (size_t)cap * L);
    re(d_ds, (size_t)cap * chord_table_size(L));
  }
  cap_walkers = cap;
}

void Engine::wait(hipStream_t st) {
  if (!sync_poll || !h_flag) { HIPCHK(hipStreamSynchronize(st)); return; }
  const unsigned int want = ++flag_seq;
  HIPCHK(hipStreamWriteValue32(st, d_flag, want, 0));
  volatile unsigned int *f = h_flag;
  long spins = 0;
  while (*f != want) {
    __builtin_ia32_pause();
    // (a launch that failed never writes: the stream itself is asked now and then, and reports the error)
    if ((++spins & 0xfffff) == 0 && hipStreamQuery(st) != hipErrorNotReady) { HIPCHK(hipStreamSynchronize(st)); break; }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

void Engine::ensure_pin(size_t bytes) {
  if (bytes <= h_pin_bytes) return;
  if (h_pin) HIPCHK(hipHostFree(h_pin));
  h_pin = nullptr;
  last_prof = nullptr;
  HIPCHK(hipHostMalloc(&h_pin, bytes, hipHostMallocDefault));
  h_pin_bytes = bytes;
}

void Engine::run_dev(const double *d_prof_in, int n, double *d_spec_out,
                     unsigned char *d_okp, hipStream_t st, bool want_tau) {
  if (n <= 0) return;
  // per-walker workspaces (records, flags) are sized by cap_walkers; the
  // caller's profile and spectrum buffers are used in place
  if (n > cap_walkers && d_prof_in != d_prof) ensure_walkers(n);
  if (lbl && solution == 0 && !want_tau && !want_intens && !lbl_eager && integ == 0 && !cut_slant) {
    // lazy fused path: layers' line sums are evaluated only as deep as the
    // optical depth requires
    run_chunk(d_prof_in, n, d_spec_out, d_okp, st, false, nullptr, true);
    return;
  }
  if (lbl) {
    // eager path (optical-depth / intensity outputs, transit geometry):
    // [walkers][L][W] extinction first, in bounded chunks.  The walkers' one-shot
    // radius / cloud / scattering overrides are [n][3]: every chunk gets its own
    // rows (run_chunk consumes the pointer it is given, so it is re-armed per chunk).
    const size_t per = (size_t)L * W() * sizeof(double);
    size_t cap_bytes = (size_t)2 << 30;
    if (const char *c = std::getenv("BARTRT_LBL_CHUNK_BYTES")) cap_bytes = std::max<size_t>(1, std::strtoull(c, nullptr, 10));
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, cap_bytes / per));
    const int nprof = (S + 1) * L;
    const double *over = prep_over_once;
    for (int off = 0; off < n; off += chunk) {
      const int m = std::min(chunk, n - off);
      lbl_extinction(*this, d_prof_in + (size_t)off * nprof, m, st);
      prep_over_once = over ? over + (size_t)3 * off : nullptr;
      run_chunk(d_prof_in + (size_t)off * nprof, m, d_spec_out + (size_t)off * W(),
                (d_okp ? d_okp : d_ok) + off, st, want_tau, lbl->d_ext);
    }
    prep_over_once = nullptr;
    return;
  }
  // `cut slant`, rule 1: the single-wave kernels keep an event log of 100 bytes per (walker, wavenumber) lane
  // (RtArgs::slog) -- 1 MB per walker at W = 1e4.  A batch whose log would pass BARTRT_SLOG_CAP_BYTES (1 GiB) goes
  // out in chunks of walkers (results are per walker: the same bits either way).
  if (cut_slant && integ == 1 && solution == 0 && !prep_hook) {   // (the fused per-step launch addresses its batch whole)
    static const size_t cap = [] {
      const char *c = std::getenv("BARTRT_SLOG_CAP_BYTES");
      return c && *c ? std::max<size_t>(1, std::strtoull(c, nullptr, 10)) : (size_t)1 << 30;
    }();
    const size_t per = std::max<size_t>(1, slant_log_bytes(1, (W() + 63) / 64, 64, A));
    const int chunk = (int)std::max<size_t>(1, cap / per);
    if (n > chunk) {
      const int nprof = (S + 1) * L;
      const double *over = prep_over_once;
      for (int off = 0; off < n; off += chunk) {
        const int m = std::min(chunk, n - off);
        prep_over_once = over ? over + (size_t)3 * off : nullptr;
        run_chunk(d_prof_in + (size_t)off * nprof, m, d_spec_out + (size_t)off * W(), (d_okp ? d_okp : d_ok) + off, st, want_tau, nullptr);
      }
      prep_over_once = nullptr;
      return;
    }
  }
  run_chunk(d_prof_in, n, d_spec_out, d_okp, st, want_tau, nullptr);
}

void Engine::run_chunk(const double *d_prof_in, int n, double *d_spec_out,
                       unsigned char *d_okp, hipStream_t st, bool want_tau, const double *d_ext,
                       bool lbl_fused) {
  // coefficient workspaces are sized by cap_walkers; the caller's profile and
  // spectrum buffers are used in place
  if (n > cap_walkers) ensure_walkers(n);
  // prefetched preparation: only the plain table path of the eclipse geometry takes part
  const bool pf_ok = !prep_hook && !prep_over_once && !lbl_fused && !d_ext && solution == 0 && !want_tau &&
                     !want_intens && !lbl;
  bool want_next = pf_ok && pf_req_prof && pf_req_n > 0;
  if (want_next && pf_req_n > cap_walkers) {
    // the workspaces have to grow for the named batch: not under a call whose own buffers are the
    // engine's (a host-buffer batch: growing frees what it is about to read and write) -- such a
    // request is dropped, the named batch is prepared by its own call
    const bool own = d_prof_in == d_prof || d_spec_out == d_spec || d_okp == d_ok;
    if (own) want_next = false;
    else ensure_walkers(pf_req_n);   // (drops prefetched records)
  }
  // the settings the layer records are built from, as they stand for THIS call: records prefetched
  // under other settings (a bartrt_set_radius / _cloudtop / _scattering in between) or on another
  // stream are not used -- the call prepares its own
  const PrepSettings now{refradius, gsurf, cloudtop, scat_value, cloud_rup, cloud_rdown, cloud_ext, has_cloud, scat_flag};
  const bool have = pf_ok && pf_have_prof && pf_have_prof == d_prof_in && pf_have_n == n && pf_have_stream == st &&
                    pf_have_set == now;
  const int bset = have ? pf_have_buf : 0;        // record buffers this call's RT kernel reads
  pf_have_prof = nullptr;
  if (want_next) {
    if (cap2 < cap_walkers) {   // first request: the second set of record buffers (nothing prefetched yet)
      HIPCHK(hipDeviceSynchronize());
      auto re2 = [&](auto *&p, size_t count) {
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        HIPCHK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(*p)));
      };
      re2(d_coef2, (size_t)cap_walkers * L * coef_stride(M, C));
      re2(d_idx2, (size_t)cap_walkers * L * idx_stride(C));
      re2(d_kstop2, (size_t)cap_walkers);
      re2(d_ok2, (size_t)cap_walkers);
      cap2 = cap_walkers;
    }
  }
  const bool use_have = have;
  double *coef_b[2] = {d_coef, d_coef2};
  idx_t *idx_b[2] = {d_idx, d_idx2};
  int *kstop_b[2] = {d_kstop, d_kstop2};
  unsigned char *ok_b[2] = {d_ok, d_ok2};
  PrepArgs pa = prep;
  pa.nwalkers = n;
  pa.prof = d_prof_in;
  pa.gsurf = gsurf; pa.refradius = refradius;
  pa.scat_flag = scat_flag; pa.scat_value = scat_value;
  pa.has_cloud = has_cloud; pa.cloudtop = cloudtop;
  pa.cloud_rup = cloud_rup; pa.cloud_rdown = cloud_rdown; pa.cloud_ext = cloud_ext;
  pa.coef = coef_b[bset]; pa.idx = idx_b[bset]; pa.kstop = kstop_b[bset];
  pa.ok = d_okp ? d_okp : d_ok;
  pa.rad_out = d_rad;
  pa.over = prep_over_once;
  const bool over_cloud = prep_over_cloud && (prep_over_once || prep_hook);
  prep_over_once = nullptr;
  pa.rtop = solution == 1 ? d_rtop : nullptr;
  pa.ds = solution == 1 ? d_ds : nullptr;
  // one to four walkers of the plain table path: the RT kernel may prepare them itself (launch_rt_folded, below)
  const bool try_fold = !use_have && !prep_hook && !want_next && solution == 0 && !lbl && !d_ext && !lbl_fused && !want_tau &&
                        !want_intens && std::max(n, sel_walkers) <= 4 && integ == 1 && cut_slant && A == 5;
  if (use_have) {
    // prepared by the previous call's RT launch; its flags go where this call wants them
    if (d_okp) HIPCHK(hipMemcpyAsync(d_okp, ok_b[bset], (size_t)n, hipMemcpyDeviceToDevice, st));
  } else if (prep_hook) HIPCHK(prep_hook(pa, st, prep_hook_ctx));
  else if (!try_fold) HIPCHK(launch_prep(pa, st));
  if (solution == 1) HIPCHK(launch_chord_table(pa, st));

  RtArgs r = rt;
  r.nwalkers = n;
  r.nsel = sel_walkers > n ? sel_walkers : 0;
  if (kernel_by_local) r.Wfull = r.W;
  r.coef = coef_b[bset]; r.idx = idx_b[bset]; r.kstop = kstop_b[bset];
  r.ext = d_ext;
  r.nprep = 0;
  if (want_next) {
    PrepArgs pn = pa;      // same engine settings; the next batch's profiles into the other buffer set
    pn.nwalkers = pf_req_n;
    pn.prof = pf_req_prof;
    pn.coef = coef_b[1 - bset]; pn.idx = idx_b[1 - bset]; pn.kstop = kstop_b[1 - bset];
    pn.ok = ok_b[1 - bset];
    pn.over = nullptr;
    pn.rad_out = nullptr;   // (bartrt_get_radius: the radii of the batch this call computes)
    r.nprep = pf_req_n;
    r.prep_next = pn;
  }
  r.cloud_on = has_cloud || over_cloud;
  r.integ = integ;
  r.cut_slant = cut_slant ? 1 : 0;
  r.toomuch = toomuch;
  if (cut_slant) slant_thresholds(r);
  r.spec = d_spec_out;
  r.tau_out = (want_tau && n == 1) ? d_tau : nullptr;
  r.last_out = (want_tau && n == 1) ? d_last : nullptr;
  r.intens_out = nullptr;
  if (want_intens && n == 1 && solution == 0) {
    if (!d_intens) HIPCHK(hipMalloc(&d_intens, sizeof(double) * (size_t)A * W()));
    r.intens_out = d_intens;
  }
  // one wave per workgroup at every batch size (measured against 128 and 256 lanes:
  // 2-5 % faster from 64 walkers up, finer turnover of the SIMDs' wave slots)
  static const int block = [] {
    const char *e = std::getenv("BARTRT_BLOCK");   // A/B runs: 64 (default), 128 or 256 lanes per workgroup
    const int b = e ? std::atoi(e) : 64;
    return (b == 128 || b == 256) ? b : 64;
  }();
  r.ntiles = (r.W + block - 1) / block;
  r.rtop = d_rtop; r.ds = d_ds;
  r.slog = nullptr;
  if (cut_slant && integ == 1 && solution == 0 && !lbl_fused) {
    // the event log of rule 1's single-wave `cut slant` kernels (rt_eclipse_s1s.hpp): 100 bytes per lane
    // (rules 0 / 2 -- rt_eclipse_fast<SLANT> -- keep none)
    const size_t need = slant_log_bytes(n, r.ntiles, block, A);
    if (need > slog_cap) {
      HIPCHK(hipDeviceSynchronize());   // (an earlier launch on any stream may still write the old log)
      if (d_slog) HIPCHK(hipFree(d_slog));
      d_slog = nullptr;
      slog_cap = 0;
      HIPCHK(hipMalloc(&d_slog, need));
      slog_cap = need;
    }
    r.slog = d_slog;
  }
  r.inv_starrad2 = solution == 1 ? 1.0 / (starrad * starrad) : 0.0;
  r.transparent = transparent ? 1 : 0;
  // Timing: the RT kernel's own dispatch stamps the two events (BARTRT_RT_LAUNCH) -- no marker
  // packets in the stream; only the fused line-by-line path (several kernels) is bracketed
  // by event records.
  // (a call that launches nothing -- no walkers -- takes no event pair: bartrt_timing_end
  // would read events no dispatch has stamped)
  const bool timed = timing && n > 0 && r.W > 0 && (timing_seen++ % timing_stride == 0);
  r.ev_start = r.ev_stop = nullptr;
  if (timed) {
    while ((int)ev.size() < ev_used + 2) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      ev.push_back(e);
    }
    if (lbl_fused) HIPCHK(hipEventRecord(ev[ev_used], st));
    else { r.ev_start = ev[ev_used]; r.ev_stop = ev[ev_used + 1]; }
  }
  r.walked_out = nullptr;
  if (want_walked && solution == 0 && !lbl_fused) {
    // the finest column any eclipse kernel records is ONE wavenumber wide (rt_eclipse_quad with one ray per lane, R = 8)
    const size_t need = (size_t)n * ((size_t)r.W + 64);
    if (need > walked_cap) {
      if (d_walked) HIPCHK(hipFree(d_walked));
      d_walked = nullptr;
      HIPCHK(hipMalloc(&d_walked, need * sizeof(int)));
      walked_cap = need;
    }
    HIPCHK(hipMemsetAsync(d_walked, 0, need * sizeof(int), st));
    r.walked_out = d_walked;
    walked_nwalkers = n;
  }
  if (lbl_fused) lbl_rt_eclipse(*this, d_prof_in, n, r, st);
  else if (solution == 1) HIPCHK(launch_transit(r, st));
  else {
    RtLaunchInfo li;
    bool folded = false;
    if (try_fold) HIPCHK(launch_rt_folded(r, pa, block, st, &li, &folded));
    if (!folded) {
      if (try_fold) HIPCHK(launch_prep(pa, st));
      li = RtLaunchInfo{};
      HIPCHK(launch_rt(r, block, st, &li));
    }
    if (want_walked) walked_info = li;
    if (want_next && li.prep_fused) {
      pf_have_prof = pf_req_prof; pf_have_n = pf_req_n; pf_have_buf = 1 - bset;
      pf_have_stream = st; pf_have_set = now;
    }
  }
  pf_req_prof = nullptr;
  pf_req_n = 0;
  if (timed) {
    if (lbl_fused) HIPCHK(hipEventRecord(ev[ev_used + 1], st));
    ev_used += 2;
  }
}

}  // namespace bartrt
