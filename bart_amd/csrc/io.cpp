// Host-side readers for the engine's input files (transit formats).
#include "io.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace bartrt {

static std::vector<std::string> split_ws(const std::string &s) {
  std::vector<std::string> out;
  std::istringstream is(s);
  std::string t;
  while (is >> t) out.push_back(t);
  return out;
}

static std::string trim(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r\n");
  if (a == std::string::npos) return "";
  size_t b = s.find_last_not_of(" \t\r\n");
  return s.substr(a, b - a + 1);
}

static double to_num(const std::string &s, const std::string &what) {
  char *end = nullptr;
  double v = std::strtod(s.c_str(), &end);
  if (end == s.c_str()) throw IoError{"cannot parse number '" + s + "' in " + what};
  return v;
}

TCfg read_tcfg(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw IoError{"cannot open transit configuration file '" + path + "'"};
  TCfg c;
  std::string line;
  while (std::getline(f, line)) {
    line = trim(line);
    if (line.empty() || line[0] == '#' || line[0] == ';') continue;
    size_t sp = line.find_first_of(" \t");
    std::string key = line.substr(0, sp);
    std::string val = sp == std::string::npos ? "" : trim(line.substr(sp));
    // makecfg.makeTransit writes one `key value` line per value of a multi-line
    // BART option (code/makecfg.py:93-104: `linedb a.tli` then `linedb b.tli`):
    // the file lists accumulate, comma-separated like csfile; any other repeated
    // key keeps its last value
    auto it = c.find(key);
    if (it != c.end() && !it->second.empty() && !val.empty() && (key == "linedb" || key == "csfile"))
      it->second += "," + val;
    else
      c[key] = val;
  }
  return c;
}

std::vector<std::string> split_file_list(const std::string &s) {
  std::vector<std::string> out;
  std::string cur;
  auto flush = [&] {
    const std::string t = trim(cur);
    if (!t.empty()) out.push_back(t);
    cur.clear();
  };
  for (char ch : s) {
    if (ch == ',' || ch == '\n') flush();
    else cur.push_back(ch);
  }
  flush();
  return out;
}

bool cfg_has(const TCfg &c, const std::string &k) {
  auto it = c.find(k);
  return it != c.end() && !it->second.empty();
}

double cfg_num(const TCfg &c, const std::string &k, double dflt) {
  if (!cfg_has(c, k)) return dflt;
  return to_num(split_ws(c.at(k))[0], "key " + k);
}

std::vector<double> cfg_list(const TCfg &c, const std::string &k) {
  std::vector<double> out;
  if (!cfg_has(c, k)) return out;
  for (auto &t : split_ws(c.at(k))) out.push_back(to_num(t, "key " + k));
  return out;
}

Atm read_atm(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw IoError{"cannot open atmosphere file '" + path + "'"};
  Atm a;
  double ur = 1e5, up = 1e6, ut = 1.0;  // defaults: km, bar, K
  std::string line;
  bool in_data = false;
  while (std::getline(f, line)) {
    std::string s = trim(line);
    if (in_data) {
      if (s.empty()) break;
      auto t = split_ws(s);
      size_t S = a.species.size();
      if (t.size() != S + 3)
        throw IoError{"atmosphere file '" + path + "': expected radius, pressure, "
                      "temperature and " + std::to_string(S) + " abundances per row"};
      a.radius.push_back(to_num(t[0], path) * ur);
      a.press.push_back(to_num(t[1], path) * up);
      a.temp.push_back(to_num(t[2], path) * ut);
      for (size_t i = 0; i < S; i++) a.abund.push_back(to_num(t[3 + i], path));
      continue;
    }
    auto t = split_ws(s);
    if (t.size() == 2 && t[0] == "ur") ur = to_num(t[1], path);
    else if (t.size() == 2 && t[0] == "up") up = to_num(t[1], path);
    else if (t.size() == 2 && t[0] == "ut") ut = to_num(t[1], path);
    else if (t.size() == 2 && t[0] == "q") {
      if (t[1] != "number") throw IoError{"atmosphere file: only 'q number' abundances are supported"};
    } else if (s == "#SPECIES") {
      if (!std::getline(f, line)) break;
      a.species = split_ws(line);
    } else if (s == "#TEADATA") {
      std::getline(f, line);  // column header
      in_data = true;
    }
  }
  if (a.species.empty() || a.press.empty())
    throw IoError{"atmosphere file '" + path + "': no #SPECIES / #TEADATA content"};
  return a;
}

int MolInfo::find_name(const std::string &n) const {
  for (size_t i = 0; i < name.size(); i++)
    if (name[i] == n) return (int)i;
  return -1;
}
int MolInfo::find_id(int v) const {
  for (size_t i = 0; i < id.size(); i++)
    if (id[i] == v) return (int)i;
  return -1;
}

MolInfo read_molfile(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw IoError{"cannot open molecule file '" + path + "'"};
  MolInfo m;
  std::string line;
  while (std::getline(f, line)) {
    size_t h = line.find('#');
    if (h != std::string::npos) line = line.substr(0, h);
    auto t = split_ws(line);
    if (t.size() < 4) continue;
    m.id.push_back((int)to_num(t[0], path));
    m.name.push_back(t[1]);
    m.mass.push_back(to_num(t[2], path));
    m.diam.push_back(to_num(t[3], path));
  }
  if (m.id.empty()) throw IoError{"molecule file '" + path + "' holds no entries"};
  return m;
}

OpacityHeader read_opacity_header(const std::string &path) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) throw IoError{"cannot open opacity file '" + path + "'"};
  OpacityHeader h;
  long dims[4];
  bool ok = std::fread(dims, sizeof(long), 4, fp) == 4;
  if (ok) {
    h.nmol = dims[0]; h.ntemp = dims[1]; h.nlayer = dims[2]; h.nwave = dims[3];
    ok = h.nmol > 0 && h.ntemp > 1 && h.nlayer > 0 && h.nwave > 0 &&
         h.nmol < 4096 && h.ntemp < 100000 && h.nlayer < 100000;
  }
  if (ok) {
    // the counts must fit the file before anything is sized by them
    const long here = std::ftell(fp);
    std::fseek(fp, 0, SEEK_END);
    const long fsize = std::ftell(fp);
    std::fseek(fp, here, SEEK_SET);
    const long double need = 32.0L + 4.0L * h.nmol + 8.0L * (h.ntemp + h.nlayer + h.nwave) +
                             8.0L * h.nlayer * h.ntemp * h.nmol * (long double)h.nwave;
    ok = need <= (long double)fsize;
  }
  if (ok) {
    h.molid.resize(h.nmol); h.temp.resize(h.ntemp);
    h.press.resize(h.nlayer); h.wn.resize(h.nwave);
    ok = std::fread(h.molid.data(), sizeof(int), h.nmol, fp) == (size_t)h.nmol &&
         std::fread(h.temp.data(), sizeof(double), h.ntemp, fp) == (size_t)h.ntemp &&
         std::fread(h.press.data(), sizeof(double), h.nlayer, fp) == (size_t)h.nlayer &&
         std::fread(h.wn.data(), sizeof(double), h.nwave, fp) == (size_t)h.nwave;
    h.data_offset = std::ftell(fp);
  }
  if (ok) {
    std::fseek(fp, 0, SEEK_END);
    long need = h.data_offset + 8L * h.nlayer * h.ntemp * h.nmol * h.nwave;
    ok = std::ftell(fp) >= need;
  }
  std::fclose(fp);
  if (!ok) throw IoError{"opacity file '" + path + "' is truncated or not an opacity grid"};
  return h;
}

void read_opacity_block(const std::string &path, const OpacityHeader &h, long lo,
                        long hi, double *dst) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) throw IoError{"cannot open opacity file '" + path + "'"};
  const long nrows = h.nlayer * h.ntemp * h.nmol, w = hi - lo;
  bool ok = true;
  if (lo == 0 && hi == h.nwave) {
    std::fseek(fp, h.data_offset, SEEK_SET);
    ok = std::fread(dst, sizeof(double), (size_t)nrows * w, fp) == (size_t)nrows * w;
  } else {
    for (long r = 0; r < nrows && ok; r++) {
      std::fseek(fp, h.data_offset + 8L * (r * h.nwave + lo), SEEK_SET);
      ok = std::fread(dst + r * w, sizeof(double), w, fp) == (size_t)w;
    }
  }
  std::fclose(fp);
  if (!ok) throw IoError{"short read from opacity file '" + path + "'"};
}

void read_opacity_rows(const std::string &path, const OpacityHeader &h, long lo, long hi, long row0,
                       long nrows, double *dst) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) throw IoError{"cannot open opacity file '" + path + "'"};
  const long total = h.nlayer * h.ntemp * h.nmol, w = hi - lo;
  bool ok = row0 >= 0 && nrows >= 0 && row0 + nrows <= total;
  if (ok && lo == 0 && hi == h.nwave) {
    std::fseek(fp, h.data_offset + 8L * row0 * h.nwave, SEEK_SET);
    ok = std::fread(dst, sizeof(double), (size_t)nrows * w, fp) == (size_t)nrows * w;
  } else {
    for (long r = 0; r < nrows && ok; r++) {
      std::fseek(fp, h.data_offset + 8L * ((row0 + r) * h.nwave + lo), SEEK_SET);
      ok = std::fread(dst + r * w, sizeof(double), w, fp) == (size_t)w;
    }
  }
  std::fclose(fp);
  if (!ok) throw IoError{"short read from opacity file '" + path + "'"};
}

// Two layouts, told apart by the first line that is neither blank nor a comment:
//  * a line starting with '@': the sectioned text layout (`@SPECIES`, `@TEMPERATURES`, `@DATA` rows of
//    `wn alpha(T1) alpha(T2) ...`, cm-1 amagat-2) of the CIA files shipped with the reference's engine
//    (examples/demo/BART_eclipse.cfg:119 names one; the files themselves are not in the checkout);
//  * anything else: the HITRAN collision-induced-absorption layout, the only format the reference's manual
//    names ("They must be in HITRAN cross-section format", doc/BART_user_manual/BART_user_manual.tex:506-510):
//    per temperature a 100-column header -- symbol `A-B` [0,20), first and last wavenumber [20,30) [30,40),
//    number of points [40,47), temperature [47,54), then maximum, resolution, comment, reference -- followed
//    by that many `wavenumber  value` rows in cm5 molecule-2.  Converted on reading to the engine's cm-1
//    amagat-2 (x Loschmidt^2, the constant the kernels divide the densities by).  All blocks of a file must
//    be on one wavenumber grid (one spectral range per file); blocks are sorted by temperature.
static Cia read_cia_hitran(std::ifstream &f, const std::string &path, std::string line) {
  constexpr double kLoschmidt = 2.68679e19;     // = kAMAGAT (kernels.hpp)
  Cia c;
  struct Block { double T; std::vector<double> wn, v; };
  std::vector<Block> blocks;
  auto header = [&](const std::string &h, std::string &sym, double &w0, double &w1, long &n, double &T) {
    auto num = [&](const std::string &x, double &out) {
      const std::string t = trim(x);
      if (t.empty()) return false;
      char *end = nullptr;
      out = std::strtod(t.c_str(), &end);
      return end && *end == 0 && std::isfinite(out);
    };
    double dn = 0;
    if (h.size() >= 54 && num(h.substr(20, 10), w0) && num(h.substr(30, 10), w1) && num(h.substr(40, 7), dn) &&
        num(h.substr(47, 7), T)) {
      sym = trim(h.substr(0, 20));
    } else {
      auto t = split_ws(h);      // (a hand-made file: the same fields separated by blanks)
      if (t.size() < 5 || !num(t[1], w0) || !num(t[2], w1) || !num(t[3], dn) || !num(t[4], T))
        throw IoError{"cross-section file '" + path + "': not a HITRAN CIA header: '" + trim(h).substr(0, 60) + "'"};
      sym = t[0];
    }
    if (!(dn >= 2) || dn > 5e7 || dn != std::floor(dn)) throw IoError{"cross-section file '" + path + "': bad number of points in a block header"};
    if (!(T > 0) || !(w1 > w0)) throw IoError{"cross-section file '" + path + "': bad temperature or spectral range in a block header"};
    n = (long)dn;
  };
  std::string sym0;
  for (;;) {
    std::string sym;
    double w0, w1, T;
    long n;
    header(line, sym, w0, w1, n, T);
    if (sym0.empty()) sym0 = sym;
    else if (sym != sym0) throw IoError{"cross-section file '" + path + "': blocks of different pairs (" + sym0 + ", " + sym + ")"};
    Block b;
    b.T = T;
    b.wn.reserve((size_t)std::min<long>(n, 1 << 20)); b.v.reserve((size_t)std::min<long>(n, 1 << 20));
    for (long i = 0; i < n; i++) {
      if (!std::getline(f, line)) throw IoError{"cross-section file '" + path + "' is truncated (a block is shorter than its header says)"};
      auto t = split_ws(line);
      if (t.size() < 2) throw IoError{"cross-section file '" + path + "': a data row needs a wavenumber and a value"};
      b.wn.push_back(to_num(t[0], path));
      b.v.push_back(to_num(t[1], path) * kLoschmidt * kLoschmidt);
      if (i && !(b.wn[i] > b.wn[i - 1])) throw IoError{"cross-section file '" + path + "': wavenumbers must increase within a block"};
    }
    blocks.push_back(std::move(b));
    bool more = false;
    while (std::getline(f, line)) {
      const std::string s = trim(line);
      if (!s.empty() && s[0] != '#') { more = true; break; }
    }
    if (!more) break;
  }
  const size_t dash = sym0.find('-');
  if (dash == std::string::npos || dash == 0 || dash + 1 >= sym0.size())
    throw IoError{"cross-section file '" + path + "': the pair must read A-B, not '" + sym0 + "'"};
  c.s1 = sym0.substr(0, dash);
  c.s2 = sym0.substr(dash + 1);
  std::stable_sort(blocks.begin(), blocks.end(), [](const Block &a, const Block &b) { return a.T < b.T; });
  for (size_t i = 0; i < blocks.size(); i++) {
    if (i && blocks[i].T == blocks[i - 1].T)
      throw IoError{"cross-section file '" + path + "': two blocks at one temperature (one spectral range per file)"};
    if (blocks[i].wn != blocks[0].wn)
      throw IoError{"cross-section file '" + path + "': the blocks are on different wavenumber grids (resample the file onto one)"};
  }
  const size_t nw = blocks[0].wn.size(), nt = blocks.size();
  c.wn = blocks[0].wn;
  c.temp.resize(nt);
  c.alpha.resize(nt * nw);
  for (size_t t = 0; t < nt; t++) {
    c.temp[t] = blocks[t].T;
    std::copy(blocks[t].v.begin(), blocks[t].v.end(), c.alpha.begin() + t * nw);
  }
  return c;
}

Cia read_cia(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw IoError{"cannot open cross-section file '" + path + "'"};
  Cia c;
  std::string line, mode;
  std::vector<std::vector<double>> rows;
  bool first = true;
  while (std::getline(f, line)) {
    std::string s = trim(line);
    if (s.empty() || s[0] == '#') continue;
    if (first && s[0] != '@') return read_cia_hitran(f, path, line);
    first = false;
    if (s[0] == '@') { mode = s; continue; }
    auto t = split_ws(s);
    if (mode == "@SPECIES" && c.s1.empty()) {
      if (t.size() != 2) throw IoError{"cross-section file '" + path + "': @SPECIES needs two names"};
      c.s1 = t[0]; c.s2 = t[1];
    } else if (mode == "@TEMPERATURES" && c.temp.empty()) {
      for (auto &x : t) c.temp.push_back(to_num(x, path));
    } else if (mode == "@DATA") {
      if (t.size() != c.temp.size() + 1)
        throw IoError{"cross-section file '" + path + "': row width does not match @TEMPERATURES"};
      std::vector<double> r;
      for (auto &x : t) r.push_back(to_num(x, path));
      rows.push_back(r);
    }
  }
  if (c.s1.empty() || c.temp.empty() || rows.size() < 2)
    throw IoError{"cross-section file '" + path + "' is incomplete"};
  size_t nw = rows.size(), nt = c.temp.size();
  c.wn.resize(nw);
  c.alpha.resize(nt * nw);
  for (size_t i = 0; i < nw; i++) {
    c.wn[i] = rows[i][0];
    for (size_t t = 0; t < nt; t++) c.alpha[t * nw + i] = rows[i][1 + t];
  }
  return c;
}

namespace {
struct BinReader {
  FILE *fp;
  const std::string &path;
  long fsize = -1;
  // bytes between the read position and the end of the file
  long left() {
    if (fsize < 0) {
      const long here = std::ftell(fp);
      std::fseek(fp, 0, SEEK_END);
      fsize = std::ftell(fp);
      std::fseek(fp, here, SEEK_SET);
    }
    return fsize - std::ftell(fp);
  }
  template <class T>
  T get() {
    T v;
    if (std::fread(&v, sizeof(T), 1, fp) != 1) throw IoError{"TLI file '" + path + "' is truncated"};
    return v;
  }
  std::string str() {
    unsigned short n = get<unsigned short>();
    std::string s(n, '\0');
    if (n && std::fread(&s[0], 1, n, fp) != n) throw IoError{"TLI file '" + path + "' is truncated"};
    return s;
  }
  template <class T>
  void arr(std::vector<T> &v, size_t n) {
    // a count the file cannot hold is a corrupt header, not an allocation request
    if (n > (size_t)left() / sizeof(T)) throw IoError{"TLI file '" + path + "' is truncated"};
    v.resize(n);
    if (n && std::fread(v.data(), sizeof(T), n, fp) != n)
      throw IoError{"TLI file '" + path + "' is truncated"};
  }
};
}  // namespace

Tli read_tli(const std::string &path) {
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) throw IoError{"cannot open TLI file '" + path + "'"};
  Tli t;
  try {
    BinReader r{fp, path};
    if (r.get<int>() != 0x494C54FF) throw IoError{"'" + path + "' is not a TLI file (bad magic)"};
    r.get<unsigned short>(); r.get<unsigned short>(); r.get<unsigned short>();
    t.wn_lo = r.get<double>();
    t.wn_hi = r.get<double>();
    unsigned short ndb = r.get<unsigned short>();
    t.db.resize(ndb);
    for (auto &d : t.db) {
      d.name = r.str();
      d.molecule = r.str();
      unsigned short nt = r.get<unsigned short>(), ni = r.get<unsigned short>();
      if (nt < 1 || ni < 1) throw IoError{"TLI file '" + path + "': empty database header"};
      r.arr(d.temp, nt);
      d.iso.resize(ni);
      for (auto &i : d.iso) {
        i.name = r.str();
        i.mass = r.get<double>();
        i.ratio = r.get<double>();
        r.arr(i.Z, nt);
      }
    }
    for (auto &d : t.db) {
      long long n = r.get<long long>();
      if (n < 0 || n > (1LL << 33)) throw IoError{"TLI file '" + path + "': bad transition count"};
      r.arr(d.wn, (size_t)n);
      r.arr(d.isoid, (size_t)n);
      r.arr(d.elow, (size_t)n);
      r.arr(d.gf, (size_t)n);
      for (size_t k = 0; k < (size_t)n; k++) {
        if (d.isoid[k] < 0 || (size_t)d.isoid[k] >= d.iso.size())
          throw IoError{"TLI file '" + path + "': isotope index out of range"};
        if (k && d.wn[k] < d.wn[k - 1])
          throw IoError{"TLI file '" + path + "': transitions are not sorted by wavenumber"};
      }
    }
  } catch (...) {
    std::fclose(fp);
    throw;
  }
  std::fclose(fp);
  return t;
}

}  // namespace bartrt
