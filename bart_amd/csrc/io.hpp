// Host-side readers for the engine's input files (transit formats).
#pragma once
#include <map>
#include <string>
#include <vector>

namespace bartrt {

struct IoError {
  std::string msg;
};

// transit configuration file: whitespace-separated `key value` lines, '#' or
// ';' comments (reference examples/demo/transit_demo.cfg:1-6).
using TCfg = std::map<std::string, std::string>;
TCfg read_tcfg(const std::string &path);
double cfg_num(const TCfg &c, const std::string &k, double dflt);
bool cfg_has(const TCfg &c, const std::string &k);
std::vector<double> cfg_list(const TCfg &c, const std::string &k);

// Atmosphere file as written by makeatm.makeRadius + makeatm.reformat
// (reference code/makeatm.py:551-603, 841-896).  Layers bottom -> top.
struct Atm {
  std::vector<std::string> species;
  std::vector<double> radius;  // cm
  std::vector<double> press;   // barye
  std::vector<double> temp;    // K
  std::vector<double> abund;   // [L][S] mole mixing ratio
};
Atm read_atm(const std::string &path);

// Molecule file: `ID name mass[amu] diameter[A]` per line.
struct MolInfo {
  std::vector<int> id;
  std::vector<std::string> name;
  std::vector<double> mass, diam;
  int find_name(const std::string &n) const;
  int find_id(int i) const;
};
MolInfo read_molfile(const std::string &path);

// Opacity grid header (payload is read by row ranges).
struct OpacityHeader {
  long nmol = 0, ntemp = 0, nlayer = 0, nwave = 0;
  std::vector<int> molid;
  std::vector<double> temp, press, wn;
  long data_offset = 0;  // byte offset of o[0][0][0][0]
};
OpacityHeader read_opacity_header(const std::string &path);
// Copies o[:, :, :, lo:hi] into dst ([L][Nt][M][hi-lo], host memory).
// rows [row0, row0 + nrows) of the grid (a row = one (layer, temperature, molecule)), samples [lo, hi)
void read_opacity_rows(const std::string &path, const OpacityHeader &h, long lo, long hi, long row0,
                       long nrows, double *dst);
// File lists of the transit cfg (`linedb`, `csfile`): comma separated -- one `key value` line
// per file is joined with commas by read_tcfg, code/makecfg.py:87-104 -- or one per line;
// blanks around a name are dropped, blanks INSIDE a path are kept.
std::vector<std::string> split_file_list(const std::string &s);
void read_opacity_block(const std::string &path, const OpacityHeader &h, long lo,
                        long hi, double *dst);

// Cross-section (CIA) file.
struct Cia {
  std::string s1, s2;
  std::vector<double> temp, wn;
  std::vector<double> alpha;  // [ntemp][nwn], cm-1 amagat-2
};
Cia read_cia(const std::string &path);

// Transit-line-information (TLI) file: per database (= one molecule) its
// isotopes with tabulated partition functions, then the transitions sorted by
// wavenumber (per-line fields of doc/BART_user_manual.tex:449-455).
struct TliIso {
  std::string name;
  double mass = 0, ratio = 1;
  std::vector<double> Z;  // [ntemp]
};
struct TliDb {
  std::string name, molecule;
  std::vector<double> temp;
  std::vector<TliIso> iso;
  std::vector<double> wn, elow, gf;  // [nlines], wn ascending
  std::vector<short> isoid;          // index into iso
};
struct Tli {
  double wn_lo = 0, wn_hi = 0;
  std::vector<TliDb> db;
};
Tli read_tli(const std::string &path);

}  // namespace bartrt
