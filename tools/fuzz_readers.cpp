// Sanitizer driver for the host-side readers (bart_amd/csrc/io.cpp): `fuzz_readers <kind> <file>` parses one file
// and prints "ok" or "IoError: ..."; anything else (a sanitizer report, a crash, an uncaught exception) is a defect.
// Built with -fsanitize=address,undefined by tools/fuzz_readers.py and tests/test_readers_fuzz.py.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../bart_amd/csrc/io.hpp"

using namespace bartrt;

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const std::string kind = argv[1], path = argv[2];
  try {
    if (kind == "cfg") { auto c = read_tcfg(path); (void)cfg_list(c, "raygrid"); }
    else if (kind == "atm") (void)read_atm(path);
    else if (kind == "mol") (void)read_molfile(path);
    else if (kind == "cia") {
      Cia c = read_cia(path);
      if (c.alpha.size() != c.temp.size() * c.wn.size()) { std::puts("BAD: inconsistent table"); return 3; }
    } else if (kind == "tli") (void)read_tli(path);
    else if (kind == "opacity") {
      OpacityHeader h = read_opacity_header(path);
      std::vector<double> row((size_t)h.nwave);
      if (h.nlayer * h.ntemp * h.nmol > 0) read_opacity_rows(path, h, 0, h.nwave, 0, 1, row.data());
    } else return 2;
  } catch (const IoError &e) {
    std::printf("IoError: %s\n", e.msg.c_str());
    return 0;
  }
  std::puts("ok");
  return 0;
}
