/* A C host on the C ABI (include/bartrt.h): the calls BART's worker makes on
 * `transit_module` (reference code/BARTfunc.py:229-234, 363, 406), then a batch.
 *
 *   gcc -std=c99 -Iinclude examples/c_host.c -Lbart_amd -lbartrt -Wl,-rpath,$PWD/bart_amd -o c_host
 *   ./c_host transit.cfg spectrum.txt        (the atmosphere file's own profile is evaluated)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bartrt.h"

static void die(const char *what) {
  fprintf(stderr, "%s: %s\n", what, bartrt_last_error());
  exit(1);
}

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <transit cfg> <output file>\n", argv[0]);
    return 2;
  }
  const char *args[] = {"transit", "-c", argv[1]};
  if (bartrt_init(3, args) < 0) die("bartrt_init");
  const int nwave = bartrt_get_no_samples(), nprof = bartrt_get_nprof();
  if (nwave < 0 || nprof < 0) die("sizes");
  double *wn = malloc(sizeof(double) * nwave), *spec = malloc(sizeof(double) * nwave);
  double *prof = malloc(sizeof(double) * nprof * 3), *batch = malloc(sizeof(double) * nwave * 3);
  unsigned char ok[3];
  if (bartrt_get_waveno_arr(wn, nwave) < 0) die("get_waveno_arr");
  if (bartrt_get_atm_profile(prof, nprof) < 0) die("get_atm_profile");
  if (bartrt_run_transit(prof, nprof, spec, nwave) < 0) die("run_transit");
  /* the same profile, 5 % hotter, 5 % cooler: one batched call */
  const int nlayers = bartrt_get_nlayers();
  for (int w = 1; w < 3; w++) {
    memcpy(prof + (size_t)w * nprof, prof, sizeof(double) * nprof);
    for (int l = 0; l < nlayers; l++) prof[(size_t)w * nprof + l] *= w == 1 ? 1.05 : 0.95;
  }
  if (bartrt_run_transit_batch(prof, 3, nprof, batch, nwave, ok) < 0) die("run_transit_batch");
  FILE *f = fopen(argv[2], "w");
  if (!f) return 1;
  for (int i = 0; i < nwave; i++)
    fprintf(f, "%.17g %.17g %.17g %.17g %.17g\n", wn[i], spec[i], batch[i], batch[nwave + i],
            batch[2 * (size_t)nwave + i]);
  fclose(f);
  printf("%d samples, ok flags %d %d %d\n", nwave, ok[0], ok[1], ok[2]);
  bartrt_free_memory();
  free(wn); free(spec); free(prof); free(batch);
  return 0;
}
