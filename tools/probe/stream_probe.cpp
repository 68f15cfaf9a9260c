// What does this GPU deliver on the ten-walker launch's compulsory bytes when nothing else is asked of it?  A table the
// size of the headline grid (864 MB); one pass reads ~300 MB of it as whole (layer, T plane) rows of 320 kB picked
// at random, 16 bytes per lane, and adds them up -- the RT kernel's unique traffic without its arithmetic.  Cold
// (after a 1 GiB sweep of another buffer) and warm (the same rows again, and the bench's cycle of 16 row sets).
//   hipcc -O2 --offload-arch=gfx950 tools/probe/stream_probe.cpp -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

__global__ __launch_bounds__(256) void read_rows(const double2 *tab, const long *row_off, int row_d2, double *out) {
  // one workgroup per (row, 2 kB piece): 128 lanes x 16 B; 256 lanes take two pieces
  const long base = row_off[blockIdx.y];
  double2 acc = {0.0, 0.0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < row_d2; i += gridDim.x * 256) {
    const double2 v = tab[base + i];
    acc.x += v.x; acc.y += v.y;
  }
  if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

__global__ void sweep(double *p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] += 1.0;
}

int main() {
  const long table_bytes = 864000000L, row_bytes = 320000L, nrows_total = table_bytes / row_bytes;
  const int nrows = 940;   // ~300 MB
  double2 *tab; double *out, *scratch; long *d_off;
  hipMalloc(&tab, table_bytes); hipMemset(tab, 0, table_bytes);
  hipMalloc(&out, 8);
  const long nscr = (1L << 30) / 8;
  hipMalloc(&scratch, nscr * 8); hipMemset(scratch, 0, nscr * 8);
  std::mt19937 rng(7);
  std::vector<std::vector<long>> sets(16, std::vector<long>(nrows));
  for (auto &s : sets)
    for (auto &r : s) r = (long)(rng() % nrows_total) * (row_bytes / 16);
  hipMalloc(&d_off, sizeof(long) * nrows * 16);
  for (int k = 0; k < 16; k++) hipMemcpy(d_off + (long)k * nrows, sets[k].data(), sizeof(long) * nrows, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int row_d2 = (int)(row_bytes / 16);
  auto pass = [&](int k) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(read_rows, dim3(8, nrows), dim3(256), 0, 0, tab, d_off + (long)k * nrows, row_d2, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return (double)ms;
  };
  const double mb = nrows * (double)row_bytes / 1e6;
  for (int i = 0; i < 3; i++) pass(0);
  double cold = 0, warm_same = 0, warm_cycle = 0;
  for (int i = 0; i < 8; i++) {
    hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, scratch, nscr); hipDeviceSynchronize();
    cold += pass(i % 16);
  }
  for (int i = 0; i < 20; i++) warm_same += pass(3);
  for (int i = 0; i < 64; i++) warm_cycle += pass(i % 16);
  printf("%.0f MB per pass as %d rows of 320 kB out of an 864 MB table\n", mb, nrows);
  printf("cold (after a 1 GiB sweep):      %.1f us  %.2f TB/s\n", cold / 8 * 1e3, mb / (cold / 8) / 1e6 * 1e3);
  printf("warm, the same rows again:       %.1f us  %.2f TB/s\n", warm_same / 20 * 1e3, mb / (warm_same / 20) / 1e6 * 1e3);
  printf("warm, 16 row sets in a cycle:    %.1f us  %.2f TB/s\n", warm_cycle / 64 * 1e3, mb / (warm_cycle / 64) / 1e6 * 1e3);
  return 0;
}
