// What can ONE wave issue on a SIMD, and what do two?  fp64 FMA chains (1, 2, 4, 8 independent accumulators per
// lane) timed with the shader clock, one or two waves per SIMD (grid sized to the 1024 SIMDs), plus the other
// instruction kinds of the RT kernel's layer loop (v_mul / v_add / v_rcp_f64, 64-bit selects, ds_read broadcast,
// scalar instructions between the FMAs).
//   hipcc -O2 --offload-arch=gfx950 tools/probe/issue_probe.cpp -o tools/probe/issue_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int CH, int KIND>
__global__ __launch_bounds__(64) void chains(double *out, long long *cyc, int iters, double seed) {
  __shared__ double lds[64];
  lds[threadIdx.x] = seed + threadIdx.x;
  __syncthreads();
  double a[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) a[c] = seed + c + threadIdx.x * 1e-3;
  const double m = 1.0000001, b = 1e-9;
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        if (KIND == 0) a[c] = fma(a[c], m, b);
        if (KIND == 1) a[c] = a[c] * m;
        if (KIND == 2) a[c] = a[c] + b;
        if (KIND == 3) a[c] = __builtin_amdgcn_rcp(a[c]);
        if (KIND == 4) a[c] = a[c] > 1.5 ? a[c] * m : b;     // compare + 64-bit select + mul
        if (KIND == 5) a[c] = fma(a[c], lds[(u * CH + c) & 63], b);   // broadcast LDS read feeding an FMA
        if (KIND == 6) {   // an FMA and two independent scalar moves (what a 64-bit literal costs)
          int t0_, t1_;
          asm volatile("s_mov_b32 %0, 0x12345678\n\ts_mov_b32 %1, 0x3ff00000" : "=s"(t0_), "=s"(t1_));
          a[c] = fma(a[c], m, b);
        }
        if (KIND == 7) {   // an FMA and an independent scalar ALU instruction
          int t;
          asm volatile("s_add_u32 %0, 1, 2" : "=s"(t) : : "scc");
          a[c] = fma(a[c], m, b);
        }
      }
    }
  }
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; c++) s += a[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = c1 - c0; cyc[2 * blockIdx.x + 1] = t1 - t0; }
}

template <int CH, int KIND>
static void run(const char *name, int nwg, double *d_out, long long *d_cyc) {
  const int iters = 256;
  hipLaunchKernelGGL((chains<CH, KIND>), dim3(nwg), dim3(64), 0, 0, d_out, d_cyc, iters, 1.25);
  hipLaunchKernelGGL((chains<CH, KIND>), dim3(nwg), dim3(64), 0, 0, d_out, d_cyc, iters, 1.25);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(2 * nwg);
  (void)hipMemcpy(h.data(), d_cyc, sizeof(long long) * 2 * nwg, hipMemcpyDeviceToHost);
  double sc = 0, sw = 0;
  for (int i = 0; i < nwg; i++) { sc += h[2 * i]; sw += h[2 * i + 1]; }
  const double ops = (double)iters * 16 * CH;
  printf("%-28s chains %d, %4d waves: %6.2f shader cycles per op per wave (%.2f wall ns; clock %.2f GHz)\n", name, CH, nwg,
         sc / nwg / ops, sw / nwg / ops * 10.0, (sc / nwg) / (sw / nwg * 10.0));
}

int main() {
  double *d_out;
  long long *d_cyc;
  (void)hipMalloc(&d_out, sizeof(double) * 64 * 4096);
  (void)hipMalloc(&d_cyc, sizeof(long long) * 2 * 4096);
  for (int nwg : {1024, 2048}) {
    run<1, 0>("v_fma_f64 dependent", nwg, d_out, d_cyc);
    run<2, 0>("v_fma_f64", nwg, d_out, d_cyc);
    run<4, 0>("v_fma_f64", nwg, d_out, d_cyc);
    run<8, 0>("v_fma_f64", nwg, d_out, d_cyc);
    run<4, 1>("v_mul_f64", nwg, d_out, d_cyc);
    run<4, 2>("v_add_f64", nwg, d_out, d_cyc);
    run<4, 3>("v_rcp_f64", nwg, d_out, d_cyc);
    run<4, 4>("cmp + select64 + mul", nwg, d_out, d_cyc);
    run<4, 5>("ds_read broadcast + fma", nwg, d_out, d_cyc);
    run<4, 6>("2 s_mov + fma", nwg, d_out, d_cyc);
    run<4, 7>("s_add + fma", nwg, d_out, d_cyc);
  }
  return 0;
}
