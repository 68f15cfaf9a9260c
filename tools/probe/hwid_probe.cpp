// Where does the dispatcher put the one-wave workgroups of a ten-walker launch?  1570 workgroups of 64 lanes, the
// RT kernel's resources (dynamic LDS, a register count that allows two waves per SIMD), each spinning for a fixed
// time; every wave records its XCC / SE / CU / SIMD and its start and end clock.
//   hipcc -O2 --offload-arch=gfx950 tools/probe/hwid_probe.cpp -o /tmp/hwid_probe && /tmp/hwid_probe [nwg] [lds_bytes] [spin_us]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

struct Rec { unsigned hwid, xcc; unsigned long long t0, t1; };

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(Rec *out, long spin_ticks, double *sink) {
  extern __shared__ double lds[];
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double a = threadIdx.x;
  unsigned long long t1 = t0;
  while ((long)(t1 - t0) < spin_ticks) {
    for (int i = 0; i < 64; i++) a = a * 1.0000001 + 1e-9;
    t1 = __builtin_readcyclecounter();
  }
  lds[threadIdx.x] = a;
  if (threadIdx.x == 0) {
    out[blockIdx.x] = {hw, xcc, t0, t1};
    if (a == 12345.0) sink[0] = lds[1];
  }
}

int main(int argc, char **argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 1578;
  const size_t lds = argc > 2 ? atol(argv[2]) : 14000;
  const double spin_us = argc > 3 ? atof(argv[3]) : 30.0;
  Rec *d;
  double *sink;
  hipMalloc(&d, sizeof(Rec) * nwg);
  hipMalloc(&sink, 8);
  const long ticks = (long)(spin_us * 100.0);   // s_memtime / readcyclecounter: 100 MHz
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(64), lds, 0, d, ticks, sink);
    hipDeviceSynchronize();
  }
  std::vector<Rec> h(nwg);
  hipMemcpy(h.data(), d, sizeof(Rec) * nwg, hipMemcpyDeviceToHost);
  std::map<unsigned long long, int> per_simd, per_cu;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto &r : h) {
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    const unsigned simd = (r.hwid >> 4) & 3, cu = (r.hwid >> 8) & 15, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7;
    const unsigned long long cuk = ((unsigned long long)(r.xcc & 15) << 16) | (se << 8) | (sh << 4) | cu;
    per_cu[cuk]++;
    per_simd[(cuk << 2) | simd]++;
    tmin = std::min(tmin, r.t0);
    tmax = std::max(tmax, r.t1);
  }
  std::map<int, int> hs, hc;
  for (auto &kv : per_simd) hs[kv.second]++;
  for (auto &kv : per_cu) hc[kv.second]++;
  printf("workgroups %d, LDS %zu B, spin %.1f us: %zu CUs and %zu SIMDs used; span %.1f us\n", nwg, lds, spin_us,
         per_cu.size(), per_simd.size(), (tmax - tmin) / 100.0);
  printf("waves per SIMD -> number of SIMDs:");
  for (auto &kv : hs) printf("  %d: %d", kv.first, kv.second);
  printf("\nwaves per CU -> number of CUs:");
  for (auto &kv : hc) printf("  %d: %d", kv.first, kv.second);
  // start-time spread: how many waves started later than 5 us after the first
  int late = 0;
  for (auto &r : h) late += (r.t0 - tmin) > 500;
  printf("\nwaves that started more than 5 us after the first: %d\n", late);
  return 0;
}
