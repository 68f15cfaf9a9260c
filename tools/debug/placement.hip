// Where does the dispatcher put the single-wave workgroups of a launch?  (round 5: columns that migrate between SIMDs)
// hipcc --offload-arch=gfx950 -O2 tools/debug/placement.hip -o /tmp/placement && /tmp/placement [blocks] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void probe(unsigned *out, int spin) {
  extern __shared__ double sm[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_readcyclecounter();
  // eight independent fp64 chains: one wave alone keeps its SIMD's fp64 pipe issuing
  double x[8];
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
  for (int i = 0; i < spin; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = fma(x[j], 1.0000001, 1e-9);
  }
  double xs = 0;
  for (int j = 0; j < 8; j++) xs += x[j];
  sm[threadIdx.x] = xs;
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = hw;
    out[4 * blockIdx.x + 1] = xcc;
    out[4 * blockIdx.x + 2] = (unsigned)(t0 >> 4);
    out[4 * blockIdx.x + 3] = (unsigned)((t1 - t0) >> 4);
  }
}
int main(int argc, char **argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 1570, lds = argc > 2 ? atoi(argv[2]) : 16384;
  unsigned *d;
  hipMalloc(&d, 16 * nb);
  std::vector<unsigned> h(4 * nb);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(probe, dim3(nb), dim3(64), lds, 0, d, 4000);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d, 16 * nb, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_simd, per_cu;
  for (int b = 0; b < nb; b++) {
    const unsigned hw = h[4 * b], xcc = h[4 * b + 1] & 15u;
    const unsigned cu = (xcc << 8) | ((hw >> 8) & 0xff), simd = (hw >> 4) & 3;
    per_simd[(cu << 2) | simd]++;
    per_cu[cu]++;
  }
  // how long a wave took: alone on its SIMD / the older (lower workgroup index) and the younger one of a SIMD with two
  {
    std::map<unsigned, std::vector<int>> on;
    for (int b = 0; b < nb; b++) {
      const unsigned hw = h[4 * b], xcc = h[4 * b + 1] & 15u;
      on[(((xcc << 8) | ((hw >> 8) & 0xff)) << 2) | ((hw >> 4) & 3)].push_back(b);
    }
    double s1 = 0, so = 0, sy = 0, e1 = 0, eo = 0, ey = 0; int n1 = 0, n2 = 0;
    unsigned tmin = ~0u;
    for (int b = 0; b < nb; b++) tmin = h[4 * b + 2] < tmin ? h[4 * b + 2] : tmin;
    for (auto &kv : on) {
      if (kv.second.size() == 1) { s1 += h[4 * kv.second[0] + 3]; e1 += h[4 * kv.second[0] + 2] - tmin + h[4 * kv.second[0] + 3]; n1++; }
      if (kv.second.size() == 2) {
        so += h[4 * kv.second[0] + 3]; sy += h[4 * kv.second[1] + 3];
        eo += h[4 * kv.second[0] + 2] - tmin + h[4 * kv.second[0] + 3]; ey += h[4 * kv.second[1] + 2] - tmin + h[4 * kv.second[1] + 3];
        n2++;
      }
    }
    if (n1) printf("  a wave alone on its SIMD: %.0f x16 cycles, ends at %.0f\n", s1 / n1, e1 / n1);
    if (n2) printf("  SIMDs with two: the older wave %.0f x16 cycles (ends at %.0f), the younger %.0f (ends at %.0f)\n", so / n2, eo / n2, sy / n2, ey / n2);
  }
  std::map<int, int> hs, hc;
  for (auto &kv : per_simd) hs[kv.second]++;
  for (auto &kv : per_cu) hc[kv.second]++;
  printf("%d workgroups of one wave, %d bytes of LDS: %zu SIMDs and %zu CUs in use\n", nb, lds, per_simd.size(), per_cu.size());
  for (auto &kv : hs) printf("  SIMDs with %d waves: %d\n", kv.first, kv.second);
  for (auto &kv : hc) printf("  CUs with %d waves: %d\n", kv.first, kv.second);
  printf("first blocks (hw_id, xcc): ");
  for (int b = 0; b < 12; b++) printf("%08x/%u ", h[4 * b], h[4 * b + 1]);
  printf("\n");
  // per CU: the pattern of waves per SIMD
  std::map<std::string, int> pat;
  for (auto &kv : per_cu) {
    char s[32]; int c[4];
    for (int i = 0; i < 4; i++) { auto it = per_simd.find((kv.first << 2) | i); c[i] = it == per_simd.end() ? 0 : it->second; }
    snprintf(s, sizeof s, "%d%d%d%d", c[0], c[1], c[2], c[3]);
    pat[s]++;
  }
  for (auto &kv : pat) printf("  CU pattern (waves on SIMD 0..3) %s: %d\n", kv.first.c_str(), kv.second);
  return 0;
}
