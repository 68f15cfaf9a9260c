// Where does the dispatcher put the single-wave workgroups of a launch?  (round 5: columns that migrate between SIMDs)
// hipcc --offload-arch=gfx950 -O2 tools/debug/placement.hip -o /tmp/placement && /tmp/placement [blocks] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void probe(unsigned *out, int spin) {
  extern __shared__ double sm[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_readcyclecounter();
  double x = threadIdx.x;
  for (int i = 0; i < spin; i++) x = fma(x, 1.0000001, 1e-9);
  sm[threadIdx.x] = x;
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
    hipLaunchKernelGGL(probe, dim3(nb), dim3(64), lds, 0, d, 20000);
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
