// Which compute units does a CU-masked stream use?  hipExtStreamCreateWithCUMask takes a bit vector; this probe launches a
// kernel on streams with a few masks and prints the (XCC, SE, CU) of every workgroup (HW_REG_XCC_ID, HW_REG_HW_ID).
//   hipcc --offload-arch=gfx950 -O2 tools/lab/cumask_probe.hip -o tools/lab/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <set>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
  const long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}
static void run(const char* name, const std::vector<int>& bits) {
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b : bits) mask[b / 32] |= 1u << (b % 32);
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
  const int nb = 2048;
  unsigned* out;
  hipMalloc(&out, nb * 8);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, s, out, 20000);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(2 * nb);
  hipMemcpy(h.data(), out, nb * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int i = 0; i < nb; ++i) per_xcc[h[2 * i] & 0xf].insert(h[2 * i + 1] & 0xfff0);  // HW_ID: wave/simd in the low bits, cu_id [11:8], sh [12], se [15:13]
  printf("%-28s %zu bits ->", name, bits.size());
  int tot = 0;
  for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
  printf("  (%d distinct CUs)\n", tot);
  hipFree(out);
  hipStreamDestroy(s);
}
int main() {
  std::vector<int> all, s8, s4, first32, first64, low8of64;
  for (int i = 0; i < 256; ++i) all.push_back(i);
  for (int i = 0; i < 256; i += 8) s8.push_back(i);
  for (int i = 0; i < 256; i += 4) s4.push_back(i);
  for (int i = 0; i < 32; ++i) first32.push_back(i);
  for (int i = 0; i < 64; ++i) first64.push_back(i);
  std::vector<int> not0;
  for (int i = 0; i < 256; ++i) if (i % 8 != 0) not0.push_back(i);
  run("all 256 bits", all);
  run("bits 0, 8, 16, ... (32)", s8);
  run("bits 0, 4, 8, ... (64)", s4);
  run("bits 0 .. 31", first32);
  run("bits 0 .. 63", first64);
  run("bits i % 8 != 0 (224)", not0);
  return 0;
}
