// Probe of v_mfma_f32_4x4x4_16B_f16 on gfx950 (development aid): which lane / register holds which element of the 16
// independent 4 x 4 x 4 products, and how many cycles one instruction takes back to back (for a matrix-core depthwise conv).
//   hipcc --offload-arch=gfx950 -O3 tools/lab/mfma4_probe.hip -o tools/lab/mfma4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const _Float16* A, const _Float16* B, float* D) {
  const int l = threadIdx.x;
  h4 a, b;
  for (int i = 0; i < 4; ++i) a[i] = A[l * 4 + i], b[i] = B[l * 4 + i];
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}
template <int NACC>
__global__ void rate_kernel(float* out, int iters, long long* cyc) {
  h4 a = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)2.f}, b = a;
  f4 c[NACC];
  for (int k = 0; k < NACC; ++k) c[k] = f4{0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c[k], 0, 0, 0);
  const long long t1 = clock64();
  float s = 0;
  for (int k = 0; k < NACC; ++k) s += c[k][0] + c[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  _Float16 hA[256], hB[256];
  float hD[256];
  // A element value encodes (lane, i): A = lane + i / 8; B = 1 at one (lane, i) at a time would be slow: use two runs with structured data
  _Float16 *dA, *dB;
  float* dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024);
  // run 1: B = all ones -> D[lane][r] = sum_k A_block[row?][k]: tells which A lane feeds which D register / lane
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) hA[l * 4 + i] = (_Float16)(l == 5 && i == 2 ? 1.f : 0.f), hB[l * 4 + i] = (_Float16)1.f;
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  printf("A[lane 5][i 2] = 1, B = 1: nonzero D at"); for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hD[l * 4 + r] != 0) printf(" (lane %d reg %d)=%g", l, r, hD[l * 4 + r]); printf("\n");
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) hA[l * 4 + i] = (_Float16)1.f, hB[l * 4 + i] = (_Float16)(l == 6 && i == 1 ? 1.f : 0.f);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  printf("B[lane 6][i 1] = 1, A = 1: nonzero D at"); for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hD[l * 4 + r] != 0) printf(" (lane %d reg %d)=%g", l, r, hD[l * 4 + r]); printf("\n");
  // both single: A[lane 5][i 2] = 1 and B[lane L][i I] = 1 for the lanes of block 1: which (L, I) pair produces output, and where
  for (int L = 4; L < 8; ++L) for (int I = 0; I < 4; ++I) {
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) hA[l * 4 + i] = (_Float16)(l == 5 && i == 2 ? 1.f : 0.f), hB[l * 4 + i] = (_Float16)(l == L && i == I ? 1.f : 0.f);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hD[l * 4 + r] != 0) printf("  A(5,2) x B(%d,%d) -> D(lane %d, reg %d)\n", L, I, l, r);
  }
  float* dout; long long* dcyc; long long cyc;
  hipMalloc(&dout, 256 * 4 * 1024); hipMalloc(&dcyc, 8);
  rate_kernel<1><<<1, 64>>>(dout, 10000, dcyc); hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  printf("1 accumulator  (dependent chain): %.2f cycles per MFMA\n", cyc / 10000.0);
  rate_kernel<8><<<1, 64>>>(dout, 10000, dcyc); hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  printf("8 accumulators (independent)    : %.2f cycles per MFMA\n", cyc / 80000.0);
  rate_kernel<8><<<1, 256>>>(dout, 10000, dcyc); hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  printf("8 accumulators, 4 waves (1 per SIMD): %.2f cycles per MFMA per wave\n", cyc / 80000.0);
  rate_kernel<8><<<1, 512>>>(dout, 10000, dcyc); hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
  printf("8 accumulators, 8 waves (2 per SIMD): %.2f cycles per MFMA per wave\n", cyc / 80000.0);
  return 0;
}
