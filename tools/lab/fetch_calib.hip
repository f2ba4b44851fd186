// FETCH_SIZE calibration for the access shapes of the fused-MLP kernels ("the rs kernel fetches 1.39x what
// rc2_ring<384> did, unexplained").  Every kernel reads the SAME 256 MiB once; rocprofv3 --pmc FETCH_SIZE then shows how the
// counter tallies each shape (MI355X_MICROARCH.md: wide coalesced streaming reads are tallied at half their bytes):
//   k_dword_rows : rc2's residual load -- one dword per lane, a wave instruction = two 128-byte row segments (lanes 0-31 row r,
//                  lanes 32-63 row r + 4), rows 1536 bytes apart                      (Rc2Wave::init_o)
//   k_x4_rows    : rs's residual load -- 16 bytes per lane, lane l reads row (l & 31) at column offset 16 (l >> 5) + 32 q
//                  (32 rows x two 16-byte pieces per instruction)                      (RsWave::load_tile)
//   k_x4_stream  : 16 bytes per lane, consecutive lanes consecutive addresses (the y fragments are 16-byte reads too, at a
//                  768-byte lane stride: k_x4_y)
//   build: hipcc --offload-arch=gfx950 -O3 tools/lab/fetch_calib.hip -o tools/lab/fetch_calib ; run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int C = 384;               // floats per row (1536 bytes)
constexpr size_t ROWS = 174762;      // 174762 x 1536 B = 256 MiB
__global__ void k_dword_rows(const float* __restrict__ x, float* out) {  // one wave per 32-row tile
  const int lane = threadIdx.x & 63;
  const size_t tile = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if ((tile + 1) * 32 > ROWS) return;
  const float* base = x + tile * 32 * C + 4 * (lane >> 5) * C + (lane & 31);
  float acc = 0.f;
#pragma unroll 4
  for (int r = 0; r < 16; ++r)
    for (int t = 0; t < C / 32; ++t) acc += base[(size_t)((r & 3) + 8 * (r >> 2)) * C + 32 * t];
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_x4_rows(const float* __restrict__ x, float* out) {
  const int lane = threadIdx.x & 63;
  const size_t tile = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if ((tile + 1) * 32 > ROWS) return;
  const float* base = x + tile * 32 * C + (size_t)(lane & 31) * C + 4 * (lane >> 5);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < C / 32; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc += *(const f32x4*)(base + 32 * t + 8 * q);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
__global__ void k_x4_stream(const float* __restrict__ x, float* out) {
  const size_t n4 = ROWS * C / 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += ((const f32x4*)x)[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
__global__ void k_x4_y(const float* __restrict__ x, float* out) {  // y fragments: lane l reads 16 bytes of row (l & 31) at 16 (l >> 5) + 32 s bytes, rows 768 bytes apart
  const int lane = threadIdx.x & 63;
  const size_t tile = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if ((tile + 1) * 32 > 2 * ROWS) return;
  const float* base = x + tile * 32 * (C / 2) + (size_t)(lane & 31) * (C / 2) + 4 * (lane >> 5);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int s = 0; s < C / 16; ++s) acc += *(const f32x4*)(base + 8 * s);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
int main() {
  float *x, *out;
  const size_t bytes = ROWS * C * 4;
  hipMalloc(&x, bytes + (1 << 20));
  hipMalloc(&out, 64);
  hipMemset(x, 0, bytes + (1 << 20));
  float* big;  // 512 MiB written between the kernels: nothing of x stays in the Infinity Cache
  hipMalloc(&big, 512u << 20);
  const int tiles = (int)(ROWS / 32);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(big, rep, 512u << 20);
    hipLaunchKernelGGL(k_dword_rows, dim3((tiles + 3) / 4), dim3(256), 0, 0, x, out);
    hipMemset(big, rep + 1, 512u << 20);
    hipLaunchKernelGGL(k_x4_rows, dim3((tiles + 3) / 4), dim3(256), 0, 0, x, out);
    hipMemset(big, rep + 2, 512u << 20);
    hipLaunchKernelGGL(k_x4_stream, dim3(2048), dim3(256), 0, 0, x, out);
    hipMemset(big, rep + 3, 512u << 20);
    hipLaunchKernelGGL(k_x4_y, dim3((2 * tiles + 3) / 4), dim3(256), 0, 0, x, out);
  }
  hipDeviceSynchronize();
  printf("each kernel read %.1f MiB\n", bytes / 1048576.0);
  return 0;
}
