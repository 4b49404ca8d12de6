// Probe: HIP runtime interop with torch + f32 MFMA operand/accumulator layout on gfx950.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C[32x32] = A[32xK] * B[Kx32], K multiple of 2. One wave.
__global__ void probe32(const float* A, const float* B, float* C, int K) {
  int l = threadIdx.x; int i = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) {
    float a = A[i * K + k + h];
    float b = B[(k + h) * 32 + i];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    C[row * 32 + i] = acc[r];
  }
}
// 16x16x4 variant
__global__ void probe16(const float* A, const float* B, float* C, int K) {
  int l = threadIdx.x; int i = l & 15, h = l >> 4;
  f32x4 acc = {0};
  for (int k = 0; k < K; k += 4) {
    float a = A[i * K + k + h];
    float b = B[(k + h) * 16 + i];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) C[(h * 4 + r) * 16 + i] = acc[r];
}
extern "C" int probe_run(const float* A, const float* B, float* C, int K, int which, void* stream) {
  if (which == 32) hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C, K);
  else hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C, K);
  return (int)hipGetLastError();
}
