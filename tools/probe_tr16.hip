#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k_tr(uint16_t* out) {
    __shared__ uint16_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    // each lane supplies the address of 4 contiguous elements: lane l -> elements [4l, 4l+4)
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + 4 * threadIdx.x));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
__device__ __forceinline__ uint16_t f2bf(float f) { uint32_t u = __float_as_uint(f); return (uint16_t)(u >> 16); }
__global__ void k_mfma(float* out) {
    // A[i][k] = i + 100*k (exact in bf16? use small ints): A[i][k] = (i==k) ; B[k][j] = k*32 + j  (asymmetric) -> D[i][j] = B[i][j] for i<16
    const int l = threadIdx.x, r = l & 31, g = l >> 5;
    bf8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * g + e;
        a[e] = (__bf16)(float)((r == k) ? 1.f : 0.f);         // A[i=r][k]
        b[e] = (__bf16)(float)(k * 8 + (r & 7));              // B[k][j=r] = 8k + (j&7) (small ints, exact)
    }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int e = 0; e < 16; ++e) out[l * 16 + e] = c[e];
}
int main() {
    uint16_t* d; hipMalloc(&d, 256 * 2); k_tr<<<1, 64>>>(d);
    uint16_t h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    float* o; hipMalloc(&o, 64 * 16 * 4); k_mfma<<<1, 64>>>(o);
    float ho[1024]; hipMemcpy(ho, o, 4096, hipMemcpyDeviceToHost);
    // expected D[i][j] = sum_k A[i][k] B[k][j] = (i<16) ? 8i + (j&7) : 0 ; with C/D map col=lane&31, row=(reg&3)+8*(reg>>2)+4*(lane>>5)
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) {
        const int col = l & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
        const float want = row < 16 ? 8.f * row + (col & 7) : 0.f;
        if (ho[l * 16 + e] != want) { if (bad < 8) printf("mfma mismatch lane %d reg %d: got %g want %g\n", l, e, ho[l*16+e], want); ++bad; }
    }
    printf("mfma layout check: %s (%d bad)\n", bad ? "FAIL" : "ok", bad);
    return 0;
}
