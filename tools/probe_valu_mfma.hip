// Probe: do fp32 MFMAs and plain VALU instructions of ANOTHER wave on the same SIMD overlap on gfx950?
// Workgroup = 8 waves (2 per SIMD): waves 0-3 issue MFMAs, waves 4-7 issue dependent-free VALU FMAs.
// mode bit0 = MFMA waves active, bit1 = VALU waves active, bit2 = bf16 MFMA instead of fp32.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_valu_mfma.hip -o gpurun_out/probe_valu_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

template <int mode>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        float a = threadIdx.x * 1e-3f, b = 1.0f;
        if (mode & 8) {     // same wave: 4 MFMAs interleaved with 16 (mode 8) or 32 (mode 24) independent VALU FMAs
            float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
            const float m = 1.0001f, q = 0.5f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < ((mode & 16) ? 2 : 1); ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                x0 = __builtin_fmaf(x0, m, q); x1 = __builtin_fmaf(x1, m, q); x2 = __builtin_fmaf(x2, m, q); x3 = __builtin_fmaf(x3, m, q);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                x4 = __builtin_fmaf(x4, m, q); x5 = __builtin_fmaf(x5, m, q); x6 = __builtin_fmaf(x6, m, q); x7 = __builtin_fmaf(x7, m, q);
                if (!(mode & 16)) {
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
                x0 = __builtin_fmaf(x0, m, q); x1 = __builtin_fmaf(x1, m, q); x2 = __builtin_fmaf(x2, m, q); x3 = __builtin_fmaf(x3, m, q);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
                x4 = __builtin_fmaf(x4, m, q); x5 = __builtin_fmaf(x5, m, q); x6 = __builtin_fmaf(x6, m, q); x7 = __builtin_fmaf(x7, m, q);
                } else {
                x0 = __builtin_fmaf(x0, m, q); x1 = __builtin_fmaf(x1, m, q); x2 = __builtin_fmaf(x2, m, q); x3 = __builtin_fmaf(x3, m, q);
                x4 = __builtin_fmaf(x4, m, q); x5 = __builtin_fmaf(x5, m, q); x6 = __builtin_fmaf(x6, m, q); x7 = __builtin_fmaf(x7, m, q);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
                }
                }
            }
            float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
            for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
            out[blockIdx.x * 512 + threadIdx.x] = s;
            return;
        }
        bf16x4 ab = {1, 2, 3, 4};
        for (int i = 0; i < iters; ++i) {
            if (mode & 4) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ab, ab, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ab, ab, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ab, ab, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ab, ab, c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
            }
        }
        float s = 0;
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
        const float m = 1.0001f, q = 0.5f;
        for (int i = 0; i < iters; ++i) {   // 16 independent VALU FMAs per iteration (vs 4 MFMAs = 256 cycles)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                x0 = __builtin_fmaf(x0, m, q); x1 = __builtin_fmaf(x1, m, q); x2 = __builtin_fmaf(x2, m, q); x3 = __builtin_fmaf(x3, m, q);
                x4 = __builtin_fmaf(x4, m, q); x5 = __builtin_fmaf(x5, m, q); x6 = __builtin_fmaf(x6, m, q); x7 = __builtin_fmaf(x7, m, q);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 4096 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    for (int mode : {1, 2, 3, 5, 7, 9, 25, 1, 2, 3}) {
        auto run = [&]() {
            switch (mode) {
                case 1: k<1><<<blocks, 512>>>(out, iters); break;
                case 2: k<2><<<blocks, 512>>>(out, iters); break;
                case 3: k<3><<<blocks, 512>>>(out, iters); break;
                case 5: k<5><<<blocks, 512>>>(out, iters); break;
                case 7: k<7><<<blocks, 512>>>(out, iters); break;
                case 9: k<9><<<blocks, 512>>>(out, iters); break;
                case 25: k<25><<<blocks, 512>>>(out, iters); break;
            }
        };
        run(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); run(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s%s%s): %.3f ms  -> %.1f cycles@2.4GHz per iteration (4 MFMA / 16 VALU)\n", mode, mode & 1 ? "mfma " : "", mode & 2 ? "valu " : "",
               mode & 4 ? "bf16" : "f32", ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
