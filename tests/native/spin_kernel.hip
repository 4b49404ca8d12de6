// Test infrastructure (not product): a kernel that does nothing but stay resident -- `blocks` workgroups of 256 threads, each holding
// `lds_bytes` of LDS, spinning for `ticks` ticks of the 100 MHz wall clock.  Stands in for a collective's (RCCL's) resident workgroups in
// the data-parallel robustness tests and in tools/probe_foreign_waves.py.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void spin_kernel(long long ticks, int* sink) {
    extern __shared__ int lds[];
    const long long t0 = (long long)wall_clock64();
    int it = 0;
    while ((long long)wall_clock64() - t0 < ticks) ++it;
    if (ticks < 0) { lds[threadIdx.x] = it; sink[0] = lds[(threadIdx.x + 1) & 255]; }     // (never: keeps the LDS allocation and the loop alive)
}

extern "C" int spin_launch(void* stream, int blocks, int lds_bytes, long long ticks, int* sink) {
    if (lds_bytes > 64 * 1024) lds_bytes = 64 * 1024;
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), (size_t)(lds_bytes < 1024 ? 1024 : lds_bytes), (hipStream_t)stream, ticks, sink);
    return (int)hipGetLastError();
}
