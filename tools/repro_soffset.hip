// Repro for the scalar-offset buffer_store hazard of conv_first_mfma_kernel.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/repro_soffset tools/repro_soffset.hip && /tmp/repro_soffset
// Three forms of the same 16-byte store of lane `l` to plane p(l) = l >> 5 (the first conv's kk) and piece q of 4:
//   FORM 0  everything in the lane (VGPR) offset                                   -- what the kernel does today
//   FORM 1  the wave-uniform part q * qstep in the SCALAR offset operand, the lane part in the VGPR offset
//   FORM 2  the LANE-DEPENDENT plane offset p(l) * kstep handed to the scalar offset operand (what the dropped form did: kk = lane >> 5
//           is not wave-uniform; the compiler must wrap the store in a waterfall loop over the distinct values -- or it reads ONE lane's value)
// Invalid lanes carry the out-of-range lane offset OOB; the raw-buffer range check covers lane offset + immediate only, NOT the scalar offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

template <int FORM>
__global__ void __launch_bounds__(64) store_kernel(float* y, unsigned plane_bytes, int nvalid) {
    const int l = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)blockIdx.x * 32 * 8, 0, 0x7fffffff, 0x00020000);
    const bool valid = (l & 31) < nvalid;
    const unsigned kk = l >> 5, lane_off = (l & 31) * 32u;         // voxel row of 8 floats inside a chunk plane
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 v = {(unsigned)l, (unsigned)q, blockIdx.x, 0x5eedu};
        const unsigned qoff = (unsigned)(q >> 1) * 2u * plane_bytes, koff = kk * plane_bytes;
        if (FORM == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rs, valid ? lane_off + koff + qoff : OOB, 16 * (q & 1), 0);
        if (FORM == 1) __builtin_amdgcn_raw_buffer_store_b128(v, rs, valid ? lane_off + koff : OOB, qoff + 16 * (q & 1), 0);
        if (FORM == 2) __builtin_amdgcn_raw_buffer_store_b128(v, rs, valid ? lane_off : OOB, koff + qoff + 16 * (q & 1), 0);
    }
}

int main() {
    const int blocks = 4096, nvalid = 29;
    const size_t plane_floats = (size_t)blocks * 32 * 8, total = plane_floats * 4;
    float* y; hipMalloc(&y, total * 4);
    std::vector<unsigned> ref(total), got(total);
    int bad[3] = {0, 0, 0};
    for (int form = 0; form < 3; ++form)
        for (int rep = 0; rep < 20; ++rep) {
            hipMemset(y, 0xff, total * 4);
            if (form == 0) store_kernel<0><<<blocks, 64>>>(y, (unsigned)(plane_floats * 4), nvalid);
            if (form == 1) store_kernel<1><<<blocks, 64>>>(y, (unsigned)(plane_floats * 4), nvalid);
            if (form == 2) store_kernel<2><<<blocks, 64>>>(y, (unsigned)(plane_floats * 4), nvalid);
            hipMemcpy(got.data(), y, total * 4, hipMemcpyDeviceToHost);
            if (form == 0 && rep == 0) ref = got;
            else if (memcmp(ref.data(), got.data(), total * 4)) bad[form]++;
        }
    printf("runs differing from FORM 0's first run: lane offsets %d / 20, uniform scalar offset %d / 20, lane-dependent value in the scalar operand %d / 20\n", bad[0], bad[1], bad[2]);
    return 0;
}
