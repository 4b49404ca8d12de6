// Root cause of round 5's "scalar-offset buffer_store" non-reproducibility in conv_first_mfma_kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/repro_soffset tools/repro_soffset.hip && tools/_bin/repro_soffset
//
// The hazard (gfx9 ISA guides, "manually inserted wait states"): a VMEM store of MORE THAN 64 BITS reads its data registers after issue; a VALU
// instruction that overwrites them needs wait states in between.  The guides exempt stores whose offset comes from an SGPR ("BUFFER_STORE_* operations
// that use an SGPR for offset do not require any wait states"), and LLVM's GCNHazardRecognizer::createsVALUHazard implements exactly that: it inserts
// the s_nop only when the soffset operand is NOT a register.  On gfx950 the exemption does not hold: with the wave-uniform plane offset in the scalar
// offset operand hipcc emitted
//     buffer_store_dwordx4 v[2:5], v94, s[8:11], s48 offen
//     v_pk_add_f32 v[4:5], v[16:17], v[32:33]            <- overwrites half of the store's data, no wait state
// and the stored rows were run-to-run different (tools/repro_first_soffset.sh: -DE3_FIRST_SOFFSET=1 20 distinct results in 20 runs; the same offsets in
// the lane operand, or a waterfall loop around the store, are bit-stable).  This file reproduces it in isolation with the instruction sequence pinned by
// inline asm: eight 16-byte stores per lane, each followed by a VALU write of one of its data registers after NOPS wait states.
//     forms A, A1, B, B3   SGPR soffset, 0 / 1 / 2 / 3 wait states between the store and the VALU write (A is what the compiler generated)
//     forms C, D           soffset = 0 (the DOCUMENTED hazard; hipcc never emits C, it inserts the s_nop), 0 / 2 wait states
// Result on MI355X: profiles/r06_soffset_hazard.md.
// The constraint the library keeps (tests/test_isa.py::test_no_wide_store_with_scalar_offset_is_followed_by_a_write_of_its_data): no
// buffer_store_dwordx3/x4 with an SGPR soffset whose data registers a VALU instruction writes within the next three instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned POISON = 0xdeadbeefu;

template <bool SOFF, int NOPS>
__global__ void __launch_bounds__(256) store_kernel(unsigned* y, unsigned plane_bytes) {
    const unsigned l = threadIdx.x + blockIdx.x * 256u;
    i32x4 rs;
    rs[0] = (int)(unsigned)(size_t)y; rs[1] = (int)((size_t)y >> 32) & 0xffff; rs[2] = 0x7fffffff; rs[3] = 0x00020000;
    rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned so = __builtin_amdgcn_readfirstlane(SOFF ? j * plane_bytes : 0u);
        const unsigned vo = l * 16u + (SOFF ? 0u : j * plane_bytes);
#define E3_REPRO_BODY(SOFFSET_TEXT)                                                                                                 \
        asm volatile(                                                                                                                \
            "v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\t"                                 \
            "s_nop 4\n\t"                                                                                                            \
            "buffer_store_dwordx4 v[20:23], %4, %5, " SOFFSET_TEXT " offen\n\t"                                                       \
            ".rept %8\n\ts_nop 0\n\t.endr\n\t"                                                                                     \
            "v_mov_b32 v22, %7\n\t"                                                                                                  \
            "v_mov_b32 v21, %7"                                                                                                      \
            :: "v"(l), "v"((unsigned)j), "v"(l ^ 0x5a5a5a5au), "v"(0x5eedu + j), "v"(vo), "s"(rs), "s"(so), "v"(POISON), "n"(NOPS) \
            : "v20", "v21", "v22", "v23", "memory")
        if constexpr (SOFF) E3_REPRO_BODY("%6");      // the offset of plane j in an SGPR
        else E3_REPRO_BODY("0");                      // the literal zero of the encoding (the documented hazard)
#undef E3_REPRO_BODY
    }
}

int main() {
    const unsigned blocks = 8192, lanes = blocks * 256;
    const size_t plane_words = (size_t)lanes * 4, total = plane_words * 8;
    unsigned* y; hipMalloc(&y, total * 4);
    std::vector<unsigned> got(total);
    const char* names[6] = {"A  SGPR soffset, 0 wait states ", "A1 SGPR soffset, 1 wait state  ", "B  SGPR soffset, 2 wait states ", "B3 SGPR soffset, 3 wait states ",
                            "C  soffset = 0,   0 wait states ", "D  soffset = 0,   2 wait states "};
    for (int form = 0; form < 6; ++form) {
        size_t poisoned = 0, wrong = 0;
        for (int rep = 0; rep < 10; ++rep) {
            hipMemset(y, 0, total * 4);
            const unsigned pb = (unsigned)(plane_words * 4);
            if (form == 0) store_kernel<true, 0><<<blocks, 256>>>(y, pb);
            if (form == 1) store_kernel<true, 1><<<blocks, 256>>>(y, pb);
            if (form == 2) store_kernel<true, 2><<<blocks, 256>>>(y, pb);
            if (form == 3) store_kernel<true, 3><<<blocks, 256>>>(y, pb);
            if (form == 4) store_kernel<false, 0><<<blocks, 256>>>(y, pb);
            if (form == 5) store_kernel<false, 2><<<blocks, 256>>>(y, pb);
            hipMemcpy(got.data(), y, total * 4, hipMemcpyDeviceToHost);
            for (unsigned j = 0; j < 8; ++j)
                for (unsigned l = 0; l < lanes; ++l) {
                    const unsigned* r = &got[j * plane_words + (size_t)l * 4];
                    const bool ok = r[0] == l && r[1] == j && r[2] == (l ^ 0x5a5a5a5au) && r[3] == 0x5eedu + j;
                    if (!ok) { ++wrong; if (r[1] == POISON || r[2] == POISON) ++poisoned; }
                }
        }
        printf("form %s: %zu of %zu rows wrong over 10 runs (%zu carry the value written AFTER the store)\n", names[form], wrong, (size_t)10 * 8 * lanes, poisoned);
    }
    return 0;
}
