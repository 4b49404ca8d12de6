// 3x3x3 stride-1 convolution as Winograd F(2x2x4, 3x3x3) on the fp32 matrix cores: F(2,3) along D and H, F(4,3) along W -- 96 multiplies per
// 2x2x4 output tile and (ci, co) pair = 6 per output voxel, where conv_wino.hip's F(2x2x2) tiles spend 8 and the direct kernels 27.
// (torch.nn.Conv3d(k=3, padding=1) in elektronn3's conv3 blocks, unet.py:131-149.)
//
// Who takes it (conv_wino_layout() == 2, flag CF_WINO4 set by the caller): the eval-mode forward with the folded BatchNorm epilogue (Predictor)
// and every DATA GRADIENT.  The F(4,3) transforms carry the constants 4, 5, 8 and cost 2.5x the conv error of F(2,3) (4.5e-7 instead of 1.8e-7
// rel-L2 at 64 channels): far inside the forward tolerance, and invisible in a data gradient -- but a train-mode FORWARD on these tiles flips
// enough ReLU / arg-max decisions to push encoder weight gradients over the 1e-2 bound of the full-size digest test
// (profiles/r05_f224_emulation.md), so the training forward stays on F(2x2x2).
//
// Work decomposition (one workgroup = 4 waves, ONE wave per SIMD):
//   * brick = 2x2x4 tiles = 4x4x16 output voxels (6x6x18 halo, the brick of conv_wino.hip) x 32 output channels;
//   * the 96 Winograd positions (pd 4, ph 4, pw 6) are 96 GEMMs  M = 32 channels, N = 16 tiles, K = Cin  on v_mfma_f32_16x16x4_f32 with the
//     WEIGHTS as the A operand: D[co][tile], lane l holds tile l & 15 and the channels 4 (l >> 4) .. + 3 of a 16-channel half, the packer permutes the
//     channels of a 32-channel tile so that a lane's two halves are 8 consecutive channels -> the brick leaves as 16-byte stores.  Wave w owns the 24
//     positions with pd = w: 24 x 2 accumulator quads = 192 registers;
//   * K is walked in chunks of 8 channels.  The raw halo is staged by LDS-DMA (buffer_load_dwordx4 ... lds, 6 x 1 KB per wave and chunk, the bank
//     layout is produced on the source side, zero padding = range check), double buffered across chunks AND bricks; lane (tile, kk) reads its 4 x 6
//     window of channels 2 kk, 2 kk + 1 from the two d-planes its pd needs (D pass of B^T while reading), does the H (F(2,3)) and W (F(4,3)) passes
//     on float2 and feeds the result straight to the MFMAs: the transformed tile never touches LDS.  The reads of chunk c + 1 are issued under the
//     last MFMAs of chunk c, the weights (24 x 16 bytes per lane and chunk, from L2) one chunk ahead into the registers whose MFMAs are issued;
//   * epilogue: A^T over (pw: 6 -> 4, ph: 4 -> 2) in registers, over pd through LDS (wave w then owns oh = w >> 1 and the two ow = 2 (w & 1), + 1 of
//     every tile), bias / folded BN + ReLU / running Welford statistics per lane (8 channels of one tile, one record per workgroup) / 16-byte stores.
#include <type_traits>
#include "kernels.h"
#include "brick_order.h"

#ifndef E3_W4_NT_MIN_MB
#define E3_W4_NT_MIN_MB 128   // non-temporal output stores of the training forms (rows) from this tensor size on (same-box A/B: 0 and 128 alike, step 11.205 -> 11.105 ms)
#endif
#ifndef E3_W4_ABL
#define E3_W4_ABL 0       // developer builds (-DE3_TIMING): bit mask of pieces left out (timing experiments, wrong results)
#endif

namespace {

constexpr int V_ROW = 20;                                   // slots per halo row: zw -> (zw & 3) * 5 + (zw >> 2)  (18 of 20 used)
constexpr int V_RPLANE = 128 * 8 + 16;                      // floats of one raw d-plane: 6 rows x 20 slots of 32 bytes, padded to 4 DMA pieces of 1 KB, + 64 B skew
constexpr int V_RBUF = 6 * V_RPLANE;                        // 6 raw planes: 24.4 KB per stage buffer
constexpr int V_EX = 4 * 16 * 64 * 4;                       // epilogue exchange [pd][(oh, ow, half)][lane][4] floats (64 KB)
constexpr int V_SCR = 4 * 32 * 3;                           // cross-wave merge of the statistics
constexpr int V_RUN = 9 * 256;                              // running statistics of every thread: n, mean[4], M2[4]  ([k][thread]; BNRED: 4 + 4 sums)
constexpr int V_POOLX = 4 * 64 * 8;                         // fused max-pool: [wave][lane][8 channels] (8 KB)
constexpr int V_HEADW = 160;                                // fused head: weights [4][32] + biases [4] (padded)
constexpr int V_KST = 2 * 96 + 128;                          // bias / folded scale / folded shift of the workgroup's 32 channels, two slots; BNRED: scale / shift / mean / invstd of the unit in front
constexpr int V_LC = 8 * 256;                                // the chunk loop's lane constants, parked: [thread][8]
constexpr int V_LDS_FLOATS = 3 * V_RBUF + V_EX + V_SCR + V_RUN + V_KST + V_LC;   // 158.0 KB of the CU's 160 KiB (163 840 B): one workgroup per CU (the fused pool / head scratch lives in the statistics' region)
static_assert(V_POOLX + V_HEADW <= V_RUN && V_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
typedef __attribute__((address_space(3))) void* lds_ptr_v;
typedef const volatile __attribute__((address_space(3))) f32x2v* lds_cv2;      // (volatile: see read_window)

// Layout of a raw plane: halo voxel (zh, zw) -> slot ((zh & 1) * 3 + (zh >> 1)) * 20 + (zw & 3) * 5 + (zw >> 2) of 32 bytes (8 channels): the stride-2
// tile origins along h and the stride-4 origins along w are consecutive slots; the 16-byte half q of a voxel sits at q ^ ((zh >> 1) & 1), and planes
// are 64 B mod 256 B apart -- so the 32 lanes of a ds_read_b64 group (2 x 2 x 4 tiles x 2 channel pairs) cover all 64 banks.

// POOL (AFF, inference): the 2x2x2 ceil-mode max-pool of the output in the epilogue (ConvArgs::pool_out)
// HEAD (AFF, inference, 32 output channels): the 1x1x1 head (+ softmax) on the activations in registers instead of storing them (ConvArgs::head_*)
typedef BrickStep W4PArgs;      // the logical brick order and the digits of the step between a workgroup's bricks (gridDim / 8 logical bricks): brick_order.h

// BNRED (data gradients): the store phase also takes the REDUCE sums of the BatchNorm backward of the unit in front (ConvArgs::br_*)
template <bool AFF, bool POOL = false, bool HEAD = false, bool BNRED = false>
__global__ __launch_bounds__(256, 1) void conv3_wino4_kernel(const ConvArgs a, const unsigned nblk, const int wgstats, const W4PArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NCH = a.Cin >> 3;
    constexpr unsigned OOB = 0x80000000u;       // buffer offset beyond every descriptor below: loads return 0, stores are dropped
    // arguments that only the per-brick set-up and the epilogue need are re-read from the kernarg segment there (scalar loads) instead of
    // occupying SGPRs across the chunk loop
    typedef const __attribute__((address_space(4))) ConvArgs* KArgs;
    auto KA = []() -> KArgs { KArgs q = (KArgs)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(q)); return q; };
    // channel-chunked input (ConvArgs::x_chunk): a voxel's row is the 8 channels of ONE chunk (xl = 8) and the chunks are x_chunk floats apart
    const int D = a.D, H = a.H, W = a.W, xl = a.x_chunk ? 8 : a.x_ldc;
    const unsigned plane_xb = (unsigned)((size_t)H * W * xl * 4);
    const int chunk_xb = a.x_chunk ? (int)(a.x_chunk * 4) : 32;

    // ---- staging by LDS-DMA (1 KB per wave-instruction, lane i lands at base + 16 i): a raw d-plane of the halo is four such pieces; wave w
    // issues quarter w of each of the six planes.  The layout of a plane is produced on the SOURCE side: lane i of quarter w asks for the 16
    // bytes that belong at piece g = 64 w + i.
    // (the lane constants of the chunk loop -- DMA source offset and validity bits, the four window-read addresses, the weight offset -- are taken afresh
    // from the lane index at the top of every brick: kept across the epilogue they were spilled, and a scratch reload in there waits -- the memory counter
    // retires in order -- for every request in flight)
    unsigned col_rel, col_bits;
    int rdA[2], rdB[2], b_voff;
    const int pA = wave == 0 ? 0 : (wave == 2 ? 2 : 1), pB = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    // (round 6: computed ONCE and parked in LDS -- two ds_read_b128 per brick instead of ~60 VALU instructions with three constant divisions in the epilogue)
    auto lane_consts_compute = [&]() {
        int fl;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(fl));
        const int g = wave * 64 + fl, slot = g >> 1, qd = g & 1;
        const int rs = slot / V_ROW, sw = slot % V_ROW;
        const int cls = sw / 5, idx = sw % 5;
        const int zh = 2 * (rs % 3) + rs / 3, zw = 4 * idx + cls;
        const bool used = rs < 6 && zw < 18;
        const int q = qd ^ ((rs % 3) & 1);
        col_rel = (unsigned)(((zh * W + zw) * xl + 4 * q) * 4);
        if (E3_W4_ABL & 1024) col_rel = (unsigned)(((zh * W + 1) * xl + zw * 8 + 4 * q) * 4);      // (timing only: a quarter's 16-byte pieces as contiguous runs of the h row)
        col_bits = used ? (1u << (6 + zh)) | (1u << (12 + zw)) : 0xffffffffu;     // (all-ones never matches: the unused slots get zeros)
        // read plan of lane (tile tl, channel pair kk): window rows h = 0, 1 have zh/2 = tth, rows 2, 3 have tth + 1 (the XOR of the 16-byte half follows)
        const int ftl = fl & 15, fkk = fl >> 4, fttd = ftl >> 3, ftth = (ftl >> 2) & 1, fttw = ftl & 3;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int r = ((ftth + hh) * V_ROW + fttw) * 8 + 2 * (fkk ^ (2 * ((ftth + hh) & 1)));
            rdA[hh] = (2 * fttd + pA) * V_RPLANE + r; rdB[hh] = (2 * fttd + pB) * V_RPLANE + r;
        }
        b_voff = fl * 16;
    };
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    lane_consts_compute();
    {
        i32x4v* const lc = reinterpret_cast<i32x4v*>(smem + 3 * V_RBUF + V_EX + V_SCR + V_RUN + V_KST) + 2 * tid;
        lc[0] = i32x4v{(int)col_rel, (int)col_bits, rdA[0], rdA[1]};
        lc[1] = i32x4v{rdB[0], rdB[1], b_voff, 0};
    }
    auto lane_consts = [&]() {      // (the lane's own slots, written by itself: LDS serves a wave's accesses in order)
        int fl;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(fl));
        const volatile i32x4v* const lc = reinterpret_cast<const volatile i32x4v*>(smem + 3 * V_RBUF + V_EX + V_SCR + V_RUN + V_KST) + 2 * (wave * 64 + fl);
        const i32x4v a0 = lc[0], a1 = lc[1];
        col_rel = (unsigned)a0[0]; col_bits = (unsigned)a0[1]; rdA[0] = a0[2]; rdA[1] = a0[3];
        rdB[0] = a1[0]; rdB[1] = a1[1]; b_voff = a1[2];
    };
    float m1 = -1.f, c2 = 2.f, c4 = 4.f, c8 = 8.f, cm4 = -4.f, cm5 = -5.f, cm2 = -2.f;
    asm volatile("" : "+s"(m1), "+s"(c2), "+s"(c4), "+s"(c8), "+s"(cm4), "+s"(cm5), "+s"(cm2));     // opaque constants: a + c*b becomes v_pk_fma_f32

    // ---- read plan of lane (tile tl, channel pair kk).  The D pass of B^T is done while reading: row pd = wave of the tile depth's 4
    // raw planes is  x[A] + sgn x[B]  with (A, B, sgn) = (0, 2, -), (1, 2, +), (2, 1, -), (1, 3, -).  Window rows h = 0, 1 have zh/2 = tth,
    // rows 2, 3 have tth + 1 (the XOR of the 16-byte half follows).
    const float dsg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(wave == 1 ? 0x3f800000 : (int)0xbf800000));
    auto rd_imm = [](int h, int w) { return (((h & 1) * 3) * V_ROW + (w & 3) * 5 + (w >> 2)) * 8; };

    // three stage buffers: `cur` = the unit being computed, `nx1` = the next unit (landed: its window is read under this unit's MFMAs), `nx2` = the
    // unit after that (its DMA is issued under this unit's MFMAs and has a whole unit to land)
    float* cur = smem;
    float* nx1 = smem + V_RBUF;
    float* nx2 = smem + 2 * V_RBUF;
    float* const ex = smem + 3 * V_RBUF;
    float* const scr = ex + V_EX;
    float* const run = scr + V_SCR + tid;
#ifdef E3_W4_TIMING
    const bool do_stats = false;
#else
    const bool do_stats = !AFF && !BNRED && a.stats != nullptr;
#endif
    if (do_stats || BNRED) {
#pragma unroll
        for (int k = 0; k < 9; ++k) run[k * 256] = 0.f;
    }
    // per-channel constants of the epilogue (bias, folded scale / shift) of the brick's 32 channels live in LDS: as buffer loads in the epilogue they
    // waited -- the memory counter retires in order -- for every request of the next units in front of them.  Reloaded only when a workgroup's
    // column tile changes (never when the grid tiles); two slots, because a wave may still be in the previous brick's epilogue.
    float* const kst = scr + V_SCR + V_RUN;
    int kslot = 0, k_n0 = -1;
    auto load_consts = [&](int n0) {
        if (n0 == k_n0) return;
        k_n0 = n0; kslot ^= 1;
        if (tid < 96) {
            const KArgs e = KA();
            const int which = tid >> 5, ch = n0 + (tid & 31);
            const float* const src = which == 0 ? e->bias : (which == 1 ? e->epi_scale : e->epi_shift);
            kst[kslot * 96 + tid] = (src != nullptr && ch < e->Ncols) ? src[ch] : (which == 1 ? 1.f : 0.f);
        }
    };
    if (HEAD) {          // the head's weights [class][32] and biases [4] in the running statistics' region (no statistics in this form; published by the prologue's barrier)
        float* const hw = scr + V_SCR + V_POOLX;
        const KArgs hk = KA();
        if (tid < 128) hw[tid] = tid < hk->head_cout * 32 ? hk->head_w[tid] : 0.f;
        else if (tid < 132) hw[tid] = (hk->head_b && tid - 128 < hk->head_cout) ? hk->head_b[tid - 128] : 0.f;
    }

    // ---- the workgroup's bricks: XCD x owns a contiguous eighth of the logical (XCD-blocked) brick range, its workgroups walk it with
    // stride gridDim / 8; a grid of nblk workgroups does one brick each
    unsigned L, Lend, Lstep;
    if (gridDim.x == nblk) { L = xcd_remap(blockIdx.x, nblk); Lend = L + 1; Lstep = 1; }
    else {
        const unsigned xcd = blockIdx.x & 7u, q = nblk >> 3, r = nblk & 7u;
        const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        L = base + (blockIdx.x >> 3); Lend = base + q + (xcd < r ? 1u : 0u); Lstep = gridDim.x >> 3;
    }
    auto range_mask = [](int lo, int n, int size) {      // bit z set: lo + z in [0, size), z in [0, n)
        const int first = lo < 0 ? -lo : 0, last = size - lo < n ? size - lo : n;
        return last > first ? ((1u << last) - 1u) & ~((1u << first) - 1u) : 0u;
    };
    // ---- brick cursors: mixed-radix digits (column tile, block-local w / h / d, block w / h / d, sample: brick_order.h) of a logical brick index, advanced by the workgroup's step with carries --
    // no division and no kernel-argument reload between bricks (the divisions of a per-brick decode cost ~1 k cycles outside the MFMA shadow).
    // P = the brick being computed, S = the brick of the staging cursor, which runs two units (8-channel chunks) ahead: S_c = its chunk.
    struct Cur { int nt, tw, th, td, nb; unsigned L; };
    const int tilesD = a.tilesD, tilesH = a.tilesH, tilesW = a.tilesW, ntiles = a.ntiles;
    auto advance = [&](Cur& c) {
        brick_advance(pa, true, ntiles, tilesW, tilesH, tilesD, c.nt, c.tw, c.th, c.td, c.nb);
        c.L += Lstep;
    };
    Cur P;
    {
        P.L = L;
        brick_decode(L, pa, ntiles, tilesW, tilesH, tilesD, P.nt, P.tw, P.th, P.td, P.nb);
    }
    Cur S = P;
    int S_c = 0;
    // halo origin (voxel (d0 - 1, h0 - 1, w0 - 1) of the brick, possibly in front of the tensor: only valid lanes form addresses from it) and validity
    // mask (6 d bits | 6 h bits | 18 w bits) of the staging cursor's brick; a brick beyond the end of the workgroup's range stages zeros
    const float* S_xorg; unsigned S_mask;
    auto stage_brick = [&]() {
        const int sd0 = S.td * 4 + a.org_d, sh0 = S.th * 4 + a.org_h, sw0 = S.tw * 16 + a.org_w;      // (org_*: voxel origin of the needed region's first brick)
        S_xorg = a.x + (((long long)S.nb * D + (sd0 - 1)) * ((long long)H * W * xl) + ((long long)(sh0 - 1) * W + (sw0 - 1)) * xl);
        const unsigned m_ = range_mask(sd0 - 1, 6, D) | (range_mask(sh0 - 1, 6, H) << 6) | (range_mask(sw0 - 1, 18, W) << 12);
        S_mask = S.L < Lend ? m_ : 0u;
    };
    auto stage_advance = [&]() { const bool wrap = S_c + 1 == NCH; S_c = wrap ? 0 : S_c + 1; if (wrap) { advance(S); stage_brick(); } };
    // the wave's transformed weights  U[ntile][chunk][pos 96][lane 64][ks 2][half 2]: wave = pd owns positions 24 pd .. 24 pd + 23
    auto wbase_of = [&](int nt) { return a.wt + ((size_t)nt * NCH * 96 + wave * 24) * 256; };
    // DMA piece `plane` (of this wave's quarter) of the staging cursor's unit into stage buffer `buf`: the plane's validity is the size of its
    // descriptor (scalar work only), the lane's validity the out-of-range offset
    auto issue_dma = [&](unsigned voff, float* buf, int plane) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S_xorg), 0, (((S_mask >> plane) & 1u) && !(E3_W4_ABL & 1)) ? 0x7fffffff : 0, 0x00020000);      // (ablation 1: every request out of range -- issued, answered with zeros, no traffic)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_v)(buf + plane * V_RPLANE + wave * 256), 16, voff, (int)(plane * plane_xb) + S_c * chunk_xb, 0, 0);
    };
    auto stage_voff = [&]() { return ((S_mask & col_bits) == col_bits) ? col_rel : OOB; };
    f32x4 acc[24][2];
    f32x4 Bv[24];
    f32x2v ra[4][6], rb[4][6];          // raw window of the NEXT unit (planes A and B), read under the MFMAs of the current one
    // (volatile LDS accesses: the compiler pairs plain 8-byte reads of one base into ds_read2_b64, which the LDS serves in 16-lane groups on 32 banks --
    // 8 LDS cycles per wave-instruction, and the two depth tiles of a group then meet on the same bank: 16 -- where two ds_read_b64 (32-lane groups on
    // 64 banks: the layout above is conflict-free for them) take 2 + 2.  SQ_LDS_BANK_CONFLICT 0.13 of the kernel's cycles before, profiles/r05_sq_counters.md.)
    auto read_window = [&](const float* buf, int h, int w) {
        ra[h][w] = *(lds_cv2)(buf + rdA[h >> 1] + rd_imm(h, w));
        rb[h][w] = *(lds_cv2)(buf + rdB[h >> 1] + rd_imm(h, w));
    };

#ifdef E3_W4_TIMING      // developer build (tools/phase_timing_w4.py): s_memtime stamps of the workgroup's third brick instead of statistics
    long long* const tstamp = reinterpret_cast<long long*>(KA()->stats) + (size_t)blockIdx.x * 48;
    int tbrick = 0;
#define TSTAMP(i) do { if (tid == 0 && tbrick == 2) tstamp[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i)
#endif
    load_consts(P.nt * 32);
    if (BNRED && tid < 128) {      // (one column tile per workgroup in this mode: the constants of its 32 channels, once; published by the prologue's barrier)
        const KArgs e = KA();
        const float* const src = (tid >> 5) == 0 ? e->br_scale : ((tid >> 5) == 1 ? e->br_shift : ((tid >> 5) == 2 ? e->br_mean : e->br_invstd));
        const int ch = P.nt * 32 + (tid & 31);
        (kst + 2 * 96)[tid] = ch < (e->br_cols ? e->br_cols : e->Ncols) ? src[ch] : 0.f;      // (br_cols: the unit in front owns only the first channels of y -- a concat gradient)
    }
    {   // prologue: units 0, 1 and 2 staged, the weights of unit 0 requested, the window of unit 0 read
        stage_brick();
        {
            const unsigned voff = stage_voff();
#pragma unroll
            for (int p = 0; p < 6; ++p) issue_dma(voff, cur, p);
        }
        stage_advance();
        {
            const unsigned voff = stage_voff();
#pragma unroll
            for (int p = 0; p < 6; ++p) issue_dma(voff, nx1, p);
        }
        stage_advance();
        {
            const unsigned voff = stage_voff();
#pragma unroll
            for (int p = 0; p < 6; ++p) issue_dma(voff, nx2, p);
        }
        stage_advance();
        const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase_of(P.nt)), 0, NCH * 96 * 1024, 0x00020000);
#pragma unroll
        for (int p = 0; p < 24; ++p) Bv[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_voff + (p & 3) * 1024, (p & ~3) * 1024, 0));
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(36)" ::: "memory");       // (unit 0 has landed: its 6 pieces are the oldest requests)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int w = 0; w < 6; ++w) read_window(cur, h, w);
    }

    for (;;) {
        const bool has_next = P.L + Lstep < Lend;
        Cur N = P;
        advance(N);
        const float* const P_wbase = wbase_of(P.nt);
        const float* const N_wbase = wbase_of(N.nt);

        // One 8-channel chunk c (unit u) of brick P.  Its raw window is in the registers ra / rb (read under the MFMAs of the previous unit, or
        // in the previous brick's epilogue).  Program order: D, H, W passes (VALU phase: fp32 VALU and fp32 MFMA share the FMA lanes, so
        // nothing else is put here); wait for this wave's DMA pieces of unit u + 1 (issued during unit u - 1: the 14 weight requests behind
        // the last piece may still be outstanding), barrier (publishes every wave's pieces of unit u + 1 and retires the buffer of unit u - 1);
        // then 12 position pairs of 8 MFMAs, each carrying two window reads (ds_read2_b64) of unit u + 1 and the weight requests of the same
        // positions of unit u + 1 (a ring of 24: one full unit of look-ahead); the first six pairs also carry one DMA piece of unit u + 2.
        // (Measured and dropped, round 6: a third form of the chunk for a brick's LAST chunk without its window reads -- unit u + 1 is then the next brick's
        // first unit, whose window the epilogue reads again anyway: bit-identical, step and cfg-5 tile +-0 on the same box, 6 KB more code.)
        auto chunk = [&](auto zero_tag, int c) {
            constexpr bool ZERO = decltype(zero_tag)::value;
            const bool lastc = c + 1 == NCH;
            f32x2v t[4][6];
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int w = 0; w < 6; ++w) t[h][w] = ra[h][w] + dsg * rb[h][w];
            // H pass: F(2,3) rows  t0 - t2,  t1 + t2,  t2 - t1,  t1 - t3
#pragma unroll
            for (int w = 0; w < 6 && !(E3_W4_ABL & 8); ++w) {
                const f32x2v u0 = t[0][w] + m1 * t[2][w], u1 = t[1][w] + t[2][w], u2 = t[2][w] + m1 * t[1][w], u3 = t[1][w] + m1 * t[3][w];
                t[0][w] = u0; t[1][w] = u1; t[2][w] = u2; t[3][w] = u3;
            }
            // W pass: F(4,3) rows  4 d0 - 5 d2 + d4,  -4 d1 - 4 d2 + d3 + d4,  4 d1 - 4 d2 - d3 + d4,  -2 d1 - d2 + 2 d3 + d4,  2 d1 - d2 - 2 d3 + d4,  4 d1 - 5 d3 + d5
#pragma unroll
            for (int h = 0; h < 4 && !(E3_W4_ABL & 8); ++h) {
                const f32x2v d0 = t[h][0], d1 = t[h][1], d2 = t[h][2], d3 = t[h][3], d4 = t[h][4], d5 = t[h][5];
                const f32x2v e0 = d4 + cm4 * d2, e1 = d3 + cm4 * d1, f0 = d4 + m1 * d2, f1 = d3 + m1 * d1;
                t[h][0] = (d4 + cm5 * d2) + c4 * d0;
                t[h][1] = e0 + e1;
                t[h][2] = e0 + m1 * e1;
                t[h][3] = f0 + c2 * f1;
                t[h][4] = f0 + cm2 * f1;
                t[h][5] = (d5 + cm5 * d3) + c4 * d1;
            }
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int w = 0; w < 6; ++w) asm volatile("" : "+v"(t[h][w]));     // keep the transform packed and in front of the MFMA block
            __builtin_amdgcn_sched_barrier(0);
            if (c < 8) TSTAMP(1 + 5 * c);
            // (unit u + 1's pieces went out in pairs 6 - 11 of unit u - 2; behind the last of them: that pair's 2 weight requests, then unit u - 1's 24 + 6 -- or,
            // after the prologue, unit 2's 6 pieces + 24 weight requests: 30 covers both.  In the steady state the wait never stalls: weight requests younger than
            // the pieces were consumed a unit ago, and the memory counter retires in order.)
            if (E3_W4_ABL & 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
            if (c < 8) TSTAMP(2 + 5 * c);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (c < 8) TSTAMP(3 + 5 * c);
            const unsigned d_voff = stage_voff();
            const float* const wb = lastc ? N_wbase : P_wbase;
            const int cB = lastc ? 0 : c + 1;
            // (the chunk's share of the weights starts at the descriptor's base: the scalar offsets below are compile-time constants)
            const __amdgpu_buffer_rsrc_t b_nx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wb) + (size_t)cB * 96 * 256, 0, (lastc && !has_next) ? 0 : 96 * 1024, 0x00020000);
#pragma unroll
            for (int pp = 0; pp < 24; pp += 2) {        // two positions at a time: 4 independent accumulators in flight, then their ring slots are refilled
                if (pp >= 12) issue_dma(d_voff, cur, (pp - 12) >> 1);      // unit u + 3 into the buffer of unit u (its window went to registers a unit ago)
                if (!(E3_W4_ABL & 64)) {                // window elements (h, w) and (h, w') of unit u + 1: w' = w + 4 (w = 0, 1) resp. 3 (w = 2) -- 32 / 160 bytes apart: one ds_read2_b64 per plane
                    const int k = pp >> 1, h = k / 3, wa = k % 3, wb2 = wa == 2 ? 3 : wa + 4;
                    read_window(nx1, h, wa); read_window(nx1, h, wb2);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int p = pp; p < pp + 2; ++p)
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            if (E3_W4_ABL & 32) continue;
                            const float wv = Bv[p][ks * 2 + hf], tv = t[p / 6][p % 6][ks];
                            if (ZERO && ks == 0) {
                                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                                acc[p][hf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, tv, z, 0, 0, 0);
                            } else
                                acc[p][hf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, tv, acc[p][hf], 0, 0, 0);
                        }
#pragma unroll
                for (int p = pp; p < pp + 2; ++p) {
                    if (E3_W4_ABL & 2) continue;
                    Bv[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_nx, b_voff + (p & 3) * 1024, (p & ~3) * 1024, 0));
                }
                // (the window reads are volatile and stay behind the DMA piece in front of them)
// issue order inside the pair: MFMA, [DMA piece], two window reads, 2 MFMAs, two window reads, 3 MFMAs, weight request, 2 MFMAs, weight request
                // (a position's weights are free after its last MFMA: the sixth resp. eighth of the pair).  Measured alternatives, same box, cfg-5 tile: the four
                // reads together 6.69 ms, both weight requests at the end 6.60, this order 6.58, one read per MFMA 6.63, DMA piece late 6.64, position-major
                // MFMAs with the first request after four 6.65 (profiles/r05_w4_phases.md section 9)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (pp >= 12) { __builtin_amdgcn_sched_group_barrier(0x004, 8, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (pp == 22) stage_advance();      // (the cursor's scalar arithmetic rides in the matrix shadow; the last piece of this unit has just gone out)
                if (pp == 10 && c < 8) TSTAMP(4 + 5 * c);
            }
            if (c < 8) TSTAMP(5 + 5 * c);
            { float* tsw = cur; cur = nx1; nx1 = nx2; nx2 = tsw; }
        };

        TSTAMP(0);
        chunk(std::true_type{}, 0);
        for (int c = 1; c < NCH; ++c) chunk(std::false_type{}, c);

        // ---- epilogue.  acc[ph*6+pw][half][i]: position (pd = wave, ph, pw), tile tl, channel n0 + 8 kk + 4 half + i.
        // per-channel constants first: they arrive during the output transform
        // (the lane index is taken afresh: lane constants of the epilogue kept across the main loop were spilled, and a scratch reload in here waits -- the
        // memory counter retires in order -- for every request of the next units in flight)
        int elane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
        const int etl = elane & 15, ekk = elane >> 4;
        const int ettd = etl >> 3, etth = (etl >> 2) & 1, ettw = etl & 3;
        const int etid = wave * 64 + elane;
        float* const erun = scr + V_SCR + etid;
        const int n0 = P.nt * 32, d0 = P.td * 4 + KA()->org_d, h0 = P.th * 4 + KA()->org_h, w0 = P.tw * 16 + KA()->org_w;
        const int P_nb = P.nb;
        // BNRED: the raw tensor of the unit in front at the eight voxel rows this lane will STORE (rows v = 8 j + r, piece (lane & 7) ^ r;
        // same voxels as the output, its own channel stride) is requested here, in front of the output transform: the rows come from HBM, and requested next to
        // the stores they cost a memory round trip per brick (+18 % on the launch)
        f32x4 bx[8];
        // (a concat gradient, br_cols < Ncols: the workgroups of the other column tiles ask for nothing -- every lane out of range -- and their sums are never written)
        if (BNRED) {
            const KArgs e = KA();
            const int bl = e->br_ldc;
            const int oh_ = wave >> 1, owb_ = 2 * (wave & 1);
            const int r_ = elane >> 3, pc_ = (elane & 7) ^ r_;      // (the lane's rows and piece of the store phase below)
            const int sgh_ = h0 + 2 * (r_ >> 2) + oh_, sgw_ = w0 + 4 * (r_ & 3) + owb_;
            const size_t plane_b = (size_t)H * W * bl;
            const size_t brem = (size_t)(D - d0) * plane_b * 4;
            const __amdgpu_buffer_rsrc_t x2_rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(e->br_x) + ((size_t)P_nb * D + d0) * plane_b, 0, (int)(brem < 0x7fffffffu ? brem : 0x7fffffffu), 0x00020000);
            const bool cok_ = n0 + 4 * pc_ < (e->br_cols ? e->br_cols : e->Ncols) && sgh_ < H;
            const unsigned b_off = (unsigned)(((sgh_ * W + sgw_) * bl + n0 + 4 * pc_) * 4);
            const unsigned bv[2] = {(cok_ && sgw_ < W) ? b_off : OOB, (cok_ && sgw_ + 1 < W) ? b_off : OOB};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int od = j >> 2, owi = (j >> 1) & 1, std_ = 2 * (j & 1) + od;
                bx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x2_rs, (d0 + std_ < D) ? bv[owi] : OOB, (int)((std_ * plane_b + owi * bl) * 4), 0));
            }
        }
        // A^T m A over (pw, ph) in registers, one channel half at a time, one ph row at a time (`ex` is its own LDS region: the next brick's first
        // chunk is already in the stage buffers).  W: F(4,3) rows  m0+m1+m2+m3+m4,  m1-m2+2m3-2m4,  m1+m2+4m3+4m4,  m1-m2+8m3-8m4+m5
        // (Measured and dropped, round 6: the H pass first -- 44 instead of 56 vector ops per channel half, 48 packed instructions fewer per brick -- changes the
        // rounding and moves nothing: cfg-5 tile 6.326 -> 6.317 ms, step 10.952 -> 10.979 ms on one box.)
#pragma unroll
        for (int hf = 0; hf < 2 && !(E3_W4_ABL & 128); ++hf) {
            f32x4 q[2][4];
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const f32x4 a0 = acc[ph * 6 + 0][hf], a1 = acc[ph * 6 + 1][hf], a2 = acc[ph * 6 + 2][hf], a3 = acc[ph * 6 + 3][hf], a4 = acc[ph * 6 + 4][hf], a5 = acc[ph * 6 + 5][hf];
                const f32x4 s12 = a1 + a2, d12 = a1 + m1 * a2, s34 = a3 + a4, d34 = a3 + m1 * a4;
                f32x4 o[4];
                o[0] = (a0 + s12) + s34;
                o[1] = d12 + c2 * d34;
                o[2] = s12 + c4 * s34;
                o[3] = (d12 + c8 * d34) + a5;
#pragma unroll
                for (int ow = 0; ow < 4; ++ow) {
                    if (ph == 0) q[0][ow] = o[ow];
                    else if (ph == 1) { q[0][ow] += o[ow]; q[1][ow] = o[ow]; }
                    else if (ph == 2) { q[0][ow] += o[ow]; q[1][ow] += m1 * o[ow]; }
                    else q[1][ow] += m1 * o[ow];
                }
            }
#pragma unroll
            for (int oh = 0; oh < 2; ++oh)
#pragma unroll
                for (int ow = 0; ow < 4; ++ow)
                {
                    if constexpr (HEAD) *reinterpret_cast<f32x4*>(ex + ((wave * 16 + (oh * 4 + ow) * 2 + hf) * 64 + elane) * 4) = q[oh][ow];
                    else *reinterpret_cast<f32x4*>(ex + (((wave * 8 + oh * 4 + ow) * 16 + etl) * 8 + ((2 * ekk + hf) ^ (etl & 7))) * 4) = q[oh][ow];
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        TSTAMP(41);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        TSTAMP(42);
        // wave w now owns output offsets oh = w >> 1, ow = 2 (w & 1) + {0, 1} of every tile and sums the pd axis: od = 0, 1
        const int oh = wave >> 1, owb = 2 * (wave & 1);
        if constexpr (HEAD) {
            // (the fused head works on the accumulator layout -- lane = (tile, channel octet) -- and stores no rows: the exchange keeps the lane-major layout)
            f32x4 y[2][2][2];       // [od][owi][half]
            {
                f32x4 m[2][2][4], bias[2], es[2], eh[2];
    #pragma unroll
                for (int owi = 0; owi < 2; ++owi)
    #pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
    #pragma unroll
                        for (int pd = 0; pd < 4; ++pd) m[owi][hf][pd] = *reinterpret_cast<const f32x4*>(ex + ((pd * 16 + (oh * 4 + owb + owi) * 2 + hf) * 64 + elane) * 4);
    #pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    bias[hf] = *reinterpret_cast<const f32x4*>(kst + kslot * 96 + 8 * ekk + 4 * hf);
                    if (AFF) {
                        es[hf] = *reinterpret_cast<const f32x4*>(kst + kslot * 96 + 32 + 8 * ekk + 4 * hf);
                        eh[hf] = *reinterpret_cast<const f32x4*>(kst + kslot * 96 + 64 + 8 * ekk + 4 * hf);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int owi = 0; owi < 2; ++owi)
    #pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        y[0][owi][hf] = m[owi][hf][0] + m[owi][hf][1] + m[owi][hf][2] + bias[hf];
                        y[1][owi][hf] = m[owi][hf][1] + m1 * m[owi][hf][2] + m1 * m[owi][hf][3] + bias[hf];
                        if (AFF) {
    #pragma unroll
                            for (int od = 0; od < 2; ++od)
    #pragma unroll
                                for (int e = 0; e < 4; ++e) y[od][owi][hf][e] = fmaxf(__builtin_fmaf(y[od][owi][hf][e], es[hf][e], eh[hf][e]), 0.f);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the accumulators and the exchanged sums are dead: the window of the next brick's first unit (in `cur` after the rotation; landed and published by
            // the last chunk's barrier) is read under the stores and the statistics
            // (BNRED: behind the reduction instead -- its raw rows and sums need the registers)
            // (the lane constants of the chunk loop are taken afresh right HERE, in front of their first use: the window addresses that survived the output transform
            // were spilled in the BNRED form, and their scratch reload waited -- the memory counter retires in order -- for the staging requests of two units)
            if (!BNRED) lane_consts();
            if (!(E3_W4_ABL & 64) && !BNRED) {
    #pragma unroll
                for (int h = 0; h < 4; ++h)
    #pragma unroll
                    for (int w = 0; w < 6; ++w) read_window(cur, h, w);
            }
            // voxel (d0 + 2 ettd + od, h0 + 2 etth + oh, w0 + 4 ettw + owb + owi), channels nq + 4 half .. + 3: 16-byte stores
            const int gh = h0 + 2 * etth + oh, gw = w0 + 4 * ettw + owb, gd = d0 + 2 * ettd;

            if (HEAD) {
                // conv_final_fwd_kernel's arithmetic on the registers: channel quad q of the voxel is summed as an fmaf chain from 0, the eight quad sums meet as
                // ((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7)) -- this elane holds quads 2 ekk (half 0) and 2 ekk + 1 (half 1), the lanes ^ 16, ^ 32 the others
                const float* const hw = scr + V_SCR + V_POOLX;
                const KArgs e = KA();
                const int hc = e->head_cout;
                float lg[2][2][4];
    #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 wv0 = *reinterpret_cast<const f32x4*>(hw + c * 32 + 8 * ekk), wv1 = *reinterpret_cast<const f32x4*>(hw + c * 32 + 8 * ekk + 4);
                    const float hb = hw[128 + c];
    #pragma unroll
                    for (int od = 0; od < 2; ++od)
    #pragma unroll
                        for (int owi = 0; owi < 2; ++owi) {
                            float s0 = 0.f, s1 = 0.f;
    #pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) { s0 = __builtin_fmaf(y[od][owi][0][e4], wv0[e4], s0); s1 = __builtin_fmaf(y[od][owi][1][e4], wv1[e4], s1); }
                            float p = s0 + s1;                  // q(2 ekk) + q(2 ekk + 1)
                            p = p + __shfl_xor(p, 16);          // (q0 + q1) + (q2 + q3)   resp.   (q4 + q5) + (q6 + q7)
                            p = p + __shfl_xor(p, 32);
                            lg[od][owi][c] = p + hb;
                        }
                }
                // elane ekk finishes voxel (od, owi) = (ekk >> 1, ekk & 1)
                float l4[4];
    #pragma unroll
                for (int c = 0; c < 4; ++c) l4[c] = ekk == 0 ? lg[0][0][c] : (ekk == 1 ? lg[0][1][c] : (ekk == 2 ? lg[1][0][c] : lg[1][1][c]));
                if (e->head_softmax) {
                    float m = l4[0];
    #pragma unroll
                    for (int c = 1; c < 4; ++c) m = c < hc ? fmaxf(m, l4[c]) : m;
                    float sm = 0.f;
    #pragma unroll
                    for (int c = 0; c < 4; ++c) { l4[c] = c < hc ? __expf(l4[c] - m) : 0.f; sm += l4[c]; }
                    const float inv = 1.f / sm;
    #pragma unroll
                    for (int c = 0; c < 4; ++c) l4[c] *= inv;
                }
                const int vd = gd + (ekk >> 1), vw = gw + (ekk & 1);
                const bool inb = gh < H && vw < W && vd < D && vd >= e->head_lo[0] && vd < e->head_hi[0] && gh >= e->head_lo[1] && gh < e->head_hi[1] && vw >= e->head_lo[2] && vw < e->head_hi[2];
                if (inb) {
                    float* const yo = e->head_y + (long long)P_nb * e->head_ys[0] + (long long)(vd - e->head_lo[0]) * e->head_ys[2] + (long long)(gh - e->head_lo[1]) * e->head_ys[3] + (vw - e->head_lo[2]);
    #pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < hc) yo[(long long)c * e->head_ys[1]] = l4[c];
                }
            }
        } else {
            // Round 6: the exchange is written tile-major ([pd][(oh, ow)][tile][8 pieces of 4 channels], piece 2 kk + half at slot piece ^ (tile & 7):
            // conflict-free writes: a ds_write_b128 is served in groups of 8 consecutive lanes = 8 tiles on 32 banks), so the lane that sums the pd axis can pick the pieces in STORE order: lane (r = lane >> 3, pos = lane & 7) takes the
            // piece pc = pos ^ r of the tiles 8 th + r, th = 0, 1 (the slot is pos; 8 consecutive lanes read one tile's 128 contiguous bytes, the 16-lane
            // groups of a ds_read_b128 cover all 64 banks) and holds, summed, the rows  j = 4 od + 2 owi + th  of 8 lanes x 16 bytes = one whole 128-byte voxel row per 8 lanes -- what round 5
            // obtained with a second pass through LDS (8 ds_write_b128 + 8 ds_read_b128 per lane and brick: 64 KB of the epilogue's ~290 KB of LDS
            // traffic, and one more dependent LDS round trip).  Same sums in the same order: bit-identical results.
            const int r = elane >> 3, pos = elane & 7, pc = pos ^ r;
            f32x4 yv[8];
            {
                f32x4 m[2][2][4], bias, es, eh;
#pragma unroll
                for (int owi = 0; owi < 2; ++owi)
#pragma unroll
                    for (int th = 0; th < 2; ++th)
#pragma unroll
                        for (int pd = 0; pd < 4; ++pd)
                            m[owi][th][pd] = *reinterpret_cast<const f32x4*>(ex + (((pd * 8 + oh * 4 + owb + owi) * 16 + 8 * th + r) * 8 + pos) * 4);
                bias = *reinterpret_cast<const f32x4*>(kst + kslot * 96 + 4 * pc);
                if (AFF) {
                    es = *reinterpret_cast<const f32x4*>(kst + kslot * 96 + 32 + 4 * pc);
                    eh = *reinterpret_cast<const f32x4*>(kst + kslot * 96 + 64 + 4 * pc);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int owi = 0; owi < 2; ++owi)
#pragma unroll
                    for (int th = 0; th < 2; ++th) {
                        yv[owi * 2 + th] = m[owi][th][0] + m[owi][th][1] + m[owi][th][2] + bias;
                        yv[4 + owi * 2 + th] = m[owi][th][1] + m1 * m[owi][th][2] + m1 * m[owi][th][3] + bias;
                        if (AFF) {
#pragma unroll
                            for (int od = 0; od < 2; ++od)
#pragma unroll
                                for (int e = 0; e < 4; ++e) yv[od * 4 + owi * 2 + th][e] = fmaxf(__builtin_fmaf(yv[od * 4 + owi * 2 + th][e], es[e], eh[e]), 0.f);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the accumulators and the exchanged sums are dead: the window of the next brick's first unit (in `cur` after the rotation; landed and published by
            // the last chunk's barrier) is read under the stores and the statistics (BNRED: behind the reduction instead -- its raw rows and sums need the registers).
            // The lane constants of the chunk loop are taken afresh right HERE, in front of their first use: the window addresses that survived the output
            // transform were spilled in the BNRED form, and their scratch reload waited -- the memory counter retires in order -- for the staging requests in flight.
            if (!BNRED) lane_consts();
            if (!(E3_W4_ABL & 64) && !BNRED) {
#pragma unroll
                for (int h = 0; h < 4; ++h)
#pragma unroll
                    for (int w = 0; w < 6; ++w) read_window(cur, h, w);
            }
            // lane (r, pos) stores piece pc of the rows v = 8 j + r: tile 8 (j & 1) + r = (td j & 1, th r >> 2, tw r & 3), od = j >> 2, owi = (j >> 1) & 1:
            // voxel (d0 + 2 (j & 1) + od, h0 + 2 (r >> 2) + oh, w0 + 4 (r & 3) + owb + owi), channels n0 + 4 pc .. + 3
            const int sgh = h0 + 2 * (r >> 2) + oh, sgw = w0 + 4 * (r & 3) + owb;
            const int yl = KA()->y_ldc;
            const size_t plane_y = (size_t)H * W * yl;
            // channel-chunked output (ConvArgs::y_chunk, plain stores only): a voxel's row is the 8 channels of ONE chunk plane (ys = 8), the chunk planes
            // are y_chunk floats apart -- the lane's piece goes to plane (n0 + 4 pc) / 8.  (The launcher bounds the whole tensor by 2^31 bytes.)
            const size_t ychk = BNRED ? (size_t)0 : KA()->y_chunk;
            const int ys = ychk ? 8 : yl;
            const bool nt_out = !AFF && !ychk && (size_t)KA()->N * D * H * W * KA()->Ncols * 4 > (size_t)E3_W4_NT_MIN_MB * 1048576;
            const size_t plane_s = ychk ? (size_t)H * W * 8 : plane_y;
            const size_t yrem = ychk ? (size_t)0x7fffffffu : (size_t)(D - d0) * plane_y * 4;
            const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(
                KA()->y + ((size_t)P_nb * D + d0) * plane_s, 0, (int)(yrem < 0x7fffffffu ? yrem : 0x7fffffffu), 0x00020000);
            // (store box, ConvArgs::sbox_*: the launcher sets [0, dims) when it is off)
            const int sb_d0 = KA()->sbox_lo[0], sb_d1 = KA()->sbox_hi[0], sb_h0 = KA()->sbox_lo[1], sb_h1 = KA()->sbox_hi[1], sb_w0 = KA()->sbox_lo[2], sb_w1 = KA()->sbox_hi[2];
            const bool cok = n0 + 4 * pc < KA()->Ncols && sgh < sb_h1 && sgh >= sb_h0;
            const unsigned s_voff = ychk ? (unsigned)(((size_t)((n0 + 4 * pc) >> 3) * ychk + (size_t)((sgh * W + sgw) * 8 + 4 * (pc & 1))) * 4)
                                         : (unsigned)(((sgh * W + sgw) * yl + n0 + 4 * pc) * 4);
            const unsigned sv[2] = {(cok && sgw < sb_w1 && sgw >= sb_w0) ? s_voff : OOB, (cok && sgw + 1 < sb_w1 && sgw + 1 >= sb_w0) ? s_voff : OOB};
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (E3_W4_ABL & 4) continue;
                const int od = j >> 2, owi = (j >> 1) & 1, std_ = 2 * (j & 1) + od;
                const f32x4 v = yv[j];
                const bool dok = d0 + std_ < sb_d1 && d0 + std_ >= sb_d0;
                // (training forms, rows: the gradient / raw tensor of a large grid is streamed out past the caches -- non-temporal, aux = 2 -- so that the halo lines and the
                // weights the neighbouring bricks re-read stay resident: step -0.08 ms; the 32-byte fragments of the channel-chunked inference tensors must NOT go that way: tile +6.6 %)
                if (nt_out) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), y_rs, dok ? sv[owi] : OOB, (int)((std_ * plane_s + owi * ys) * 4), 2);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), y_rs, dok ? sv[owi] : OOB, (int)((std_ * plane_s + owi * ys) * 4), 0);
                if (BNRED) {
                    // dz = dA * act'(z), z = x * scale + shift;  xhat = (x - mean) * invstd  (the expressions of bn_bwd_kernel)
                    const float* const kb = kst + 2 * 96 + 4 * pc;
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(kb), sh = *reinterpret_cast<const f32x4*>(kb + 32);
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(kb + 64), is = *reinterpret_cast<const f32x4*>(kb + 96);
                    const bool vok = dok && sv[owi] != OOB;
                    const float slope = KA()->br_slope;
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float xv = bx[j][e4];
                        const float z = __builtin_fmaf(xv, sc[e4], sh[e4]);
                        const float gv = vok ? v[e4] : 0.f, gs = slope * gv + 0.f;
                        const float dz = z > 0.f ? gv : gs;      // (act_bwd of a constant-slope activation: the caller's condition; no branch)
                        const float xh = (xv - mu[e4]) * is[e4];
                        s1[e4] += dz; s2[e4] = __builtin_fmaf(dz, xh, s2[e4]);
                    }
                }
            }
            if (BNRED) {
                // running sums of this lane's 4 channels in LDS (no registers across the chunk loop); one partial row per workgroup at the end
                f32x4 r1, r2;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) { r1[e4] = erun[e4 * 256] + s1[e4]; r2[e4] = erun[(4 + e4) * 256] + s2[e4]; }
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) { erun[e4 * 256] = r1[e4]; erun[(4 + e4) * 256] = r2[e4]; }
                if (!has_next) {       // (uniform) channel c = 4 pc + e: its 8 lanes per wave are l = 8 r' + (pc ^ r'), the 4 waves in order: fixed summation order
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    const KArgs e = KA();
                    const int eN = e->br_cols ? e->br_cols : e->Ncols;
                    if (etid < 32 && n0 + etid < eN) {
                        const int cpc = etid >> 2, ce = etid & 3;
                        float t1 = 0.f, t2 = 0.f;
                        for (int w = 0; w < 4; ++w)
                            for (int rr = 0; rr < 8; ++rr) {
                                const float* const src = scr + V_SCR + w * 64 + 8 * rr + (cpc ^ rr);
                                t1 += src[ce * 256]; t2 += src[(4 + ce) * 256];
                            }
                        const unsigned nt_ = (unsigned)e->ntiles;
                        const size_t row = (size_t)((blockIdx.x & 7u) * ((gridDim.x >> 3) / nt_) + (blockIdx.x >> 3) / nt_);
                        float* o = e->br_part + row * 3 * (size_t)eN + n0 + etid;
                        o[0] = t1; o[eN] = t2;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                lane_consts();
                if (!(E3_W4_ABL & 64)) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int h = 0; h < 4; ++h)
#pragma unroll
                        for (int w = 0; w < 6; ++w) read_window(cur, h, w);
                }
            }
            TSTAMP(43);
            // validity of the lane's eight rows (POOL, statistics)
            const bool okh = sgh < H;
            const bool okw[2] = {okh && sgw < W, okh && sgw + 1 < W};
            if (POOL) {
                // a 2x2x4 tile = two pooling windows (ow 0, 1 | ow 2, 3): max over od and the lane's two ow in registers (per tile th, the lane's 4 channels), over
                // oh = the wave pairs (w, w ^ 2) through LDS (voxels outside the tensor do not take part: ceil_mode; NaN propagates as in nn.MaxPool3d); wave w
                // then stores the tiles th = w >> 1 of window w & 1
                float* const px = scr + V_SCR;
                auto nmax = [](float a_, float b_) { return (b_ > a_ || b_ != b_) ? b_ : a_; };
                constexpr float NEG = -3.4028235e38f;
#pragma unroll
                for (int th = 0; th < 2; ++th) {
                    f32x4 pm;
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        float v = NEG;
#pragma unroll
                        for (int od = 0; od < 2; ++od)
#pragma unroll
                            for (int owi = 0; owi < 2; ++owi) v = nmax(v, (d0 + 2 * th + od < D && okw[owi]) ? yv[od * 4 + owi * 2 + th][e4] : NEG);
                        pm[e4] = v;
                    }
                    *reinterpret_cast<f32x4*>(px + ((wave * 64 + elane) * 8) + 4 * th) = pm;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // wave w: window (w & 1), tiles th = (w >> 1): max of the partials of waves (w & 1) and (w & 1) + 2
                const int win = wave & 1, tsel = wave >> 1;
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(px + ((win * 64 + elane) * 8) + 4 * tsel);
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(px + (((win + 2) * 64 + elane) * 8) + 4 * tsel);
                f32x4 best;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) best[e4] = nmax(p0[e4], p1[e4]);
                const KArgs e = KA();
                const int eN = e->Ncols;
                const int Dp = (D + 1) >> 1, Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
                const int pd_ = (d0 >> 1) + tsel, ph_ = (h0 >> 1) + (r >> 2), pw_ = (w0 >> 1) + 2 * (r & 3) + win;
                const int pch = n0 + 4 * pc;
                const bool pok = pd_ < Dp && ph_ < Hp && pw_ < Wp && pch < eN;
                const size_t pv = (((size_t)P_nb * Dp + pd_) * Hp + ph_) * Wp + pw_;
                const size_t pck = e->pool_chunk;      // (channel-chunked pooled tensor: plane pch / 8, ConvArgs::pool_chunk)
                if (pok) *reinterpret_cast<f32x4*>(e->pool_out + (pck ? (size_t)(pch >> 3) * pck + pv * 8 + (pch & 4) : pv * eN + pch)) = best;
            }
            if (do_stats) {
                // running record (count, mean[4], M2[4]) of this lane's 4 channels: Chan merge of the brick's (up to) eight values per channel
                bool ok[8];
                float cb = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { ok[j] = d0 + 2 * (j & 1) + (j >> 2) < D && okw[(j >> 1) & 1]; cb += ok[j] ? 1.f : 0.f; }
                const float rn = erun[0];
                const float nn = rn + cb;
                const float rf = cb * __builtin_amdgcn_rcpf(fmaxf(nn, 1.f));      // (approximate reciprocal: its error is far below the rounding of the sums)
                const float rc = cb > 0.f ? 1.f / cb : 0.f;
                const float rnf = rn * rf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) s += ok[j] ? yv[j][e] : 0.f;
                    const float bm = s * rc;
                    float b2 = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float dv = yv[j][e] - bm; b2 += ok[j] ? dv * dv : 0.f; }
                    const float rmean = erun[(1 + e) * 256], rm2 = erun[(5 + e) * 256];
                    const float dl = bm - rmean;
                    erun[(1 + e) * 256] = rmean + dl * rf;
                    erun[(5 + e) * 256] = rm2 + b2 + dl * dl * rnf;
                }
                erun[0] = nn;
                if (!wgstats || !has_next) {       // (uniform) one record per channel: its 8 lanes per wave are l = 8 r' + (pc ^ r'), the 4 waves in order
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (etid < 32 && n0 + etid < KA()->Ncols) {
                        const int cpc = etid >> 2, ce = etid & 3;
                        float c0 = 0.f, me = 0.f, mm = 0.f;
                        for (int w = 0; w < 4; ++w)
                            for (int rr = 0; rr < 8; ++rr) {
                                const float* const src = scr + V_SCR + w * 64 + 8 * rr + (cpc ^ rr);
                                welford_merge(c0, me, mm, src[0], src[(1 + ce) * 256], src[(5 + ce) * 256]);
                            }
                        const size_t row = wgstats ? (size_t)((blockIdx.x & 7u) * ((gridDim.x >> 3) / (unsigned)KA()->ntiles) + (blockIdx.x >> 3) / (unsigned)KA()->ntiles)
                                                   : (size_t)(((P.nb * tilesD + P.td) * tilesH + P.th) * tilesW + P.tw);
                        float* o = KA()->stats + (row * KA()->Cout + n0 + etid) * 3;
                        o[0] = c0; o[1] = me; o[2] = mm;
                    }
                    if (!wgstats) {      // one record per brick: the lanes start afresh -- behind the merge that read their records
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < 9; ++k) erun[k * 256] = 0.f;
                    }
                }
            }
        }
        TSTAMP(44);
#ifdef E3_W4_TIMING
        ++tbrick;
#endif
        if (!has_next) break;
        P = N;
        load_consts(P.nt * 32);
    }
}

}  // namespace

// ---- host side
int wino4_bricks(int N, int D, int H, int W) { return N * cdiv(D, 4) * cdiv(H, 4) * cdiv(W, 16); }

// one column tile per workgroup and every column tile covered by the workgroups of a row: XCD ranges that start at multiples of ntiles and a
// stride of 32 logical bricks that is one too
static bool wino4_wgstats(size_t nblk, int ntiles, unsigned grid) {
    return grid == 256u && nblk > 256 && nblk % 8 == 0 && (nblk / 8) % (size_t)ntiles == 0 && 32 % ntiles == 0;
}

int conv_wino4_bnred_parts(int N, int D, int H, int W, int K, int ncols) {
    if ((ncols & 3) || (K & 7)) return 0;
    const int ntiles = (ncols + 31) / 32;
    const size_t nblk = (size_t)wino4_bricks(N, D, H, W) * ntiles;
    return wino4_wgstats(nblk, ntiles, nblk >= 256 ? 256u : (unsigned)nblk) ? 256 / ntiles : 0;
}

int wino4_stats_parts(int N, int D, int H, int W, int ncols) {
    const int bricks = wino4_bricks(N, D, H, W), ntiles = (ncols + 31) / 32;
    const size_t nblk = (size_t)bricks * ntiles;
    return wino4_wgstats(nblk, ntiles, nblk >= 256 ? 256u : (unsigned)nblk) ? 256 / ntiles : bricks;
}

// Which decomposition a Winograd 3x3x3 launch takes: 0 = F(2x2x2) tiles (conv_wino.hip), 2 = F(2x2x4) tiles (this file).  THE predicate: the weight
// packers (different layouts), the statistics sizing and the launcher all ask it.  Decided on the grid of ONE sample (like conv_use_wino) so that the
// arithmetic does not depend on the batch size.  The caller allows the larger tiles per launch with CF_WINO4 (eval-mode forwards, data gradients: see the
// head of this file); a split-K launch, the timing flag and views that rule out 16-byte accesses keep F(2x2x2).
int conv_wino_layout(int flags, int D, int H, int W, int K, int ncols, int splitk) {
    static const int mode = getenv("E3_WINO4") ? atoi(getenv("E3_WINO4")) : 1;      // 0: never (A/B switch), 1: where the caller allows it, 2: every eligible launch (tests)
    static const size_t minblk = getenv("E3_WINO4_MIN") ? (size_t)atol(getenv("E3_WINO4_MIN")) : 256;      // (measured 512 / 256 / 128: 11.23 / 11.21 / 11.21 ms per step, Predictor tile 7.03 / 6.98 / 6.98 ms)
    if (splitk > 1 || (flags & 1024) || (ncols & 3) || (K & 7)) return 0;
    if (mode <= 0 || (mode == 1 && !(flags & CF_WINO4))) return 0;
    const size_t nblk1 = (size_t)wino4_bricks(1, D, H, W) * ((ncols + 31) / 32);
    return nblk1 >= minblk ? 2 : 0;
}

int launch_conv3_wino4(ConvArgs a, hipStream_t s) {
    a.tilesD = cdiv(a.D, 4); a.tilesH = cdiv(a.H, 4); a.tilesW = cdiv(a.W, 16);
    a.o_td = a.o_th = a.o_tw = 0;
    a.org_d = a.org_h = a.org_w = 0;
    if (a.box_hi[0] > 0) {      // needed region: the bricks that meet the box
        E3_REQUIRE(!a.stats, E3_ERR_INVALID, "conv with a needed region: no statistics");
        const int dims[3] = {a.D, a.H, a.W}, edge[3] = {4, 4, 16}, tile[3] = {2, 2, 4};
        int o[3], n[3];
        for (int i = 0; i < 3; ++i) {
            const int lo = a.box_lo[i] < 0 ? 0 : a.box_lo[i], hi = a.box_hi[i] > dims[i] ? dims[i] : a.box_hi[i];
            E3_REQUIRE(hi > lo, E3_ERR_INVALID, "conv with a needed region: empty box");
            // bricks start at the box's low corner rounded down to a TILE origin (2, 2, 4): every voxel keeps the Winograd tile it has in the
            // whole-tensor launch -- same arithmetic, bit-identical values
            o[i] = lo / tile[i] * tile[i];
            n[i] = cdiv(hi - o[i], edge[i]);
        }
        a.org_d = o[0]; a.org_h = o[1]; a.org_w = o[2];
        a.tilesD = n[0]; a.tilesH = n[1]; a.tilesW = n[2];
    }
    {   // store box: clipped to the tensor; off = the whole tensor
        const int dims[3] = {a.D, a.H, a.W};
        const bool on = a.sbox_hi[0] > 0;
        E3_REQUIRE(!on || (!a.stats && !(a.flags & CF_BNRED)), E3_ERR_INVALID, "conv with a store box: no statistics");
        for (int i = 0; i < 3; ++i) {
            a.sbox_lo[i] = on ? (a.sbox_lo[i] < 0 ? 0 : a.sbox_lo[i]) : 0;
            a.sbox_hi[i] = on ? (a.sbox_hi[i] > dims[i] ? dims[i] : a.sbox_hi[i]) : dims[i];
        }
    }
    a.NPad = (a.Ncols + 31) / 32 * 32;
    a.ntiles = a.NPad / 32;
    if (a.stats) a.cu_reserve = 0;      // (the statistic records are sized for the full grid; only data gradients run beside a collective)
    const size_t nblk = (size_t)a.N * a.tilesD * a.tilesH * a.tilesW * a.ntiles;
    E3_REQUIRE(nblk > 0 && nblk < (1u << 31), E3_ERR_INVALID, "conv grid out of range");
    E3_REQUIRE((size_t)6 * a.H * a.W * (size_t)(a.x_ldc > a.y_ldc ? a.x_ldc : a.y_ldc) * 4 < 0x7fffffffu, E3_ERR_UNSUPPORTED,
               "six d-planes of the conv input/output view exceed 2^31 bytes (32-bit buffer offsets)");
    E3_REQUIRE((a.y_ldc & 3) == 0 && (a.x_ldc & 3) == 0 && ((uintptr_t)a.y & 15) == 0 && ((uintptr_t)a.x & 15) == 0 && (a.Ncols & 3) == 0, E3_ERR_INVALID,
               "Winograd conv (F(2x2x4) tiles): views must be 16-byte aligned with channel counts that are multiples of 4");
    E3_REQUIRE(!(a.epi_scale && a.stats), E3_ERR_INVALID, "Winograd conv: statistics and the folded epilogue exclude each other");
    E3_REQUIRE(a.splitk <= 1 && !a.pro_scale, E3_ERR_INVALID, "Winograd conv (F(2x2x4) tiles): no split-K, no fused prologue");
    E3_REQUIRE(!a.x_chunk || (a.x_chunk == (size_t)a.N * a.D * a.H * a.W * 8 && chunked_layout_ok((size_t)a.N * a.D * a.H * a.W, a.Cin)), E3_ERR_INVALID,
               "Winograd conv (F(2x2x4) tiles): bad channel-chunked input");
    E3_REQUIRE(!a.y_chunk || (a.y_chunk == (size_t)a.N * a.D * a.H * a.W * 8 && chunked_layout_ok((size_t)a.N * a.D * a.H * a.W, a.Ncols) && !a.stats &&
                              !(a.flags & CF_BNRED) && !(a.head_w && a.head_done)), E3_ERR_INVALID,
               "Winograd conv (F(2x2x4) tiles): bad channel-chunked output (no statistics, no fused head; a fused pool's output stays in rows)");
    constexpr int lds = V_LDS_FLOATS * 4;
    constexpr int lds_x = lds;
    static bool attr = false;
    if (!attr) {
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino4_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_x));
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino4_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_x));
        attr = true;
    }
    // one workgroup per CU (cu_reserve > 0, a multiple of 8: that many CUs are left to the resident workgroups of a collective on a side stream)
    const unsigned full = 256u - (unsigned)((a.cu_reserve < 0 ? 0 : (a.cu_reserve > 128 ? 128 : a.cu_reserve)) & ~7);
    const unsigned grid = nblk >= full ? full : (unsigned)nblk;
    const int wgstats = (a.stats && wino4_wgstats(nblk, a.ntiles, grid)) ? 1 : 0;
    // a workgroup's bricks are L0, L0 + grid / 8, ... in the logical (XCD-blocked) order: digits of that step for the division-free brick cursors
    const W4PArgs pa = brick_step_make(grid == nblk ? 1u : grid / 8u, grid / 8u, a.ntiles, a.tilesW, a.tilesH, a.tilesD, 64);
    static const bool no_head = getenv("E3_WINO_NO_HEAD") != nullptr, no_pool = getenv("E3_WINO_NO_POOL") != nullptr;      // A/B switches (shared with conv_wino.hip)
    if (a.flags & CF_BNRED) {
        E3_REQUIRE(a.br_x && a.br_scale && a.br_shift && a.br_mean && a.br_invstd && a.br_part && !a.stats && !a.epi_scale && !a.bias && a.box_hi[0] <= 0 &&
                   (a.br_ldc & 3) == 0 && ((uintptr_t)a.br_x & 15) == 0 && a.cu_reserve == 0 && a.br_cols >= 0 && a.br_cols <= a.Ncols && (a.br_cols & 31) == 0, E3_ERR_INVALID,
                   "conv with the fused BatchNorm-backward reduction: bad arguments");
        E3_REQUIRE(wino4_wgstats(nblk, a.ntiles, grid), E3_ERR_INVALID, "conv with the fused BatchNorm-backward reduction: the grid does not tile (conv_wino4_bnred_parts)");
        static bool attr2 = false;
        if (!attr2) { E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino4_kernel<false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr2 = true; }
        hipLaunchKernelGGL((conv3_wino4_kernel<false, false, false, true>), dim3(grid), dim3(256), lds, s, a, (unsigned)nblk, 0, pa);
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    if (a.epi_scale && a.head_w && a.head_done && !no_head && a.Ncols == 32 && a.head_cout >= 1 && a.head_cout <= 4 && !a.pool_out) {      // + the 1x1x1 head behind it
        hipLaunchKernelGGL((conv3_wino4_kernel<true, false, true>), dim3(grid), dim3(256), lds_x, s, a, (unsigned)nblk, 0, pa);
        *a.head_done = 1;
    } else if (a.epi_scale && a.pool_out && a.pool_done && !no_pool && a.box_hi[0] <= 0 && (a.Ncols & 31) == 0) {      // + the max-pool behind it
        hipLaunchKernelGGL((conv3_wino4_kernel<true, true>), dim3(grid), dim3(256), lds_x, s, a, (unsigned)nblk, 0, pa);
        *a.pool_done = 1;
    } else if (a.epi_scale) hipLaunchKernelGGL(conv3_wino4_kernel<true>, dim3(grid), dim3(256), lds, s, a, (unsigned)nblk, wgstats, pa);
    else hipLaunchKernelGGL(conv3_wino4_kernel<false>, dim3(grid), dim3(256), lds, s, a, (unsigned)nblk, wgstats, pa);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
