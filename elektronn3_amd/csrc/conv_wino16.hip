// 3x3x3 stride-1 convolution as Winograd F(2x2x2, 3x3x3) on the fp32 matrix cores, second decomposition: 16-tile bricks,
// TWO workgroups per CU (forward and dgrad of torch.nn.Conv3d(k=3, padding=1) in elektronn3's conv3 blocks, unet.py:131-149).
//
// conv_wino.hip's kernels keep the 64 positions x (32 tiles x 32 channels) of a brick in 4 waves x 256 accumulator registers: one
// wave per SIMD, every stall of that wave (LDS store issue, scattered halo loads, barriers, the whole epilogue) is idle matrix-pipe
// time (measured: 0.58 of the fp32 MFMA peak, 6.0 k cycles per 4.1 k-cycle chunk, 5.4 k-cycle epilogue).  Here a brick is half as
// large -- 2x2x4 tiles = 4x4x8 output voxels, 6x6x10 halo -- so a wave's 16 positions need 128 registers and a lane needs < 256:
// two independent workgroups share a CU, and whatever one of them waits for, the other one's MFMAs fill (the planar kernels of
// conv_wino2d.hip run that way).  What changes with the brick:
//   * v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand: D[co][tile], lane l holds tile l & 15 and the four consecutive
//     channels 4 (l >> 4) .. + 3 of a 16-channel half -> the output leaves as 16-byte stores (channel order inside a 32-channel tile is
//     permuted by the packer so that a lane's two halves are 8 consecutive channels, 4 lanes = one 128-byte voxel row);
//   * the B operand is the transformed input: lane (tile, kk = l >> 4) feeds channels 2 kk, 2 kk + 1 of the 8-channel chunk (two
//     k-steps), so the H / W passes of B^T d B run on float2 = one packed op each, and the lane that computes a value feeds it;
//   * BatchNorm statistics: a lane owns 8 channels of one tile; it keeps a running (n, mean[8], M2[8]) over the workgroup's bricks in
//     LDS (Chan merge of the brick's two values per channel) and the workgroup writes ONE record at the end (512 / ntiles per layer).
// Same arithmetic as conv_wino.hip (same transform matrices, fp32 fmaf chains on the matrix cores) with a different summation order
// over the input channels of a chunk; which convs take this kernel is decided by conv_wino_layout() per SAMPLE grid, so results do
// not depend on the batch size.
#include <type_traits>
#include "kernels.h"

#ifndef E3_W16_PRIO
#define E3_W16_PRIO 3       // s_setprio of the VALU phases (transform, epilogue) over the other workgroup's MFMA phase
#endif
#ifndef E3_W16_ROWS
#define E3_W16_ROWS 2       // window rows read from LDS per round (1: 16, 2: 32 registers of raw values in flight)
#endif
#ifndef E3_W16_ABL
#define E3_W16_ABL 0       // developer builds: bit mask of pieces left out (timing experiments, wrong results)
#endif

namespace {

constexpr int Q_CLASS = 16;                                 // slots per (zh, zw) parity class of a plane (3 x 5 = 15 used)
constexpr int Q_RPLANE = 4 * Q_CLASS * 8 + 16;              // floats of one raw d-plane of the halo + 64 B skew: the planes of tile depth 0 / 1 (two planes apart) sit 128 B mod 256 B apart
constexpr int Q_RBUF = 6 * Q_RPLANE;                        // 6 raw planes: 12.4 KB per stage buffer
constexpr int Q_EX = 4 * 8 * 64 * 4;                        // epilogue exchange [pd][(oh, ow, half)][lane][4] floats (32 KB)
constexpr int Q_SCR = 4 * 32 * 3;                           // cross-wave merge of the statistics
constexpr int Q_RUN = 20 * 256;                             // running statistics of every thread: n, mean[8], M2[8] ([k][thread]; BNRED: [thread][20], 16 sums)
constexpr int Q_KST = 4 * 32;                               // BNRED: scale / shift / mean / invstd of the workgroup's 32 channels
constexpr int Q_LDS_FLOATS = 2 * Q_RBUF + Q_EX + Q_SCR + Q_RUN + Q_KST;   // 76 KB: two workgroups per CU

typedef float f32x2q __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned u32x4q;
typedef __attribute__((address_space(3))) void* lds_ptr_q;

// Layout of a raw plane: halo voxel (zh, zw) -> slot ((zh & 1) * 2 + (zw & 1)) * Q_CLASS + (zh >> 1) * 5 + (zw >> 1) of 32 bytes (parity classes keep
// the stride-2 tile origins contiguous); its 16-byte half q sits at q ^ ((zh >> 1) & 1), so that the 32 lanes of a ds_read_b64 group (16 tiles x 2
// channel pairs) cover all 64 banks.

// BNRED (data gradients): the epilogue also takes the REDUCE sums of the BatchNorm backward of the unit in front (ConvArgs::br_*)
template <bool AFF, bool BNRED = false>
__global__ __launch_bounds__(256, 2) void conv3_wino16_kernel(const ConvArgs a, const unsigned nblk, const int wgstats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tl = lane & 15, kk = lane >> 4;
    const int ttd = tl >> 3, tth = (tl >> 2) & 1, ttw = tl & 3;
    const int NCH = a.Cin >> 3;
    constexpr unsigned OOB = 0x80000000u;       // buffer offset beyond every descriptor below: loads return 0, stores are dropped
    // arguments that only the per-brick set-up and the epilogue need are re-read from the kernarg segment there (scalar loads) instead of
    // occupying SGPRs across the chunk loop (hipcc spilled ~70 scalars into vector lanes)
    typedef const __attribute__((address_space(4))) ConvArgs* KArgs;
    auto KA = []() -> KArgs { KArgs q = (KArgs)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(q)); return q; };
    const int D = a.D, H = a.H, W = a.W, xl = a.x_ldc;
    const unsigned plane_xb = (unsigned)((size_t)H * W * xl * 4);

    // ---- staging by LDS-DMA (buffer_load_dwordx4 ... lds: 1 KB per wave-instruction, lane i lands at base + 16 i): a raw d-plane of the
    // halo is two such pieces, a chunk is 12; wave w issues pieces w, w + 4, w + 8 = half hp = w & 1 of the planes w >> 1, 2 + (w >> 1),
    // 4 + (w >> 1).  The layout of a plane (parity classes, XOR-ed halves) is produced on the SOURCE side: lane i of half hp asks for the
    // 16 bytes that belong at piece g = 64 hp + i.
    const int hp = wave & 1;
    unsigned col_rel, col_bits;
    {
        const int g = hp * 64 + lane, slot = g >> 1, qd = g & 1;
        const int cls = slot >> 4, sic = slot & 15;
        const int zhh = sic / 5, zwh = sic % 5;
        const int zh = 2 * zhh + (cls >> 1), zw = 2 * zwh + (cls & 1), q = qd ^ (zhh & 1);
        const bool used = sic < 15;
        col_rel = (unsigned)(((zh * W + zw) * xl + 4 * q) * 4);
        col_bits = used ? (1u << (6 + zh)) | (1u << (12 + zw)) : 0xffffffffu;     // (all-ones never matches: the unused slots get zeros)
    }
    float m1 = -1.f;
    asm volatile("" : "+s"(m1));     // opaque -1: a + m1*b becomes v_pk_fma_f32 (hipcc only packs fadd/ffma, never fsub)

    // ---- read plan of lane (tile tl, channel pair kk).  The D pass of B^T is done while reading: row pd = wave of the tile depth's 4
    // raw planes is  x[A] + sgn x[B]  with (A, B, sgn) = (0, 2, -), (1, 2, +), (2, 1, -), (1, 3, -).  Window rows h = 0, 1 have zh/2 = tth,
    // rows 2, 3 have tth + 1 (the XOR of the 16-byte half follows).
    const int pA = wave == 0 ? 0 : (wave == 2 ? 2 : 1), pB = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float dsg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(wave == 1 ? 0x3f800000 : (int)0xbf800000));
    int rdA[2], rdB[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int r = ((tth + hh) * 5 + ttw) * 8 + 2 * (kk ^ (2 * ((tth + hh) & 1)));
        rdA[hh] = (2 * ttd + pA) * Q_RPLANE + r; rdB[hh] = (2 * ttd + pB) * Q_RPLANE + r;
    }
    auto rd_imm = [](int h, int w) { return (((h & 1) * 2 + (w & 1)) * Q_CLASS + (w >> 1)) * 8; };

    float* cur = smem;
    float* nxt = smem + Q_RBUF;
    float* const ex = smem + 2 * Q_RBUF;
    float* const scr = ex + Q_EX;
    float* const run = scr + Q_SCR + tid;
#ifdef E3_W16_TIMING
    const bool do_stats = false;
#else
    const bool do_stats = !AFF && !BNRED && a.stats != nullptr;
#endif
    if (do_stats || BNRED) {
#pragma unroll
        for (int k = 0; k < 20; ++k) run[k * 256] = 0.f;
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
    auto divmod = [](unsigned& x, int d) {
        int r;
        if ((d & (d - 1)) == 0) { r = (int)(x & (unsigned)(d - 1)); x >>= __builtin_ctz((unsigned)d); }
        else { r = (int)(x % (unsigned)d); x /= (unsigned)d; }
        return r;
    };
    auto range_mask = [](int lo, int n, int size) {      // bit z set: lo + z in [0, size), z in [0, n)
        const int first = lo < 0 ? -lo : 0, last = size - lo < n ? size - lo : n;
        return last > first ? ((1u << last) - 1u) & ~((1u << first) - 1u) : 0u;
    };
    // a brick: coordinates, halo descriptor at voxel (d0 - 1, h0 - 1, w0 - 1) (possibly in front of the tensor: only valid lanes form
    // addresses from it), validity mask (6 d bits | 6 h bits | 10 w bits), the wave's transformed weights
    //   U[ntile][chunk][pos 64][lane 64][ks 2][half 2]; wave = pd owns positions 16 pd .. 16 pd + 15
    // (plain scalars, no struct: selecting between two structs' fields made hipcc keep them in scratch memory)
#define E3_BRICK_VARS(X) int X##d0, X##h0, X##w0, X##nb, X##n0, X##row; unsigned X##mask; const float* X##xorg; const float* X##wbase
#define E3_DECODE(X, Lval, on) do {                                                                                                   \
        unsigned Lq_ = (Lval);                                                                                                         \
        const KArgs k_ = KA();                                                                                                         \
        const int ntile_ = divmod(Lq_, k_->ntiles);                                                                                      \
        const int tw_ = divmod(Lq_, k_->tilesW);                                                                                         \
        const int th_ = divmod(Lq_, k_->tilesH);                                                                                         \
        const int td_ = divmod(Lq_, k_->tilesD);                                                                                         \
        X##nb = (int)Lq_;                                                                                                              \
        X##d0 = (td_ + k_->o_td) * 4; X##h0 = (th_ + k_->o_th) * 4; X##w0 = (tw_ + k_->o_tw) * 8;   /* (o_*: first brick of the needed region) */ \
        X##n0 = ntile_ * 32;                                                                                                           \
        X##row = ((X##nb * k_->tilesD + td_) * k_->tilesH + th_) * k_->tilesW + tw_;                                                         \
        X##xorg = k_->x + (((long long)X##nb * D + (X##d0 - 1)) * ((long long)H * W * xl) + ((long long)(X##h0 - 1) * W + (X##w0 - 1)) * xl); \
        const unsigned m_ = range_mask(X##d0 - 1, 6, D) | (range_mask(X##h0 - 1, 6, H) << 6) | (range_mask(X##w0 - 1, 10, W) << 12);    \
        X##mask = (on) ? m_ : 0u;       /* (a brick beyond the end of the range stages zeros) */                                        \
        X##wbase = k_->wt + ((size_t)ntile_ * NCH * 64 + wave * 16) * 256;                                                               \
    } while (0)
    // the 3 DMA pieces of this wave for chunk cb of a brick (halo origin xorg, validity mask), into stage buffer `buf`
    auto issue_dma = [&](const float* xorg, unsigned mask, int cb, float* buf) {
        if (E3_W16_ABL & 1) return;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xorg), 0, 0x7fffffff, 0x00020000);
        const bool ok = (mask & col_bits) == col_bits;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int plane = 2 * k + (wave >> 1);
            const unsigned dsel = ((mask >> plane) & 1u) ? 0u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_q)(buf + plane * Q_RPLANE + hp * 256), 16, (ok ? col_rel : OOB) | dsel,
                                                     (int)(plane * plane_xb) + cb * 32, 0, 0);
        }
    };
    const int b_voff = lane * 16;
    f32x4 acc[16][2];
    f32x4 Bv[16];

#ifdef E3_W16_TIMING      // developer build (tools/phase_timing_w16.py): s_memtime stamps of the workgroup's third brick instead of statistics
    long long* const tstamp = reinterpret_cast<long long*>(KA()->stats) + (size_t)blockIdx.x * 40;
    int tbrick = 0;
#define TSTAMP(i) do { if (tid == 0 && tbrick == 2) tstamp[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i)
#endif
    E3_BRICK_VARS(P_); E3_BRICK_VARS(N_);
    E3_DECODE(P_, L, true);
    if (BNRED && tid < 128) {      // (one column tile per workgroup in this mode: the constants of its 32 channels, once)
        const KArgs e = KA();
        const float* const src = (tid >> 5) == 0 ? e->br_scale : ((tid >> 5) == 1 ? e->br_shift : ((tid >> 5) == 2 ? e->br_mean : e->br_invstd));
        const int ch = P_n0 + (tid & 31);
        (scr + Q_SCR + Q_RUN)[tid] = ch < e->Ncols ? src[ch] : 0.f;
    }
    {   // prologue: unit 0 staged, its weights requested
        issue_dma(P_xorg, P_mask, 0, cur);
        const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P_wbase), 0, NCH * 64 * 1024, 0x00020000);
#pragma unroll
        for (int p = 0; p < 16; ++p) Bv[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_voff, p * 1024, 0));
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    for (;;) {
        const bool has_next = L + Lstep < Lend;
        E3_DECODE(N_, has_next ? L + Lstep : L, has_next);

        // One 8-channel chunk c of brick P (raw halo in `cur`).  Program order: LDS reads + transform; the DMA requests of the NEXT unit
        // (chunk c + 1, or chunk 0 of brick N) into `nxt`; 16 positions x 4 MFMAs, each position pair followed by the weight requests of
        // the same positions of the next unit (a ring of 16: one full chunk of look-ahead).  The memory counter retires in order, so the
        // order of the requests is what hides their latency: DMA (long, HBM) before the 16 weight loads (short, L2) of the same chunk, both
        // needed only after this chunk's MFMAs; stores of an epilogue are younger than everything the next chunk waits for.
        auto chunk = [&](auto zero_tag, int c) {
            constexpr bool ZERO = decltype(zero_tag)::value;
            const bool lastc = c + 1 == NCH;
            if (E3_W16_PRIO) __builtin_amdgcn_s_setprio(E3_W16_PRIO);      // the VALU phase goes in front of the other workgroup's MFMAs
            f32x2q t[4][4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {        // (row by row: at most 16 registers of raw values in flight)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const f32x2q xa = *reinterpret_cast<const f32x2q*>(cur + rdA[h >> 1] + rd_imm(h, w));
                    const f32x2q xb = *reinterpret_cast<const f32x2q*>(cur + rdB[h >> 1] + rd_imm(h, w));
                    t[h][w] = xa + dsg * xb;
                }
                if (E3_W16_ROWS == 1 || (h & 1)) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x2q u0 = t[0][w] + m1 * t[2][w], u1 = t[1][w] + t[2][w], u2 = t[2][w] + m1 * t[1][w], u3 = t[1][w] + m1 * t[3][w];
                t[0][w] = u0; t[1][w] = u1; t[2][w] = u2; t[3][w] = u3;
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f32x2q u0 = t[h][0] + m1 * t[h][2], u1 = t[h][1] + t[h][2], u2 = t[h][2] + m1 * t[h][1], u3 = t[h][1] + m1 * t[h][3];
                t[h][0] = u0; t[h][1] = u1; t[h][2] = u2; t[h][3] = u3;
            }
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int w = 0; w < 4; ++w) asm volatile("" : "+v"(t[h][w]));     // keep the transform packed and in front of the MFMA block
            __builtin_amdgcn_sched_barrier(0);
            if (E3_W16_PRIO) __builtin_amdgcn_s_setprio(0);
            if (c < 8) TSTAMP(1 + 4 * c);
            issue_dma(lastc ? N_xorg : P_xorg, lastc ? N_mask : P_mask, lastc ? 0 : c + 1, nxt);
            __builtin_amdgcn_sched_barrier(0);
            const float* const wb = lastc ? N_wbase : P_wbase;
            const __amdgpu_buffer_rsrc_t b_nx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wb), 0, (lastc && !has_next) ? 0 : NCH * 64 * 1024, 0x00020000);
            const int cB = lastc ? 0 : c + 1;
#pragma unroll
            for (int pp = 0; pp < 16; pp += 2) {        // two positions at a time: 4 independent accumulators in flight, then their ring slots are refilled
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int p = pp; p < pp + 2; ++p)
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            if (E3_W16_ABL & 32) continue;
                            const float wv = Bv[p][ks * 2 + hf], tv = t[p >> 2][p & 3][ks];
                            if (ZERO && ks == 0) {
                                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                                acc[p][hf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, tv, z, 0, 0, 0);
                            } else
                                acc[p][hf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, tv, acc[p][hf], 0, 0, 0);
                        }
#pragma unroll
                for (int p = pp; p < pp + 2; ++p) {
                    if (E3_W16_ABL & 2) continue;
                    Bv[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_nx, b_voff, (cB * 64 + p) * 1024, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // this wave's DMA pieces have landed once at most the 16 weight requests behind them are outstanding; the barrier then
            // publishes every wave's pieces and retires `cur`
            if (c < 8) TSTAMP(2 + 4 * c);
            if (E3_W16_ABL & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (c < 8) TSTAMP(3 + 4 * c);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (c < 8) TSTAMP(4 + 4 * c);
            { float* tsw = cur; cur = nxt; nxt = tsw; }
        };

        TSTAMP(0);
        chunk(std::true_type{}, 0);
        for (int c = 1; c < NCH; ++c) chunk(std::false_type{}, c);

        // ---- epilogue.  acc[ph*4+pw][half][i]: position (pd = wave, ph, pw), tile tl, channel n0 + 8 kk + 4 half + i.
        // per-channel constants first: they arrive during the output transform
        if (E3_W16_PRIO) __builtin_amdgcn_s_setprio(E3_W16_PRIO);
        const int n0 = P_n0, d0 = P_d0, h0 = P_h0, w0 = P_w0;
        const int nq = n0 + 8 * kk;
        f32x4 bias[2], es[2], eh[2];
        {
            const KArgs e = KA();
            const int eN = e->Ncols;
            const __amdgpu_buffer_rsrc_t c_rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e->bias), 0, e->bias ? eN * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t c_rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e->epi_scale), 0, AFF ? eN * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t c_rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e->epi_shift), 0, AFF ? eN * 4 : 0, 0x00020000);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                bias[hf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c_rs0, (nq + 4 * hf) * 4, 0, 0));
                if (AFF) {
                    es[hf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c_rs1, (nq + 4 * hf) * 4, 0, 0));
                    eh[hf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c_rs2, (nq + 4 * hf) * 4, 0, 0));
                }
            }
        }
        // BNRED: the raw tensor of the unit in front at this lane's two voxels (requested once the accumulators are dead, used behind the stores)
        f32x4 bx[2][2];
        const int gh = h0 + 2 * tth + (wave >> 1), gw = w0 + 2 * ttw + (wave & 1), gd = d0 + 2 * ttd;
        const bool vox_ok = gh < H && gw < W;
        const bool ok0 = vox_ok && gd < D, ok1 = vox_ok && gd + 1 < D;
        auto issue_bx = [&]() {
        if (BNRED) {
            const KArgs e = KA();
            const int bl = e->br_ldc;
            const size_t plane_b = (size_t)H * W * bl;
            const size_t brem = (size_t)(D - d0) * plane_b * 4;
            const __amdgpu_buffer_rsrc_t x2_rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(e->br_x) + ((size_t)P_nb * D + d0) * plane_b, 0, (int)(brem < 0x7fffffffu ? brem : 0x7fffffffu), 0x00020000);
            const unsigned b_off = (unsigned)((((2 * ttd * H + gh) * W + gw) * bl + nq) * 4);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const bool cok = nq + 4 * hf < e->Ncols;
                bx[0][hf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x2_rs, (ok0 && cok) ? b_off + 16 * hf : OOB, 0, 0));
                bx[1][hf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x2_rs, (ok1 && cok) ? b_off + 16 * hf : OOB, (int)(plane_b * 4), 0));
            }
        }
        };
        // A^T m A over (ph, pw) in registers (`ex` is its own LDS region: the next brick's first chunk is already in the stage buffers)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f32x4 q[2][2];
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const f32x4 t0 = acc[ph * 4 + 0][hf] + acc[ph * 4 + 1][hf] + acc[ph * 4 + 2][hf];
                const f32x4 t1 = acc[ph * 4 + 1][hf] + m1 * acc[ph * 4 + 2][hf] + m1 * acc[ph * 4 + 3][hf];
                if (ph == 0) { q[0][0] = t0; q[0][1] = t1; }
                else if (ph == 1) { q[0][0] += t0; q[0][1] += t1; q[1][0] = t0; q[1][1] = t1; }
                else if (ph == 2) { q[0][0] += t0; q[0][1] += t1; q[1][0] += m1 * t0; q[1][1] += m1 * t1; }
                else { q[1][0] += m1 * t0; q[1][1] += m1 * t1; }
            }
#pragma unroll
            for (int oh = 0; oh < 2; ++oh)
#pragma unroll
                for (int ow = 0; ow < 2; ++ow)
                    *reinterpret_cast<f32x4*>(ex + ((wave * 8 + (oh * 2 + ow) * 2 + hf) * 64 + lane) * 4) = q[oh][ow];
            if (hf == 0) { __builtin_amdgcn_sched_barrier(0); issue_bx(); __builtin_amdgcn_sched_barrier(0); }      // (half of the accumulators are dead: 64 registers free)
        }
        TSTAMP(33);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        TSTAMP(34);
        // wave w now owns output offset (oh, ow) = (w >> 1, w & 1) of every tile and sums the pd axis: od = 0, 1
        const int oh = wave >> 1, ow = wave & 1;
        f32x4 y[2][2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f32x4 m[4];
#pragma unroll
            for (int pd = 0; pd < 4; ++pd) m[pd] = *reinterpret_cast<const f32x4*>(ex + ((pd * 8 + wave * 2 + hf) * 64 + lane) * 4);
            y[0][hf] = m[0] + m[1] + m[2] + bias[hf];
            y[1][hf] = m[1] + m1 * m[2] + m1 * m[3] + bias[hf];
            if (AFF) {
#pragma unroll
                for (int od = 0; od < 2; ++od)
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[od][hf][e] = fmaxf(__builtin_fmaf(y[od][hf][e], es[hf][e], eh[hf][e]), 0.f);
            }
        }
        // voxel (d0 + 2 ttd + od, h0 + 2 tth + oh, w0 + 2 ttw + ow), channels nq + 4 half .. + 3: 16-byte stores
        const int yl = KA()->y_ldc;
        const size_t plane_y = (size_t)H * W * yl;
        const size_t yrem = (size_t)(D - d0) * plane_y * 4;
        const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(
            KA()->y + ((size_t)P_nb * D + d0) * plane_y, 0, (int)(yrem < 0x7fffffffu ? yrem : 0x7fffffffu), 0x00020000);
        const unsigned y_voff = (unsigned)((((2 * ttd * H + gh) * W + gw) * yl + nq) * 4);
        const int od_off = (int)(plane_y * 4);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const bool cok = nq + 4 * hf < KA()->Ncols;
            if (E3_W16_ABL & 4) continue;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4q, y[0][hf]), y_rs, (ok0 && cok) ? y_voff + 16 * hf : OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4q, y[1][hf]), y_rs, (ok1 && cok) ? y_voff + 16 * hf : OOB, od_off, 0);
        }
        TSTAMP(35);
        if (BNRED) {
            // dz = dA * act'(z), z = x * scale + shift;  xhat = (x - mean) * invstd  (the expressions of bn_bwd_kernel); running sums of this
            // lane's 8 channels in LDS, one partial row per workgroup at the end
            const KArgs e = KA();
            const int eN = e->Ncols;
            const float slope = e->br_slope;
            const float* const kst = scr + Q_SCR + Q_RUN;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(kst + 0 * 32 + 8 * kk + 4 * hf);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(kst + 1 * 32 + 8 * kk + 4 * hf);
                const f32x4 mu = *reinterpret_cast<const f32x4*>(kst + 2 * 32 + 8 * kk + 4 * hf);
                const f32x4 is = *reinterpret_cast<const f32x4*>(kst + 3 * 32 + 8 * kk + 4 * hf);
                f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
                for (int od = 0; od < 2; ++od) {
                    const bool okd = od ? ok1 : ok0;
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float xv = bx[od][hf][e4];
                        const float z = __builtin_fmaf(xv, sc[e4], sh[e4]);
                        const float dz = okd ? act_bwd(z, y[od][hf][e4], slope) : 0.f;
                        const float xh = (xv - mu[e4]) * is[e4];
                        s1[e4] += dz; s2[e4] = __builtin_fmaf(dz, xh, s2[e4]);
                    }
                }
                float* const r2 = scr + Q_SCR + tid * 20 + 4 * hf;
                *reinterpret_cast<f32x4*>(r2) += s1;
                *reinterpret_cast<f32x4*>(r2 + 8) += s2;
            }
            if (!has_next) {       // (uniform) sums over the 16 tiles of a lane group, then the 4 waves, fixed order; row = the workgroup
                float f1[8], f2[8];
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) { f1[ch] = (scr + Q_SCR + tid * 20)[ch]; f2[ch] = (scr + Q_SCR + tid * 20)[8 + ch]; }
#pragma unroll
                for (int sft = 1; sft < 16; sft <<= 1)
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) { f1[ch] += __shfl_xor(f1[ch], sft); f2[ch] += __shfl_xor(f2[ch], sft); }
                if (tl == 0) {
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) {
                        float* sc_ = scr + (wave * 32 + 8 * kk + ch) * 3;
                        sc_[0] = f1[ch]; sc_[1] = f2[ch];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (tid < 32 && n0 + tid < eN) {
                    float t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) { t1 += scr[(w * 32 + tid) * 3]; t2 += scr[(w * 32 + tid) * 3 + 1]; }
                    const unsigned nt_ = (unsigned)e->ntiles;
                    const size_t row = (size_t)((blockIdx.x & 7u) * ((gridDim.x >> 3) / nt_) + (blockIdx.x >> 3) / nt_);
                    float* o = e->br_part + row * 3 * (size_t)eN + n0 + tid;
                    o[0] = t1; o[eN] = t2;
                }
            }
        }
        if (do_stats) {
            // running record of this lane's 8 channels: Chan merge of the brick's (up to) two values per channel, approximate reciprocal
            // (its error is far below the rounding of the sums)
            const float rn = run[0];
            const float cb = (ok0 ? 1.f : 0.f) + (ok1 ? 1.f : 0.f);
            const float nn = rn + cb;
            const float rf = cb * __builtin_amdgcn_rcpf(fmaxf(nn, 1.f));
            const float rc = cb == 2.f ? 0.5f : 1.f;
            const float rnf = rn * rf;
            const bool both = ok0 && ok1;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ch = hf * 4 + e;
                    const float v0 = ok0 ? y[0][hf][e] : 0.f, v1 = ok1 ? y[1][hf][e] : 0.f;
                    const float bm = (v0 + v1) * rc;
                    const float dd = both ? v0 - v1 : 0.f;
                    const float rmean = run[(1 + ch) * 256], rm2 = run[(9 + ch) * 256];
                    const float dl = bm - rmean;
                    run[(1 + ch) * 256] = rmean + dl * rf;
                    run[(9 + ch) * 256] = rm2 + 0.5f * dd * dd + dl * dl * rnf;
                }
            run[0] = nn;
            if (!wgstats || !has_next) {       // (uniform) merge the lanes of a channel (16 tiles, then the 4 waves) in a fixed order, one record
                float fn = run[0], fm[8], fs[8];
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) { fm[ch] = run[(1 + ch) * 256]; fs[ch] = run[(9 + ch) * 256]; }
#pragma unroll
                for (int sft = 1; sft < 16; sft <<= 1) {
                    const float n2 = __shfl_xor(fn, sft);
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) {
                        float na = fn;
                        welford_merge(na, fm[ch], fs[ch], n2, __shfl_xor(fm[ch], sft), __shfl_xor(fs[ch], sft));
                    }
                    fn += n2;
                }
                if (tl == 0) {
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) {
                        float* sc_ = scr + (wave * 32 + 8 * kk + ch) * 3;
                        sc_[0] = fn; sc_[1] = fm[ch]; sc_[2] = fs[ch];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (tid < 32 && n0 + tid < KA()->Ncols) {
                    float c0 = 0.f, me = 0.f, mm = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float* sc_ = scr + (w * 32 + tid) * 3;
                        welford_merge(c0, me, mm, sc_[0], sc_[1], sc_[2]);
                    }
                    const size_t row = wgstats ? (size_t)((blockIdx.x & 7u) * ((gridDim.x >> 3) / (unsigned)KA()->ntiles) + (blockIdx.x >> 3) / (unsigned)KA()->ntiles)
                                               : (size_t)P_row;
                    float* o = KA()->stats + (row * KA()->Cout + n0 + tid) * 3;
                    o[0] = c0; o[1] = me; o[2] = mm;
                }
                if (!wgstats) {
#pragma unroll
                    for (int k = 0; k < 17; ++k) run[k * 256] = 0.f;
                }
            }
        }
        if (E3_W16_PRIO) __builtin_amdgcn_s_setprio(0);
        TSTAMP(36);
#ifdef E3_W16_TIMING
        ++tbrick;
#endif
        if (!has_next) break;
        P_d0 = N_d0; P_h0 = N_h0; P_w0 = N_w0; P_nb = N_nb; P_n0 = N_n0; P_row = N_row; P_mask = N_mask; P_xorg = N_xorg; P_wbase = N_wbase;
        L += Lstep;
    }
}

}  // namespace

// ---- host side
int wino16_bricks(int N, int D, int H, int W) { return N * cdiv(D, 4) * cdiv(H, 4) * cdiv(W, 8); }

// one column tile per workgroup and every column tile covered by the workgroups of a row: XCD ranges that start at multiples of ntiles and a
// stride of 64 logical bricks that is one too
static bool wino16_wgstats(size_t nblk, int ntiles, unsigned grid) {
    return grid == 512u && nblk > 512 && nblk % 8 == 0 && (nblk / 8) % (size_t)ntiles == 0 && 64 % ntiles == 0;
}

int conv_wino16_bnred_parts(int N, int D, int H, int W, int K, int ncols) {
    if ((ncols & 3) || (K & 7)) return 0;
    const int ntiles = (ncols + 31) / 32;
    const size_t nblk = (size_t)wino16_bricks(N, D, H, W) * ntiles;
    return wino16_wgstats(nblk, ntiles, nblk > 512 ? 512u : (unsigned)nblk) ? 512 / ntiles : 0;
}

int wino16_stats_parts(int N, int D, int H, int W, int ncols) {
    const int bricks = wino16_bricks(N, D, H, W), ntiles = (ncols + 31) / 32;
    const size_t nblk = (size_t)bricks * ntiles;
    return wino16_wgstats(nblk, ntiles, nblk >= 512 ? 512u : (unsigned)nblk) ? 512 / ntiles : bricks;
}

int launch_conv3_wino16(ConvArgs a, hipStream_t s) {
    a.tilesD = cdiv(a.D, 4); a.tilesH = cdiv(a.H, 4); a.tilesW = cdiv(a.W, 8);
    a.o_td = a.o_th = a.o_tw = 0;
    if (a.box_hi[0] > 0) {      // needed region: the bricks that meet the box
        E3_REQUIRE(!a.stats, E3_ERR_INVALID, "conv with a needed region: no statistics");
        const int dims[3] = {a.D, a.H, a.W}, edge[3] = {4, 4, 8};
        int o[3], n[3];
        for (int i = 0; i < 3; ++i) {
            const int lo = a.box_lo[i] < 0 ? 0 : a.box_lo[i], hi = a.box_hi[i] > dims[i] ? dims[i] : a.box_hi[i];
            E3_REQUIRE(hi > lo, E3_ERR_INVALID, "conv with a needed region: empty box");
            o[i] = lo / edge[i]; n[i] = cdiv(hi, edge[i]) - o[i];
        }
        a.o_td = o[0]; a.o_th = o[1]; a.o_tw = o[2];
        a.tilesD = n[0]; a.tilesH = n[1]; a.tilesW = n[2];
    }
    a.NPad = (a.Ncols + 31) / 32 * 32;
    a.ntiles = a.NPad / 32;
    if (a.stats) a.cu_reserve = 0;      // (the statistic records are sized for the full grid; only data gradients run beside a collective)
    const size_t nblk = (size_t)a.N * a.tilesD * a.tilesH * a.tilesW * a.ntiles;
    E3_REQUIRE(nblk > 0 && nblk < (1u << 31), E3_ERR_INVALID, "conv grid out of range");
    E3_REQUIRE((size_t)6 * a.H * a.W * (size_t)(a.x_ldc > a.y_ldc ? a.x_ldc : a.y_ldc) * 4 < 0x7fffffffu, E3_ERR_UNSUPPORTED,
               "six d-planes of the conv input/output view exceed 2^31 bytes (32-bit buffer offsets)");
    E3_REQUIRE((a.y_ldc & 3) == 0 && (a.x_ldc & 3) == 0 && ((uintptr_t)a.y & 15) == 0 && ((uintptr_t)a.x & 15) == 0 && (a.Ncols & 3) == 0, E3_ERR_INVALID,
               "Winograd conv (16-tile bricks): views must be 16-byte aligned with channel counts that are multiples of 4");
    E3_REQUIRE(!(a.epi_scale && a.stats), E3_ERR_INVALID, "Winograd conv: statistics and the folded epilogue exclude each other");
    E3_REQUIRE(a.splitk <= 1 && !a.pro_scale, E3_ERR_INVALID, "Winograd conv (16-tile bricks): no split-K, no fused prologue");
    constexpr int lds = Q_LDS_FLOATS * 4;
    static bool attr = false;
    if (!attr) {
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    // two workgroups per CU (cu_reserve > 0, a multiple of 8: that many CUs are left to the resident workgroups of a collective on a side stream)
    const unsigned full = 2u * (256u - (unsigned)((a.cu_reserve < 0 ? 0 : (a.cu_reserve > 128 ? 128 : a.cu_reserve)) & ~7));
    static const unsigned gover = getenv("E3_W16_GRID") ? (unsigned)atoi(getenv("E3_W16_GRID")) : 0u;      // (timing experiments)
    const unsigned gcap = (gover && !a.stats) ? gover : full;
    const unsigned grid = nblk >= gcap ? gcap : (unsigned)nblk;
    const int wgstats = (a.stats && wino16_wgstats(nblk, a.ntiles, grid)) ? 1 : 0;
    if (a.flags & CF_BNRED) {
        E3_REQUIRE(a.br_x && a.br_scale && a.br_shift && a.br_mean && a.br_invstd && a.br_part && !a.stats && !a.epi_scale && !a.bias && a.box_hi[0] <= 0 &&
                   (a.br_ldc & 3) == 0 && ((uintptr_t)a.br_x & 15) == 0 && a.cu_reserve == 0, E3_ERR_INVALID, "conv with the fused BatchNorm-backward reduction: bad arguments");
        E3_REQUIRE(wino16_wgstats(nblk, a.ntiles, grid), E3_ERR_INVALID, "conv with the fused BatchNorm-backward reduction: the grid does not tile (conv_wino16_bnred_parts)");
        static bool attr2 = false;
        if (!attr2) { E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino16_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr2 = true; }
        hipLaunchKernelGGL((conv3_wino16_kernel<false, true>), dim3(grid), dim3(256), lds, s, a, (unsigned)nblk, 0);
    } else if (a.epi_scale) hipLaunchKernelGGL(conv3_wino16_kernel<true>, dim3(grid), dim3(256), lds, s, a, (unsigned)nblk, wgstats);
    else hipLaunchKernelGGL(conv3_wino16_kernel<false>, dim3(grid), dim3(256), lds, s, a, (unsigned)nblk, wgstats);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
