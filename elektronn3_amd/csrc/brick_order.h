// Logical order of the 4 x 4 x 16-voxel bricks of the persistent Winograd kernels (conv3_wino_pkernel, conv3_wino4_kernel).
//
// An XCD's 32 workgroups work on 32 CONSECUTIVE logical bricks at a time (the XCD owns a contiguous eighth of the logical range, its workgroups walk it
// with stride gridDim / 8), and its L2 serves the halo voxels those bricks share.  In the plain order (column tile, tw, th, td, sample) such a set is a slab of
// one brick depth -- 4 x 16 x 128 voxels at level 0, whose halo is 6 x 18 x 128: the kernels read 1.7x their input (measured: 1.77x, profiles/r05_pmc_f32.md).
// Here the three spatial digits are split into a low part (position inside a block of bd x bh x bw bricks, powers of two that divide the brick counts)
// and a high part (the block):
//     L = nt + ntiles * (lw + bw * (lh + bh * (ld + bd * (hw + Tw * (hh + Th * (hd + Td * nb))))))        Tw = tilesW / bw, ...
// so that the concurrent set is one block, e.g. 16 x 16 x 32 voxels with a halo of 18 x 18 x 34 (1.34x).  bd = bh = bw = 1 is the plain order.
// The cursors keep tw / th / td whole (low part = the low bits) and advance by the digits of the workgroup's step with carries: no division per brick.
#pragma once

struct BrickStep {
    int s_nt, s_lw, s_lh, s_ld, s_hw, s_hh, s_hd, s_nb;      // digits of the step between a workgroup's bricks
    int kw, kh, kd;                                          // log2 of the block's brick counts
};

#include <cstdio>
#include <cstdlib>
// blocks for `concurrent` consecutive logical indices (gridDim / 8) of a grid with ntiles column tiles: W grows first up to runs of w_run voxels, then the axis
// with the smallest voxel extent, while the block stays within the concurrent set and divides the brick counts.  w_run: short runs at the power-of-two row
// strides of these tensors land in few of the L2's channels -- measured on the 64-channel concat input of cfg 1's level 0 (256 B per voxel, ideal 537 MB):
// blocks 1 brick wide read 1 917 MB, 2 wide 1 491 MB, 4 wide 831 MB, the plain order 953 MB -- and the most compact blocks (2 x 4 x 4 bricks), which read
// the least on 128-byte voxels, cost 0.5 % of the step; 4 x 2 x 4 bricks cost 0.2 % and read 11 % less than the plain order (profiles/r06_brick_order.md).
// E3_WINO_BLOCK=0: the plain order (A/B switch); E3_WINO_BLOCK=kw,kh,kd: a forced block (developer sweeps).
inline BrickStep brick_step_make(unsigned step, unsigned concurrent, int ntiles, int tilesW, int tilesH, int tilesD, int w_run) {
    static const char* const env = getenv("E3_WINO_BLOCK");
    const bool plain = env && env[0] == '0' && !env[1];
    BrickStep b{};
    if (env && env[0] && env[1]) {       // developer override "kw,kh,kd" (clamped to what divides the brick counts)
        int k[3] = {0, 0, 0};
        sscanf(env, "%d,%d,%d", &k[0], &k[1], &k[2]);
        while (b.kw < k[0] && tilesW % (2 << b.kw) == 0) ++b.kw;
        while (b.kh < k[1] && tilesH % (2 << b.kh) == 0) ++b.kh;
        while (b.kd < k[2] && tilesD % (2 << b.kd) == 0) ++b.kd;
    }
    unsigned room = (plain || (env && env[0] && env[1])) ? 0u : concurrent / (unsigned)ntiles;
    for (; room >= 2 && (16 << b.kw) < w_run && tilesW % (2 << b.kw) == 0; room >>= 1) ++b.kw;      // W first, up to runs of w_run voxels
    while (room >= 2) {      // then the axis with the smallest voxel extent (D, H, W on ties)
        const int ew = 16 << b.kw, eh = 4 << b.kh, ed = 4 << b.kd;
        const bool cw = b.kw < 3 && tilesW % (2 << b.kw) == 0, ch = b.kh < 3 && tilesH % (2 << b.kh) == 0, cd = b.kd < 3 && tilesD % (2 << b.kd) == 0;
        if (cd && (!ch || ed <= eh) && (!cw || ed <= ew)) ++b.kd;
        else if (ch && (!cw || eh <= ew)) ++b.kh;
        else if (cw) ++b.kw;
        else break;
        room >>= 1;
    }
    unsigned st = step;
    auto digit = [&st](unsigned radix) { const unsigned d = st % radix; st /= radix; return (int)d; };
    b.s_nt = digit((unsigned)ntiles);
    b.s_lw = digit(1u << b.kw); b.s_lh = digit(1u << b.kh); b.s_ld = digit(1u << b.kd);
    b.s_hw = digit((unsigned)tilesW >> b.kw); b.s_hh = digit((unsigned)tilesH >> b.kh); b.s_hd = digit((unsigned)tilesD >> b.kd);
    b.s_nb = (int)st;
    return b;
}

#ifdef __HIPCC__
// digits of logical index L (once per workgroup)
__device__ __forceinline__ void brick_decode(unsigned L, const BrickStep& s, int ntiles, int tilesW, int tilesH, int tilesD, int& nt, int& tw, int& th, int& td, int& nb) {
    nt = (int)(L % (unsigned)ntiles); L /= (unsigned)ntiles;
    const unsigned lw = L & ((1u << s.kw) - 1u); L >>= s.kw;
    const unsigned lh = L & ((1u << s.kh) - 1u); L >>= s.kh;
    const unsigned ld = L & ((1u << s.kd) - 1u); L >>= s.kd;
    const unsigned Tw = (unsigned)tilesW >> s.kw, Th = (unsigned)tilesH >> s.kh, Td = (unsigned)tilesD >> s.kd;
    const unsigned hw = L % Tw; L /= Tw;
    const unsigned hh = L % Th; L /= Th;
    const unsigned hd = L % Td;
    nb = (int)(L / Td);
    tw = (int)((hw << s.kw) | lw); th = (int)((hh << s.kh) | lh); td = (int)((hd << s.kd) | ld);
}
// cursor += step if go (no branch, no division)
__device__ __forceinline__ void brick_advance(const BrickStep& s, bool go, int ntiles, int tilesW, int tilesH, int tilesD, int& nt, int& tw, int& th, int& td, int& nb) {
    int v = nt + (go ? s.s_nt : 0); int cy = v >= ntiles ? 1 : 0; nt = v - (cy ? ntiles : 0);
    const int bw = 1 << s.kw, bh = 1 << s.kh, bd = 1 << s.kd;
    v = (tw & (bw - 1)) + (go ? s.s_lw : 0) + cy; cy = v >= bw ? 1 : 0; const int lw = v - (cy ? bw : 0);
    v = (th & (bh - 1)) + (go ? s.s_lh : 0) + cy; cy = v >= bh ? 1 : 0; const int lh = v - (cy ? bh : 0);
    v = (td & (bd - 1)) + (go ? s.s_ld : 0) + cy; cy = v >= bd ? 1 : 0; const int ld = v - (cy ? bd : 0);
    const int Tw = tilesW >> s.kw, Th = tilesH >> s.kh, Td = tilesD >> s.kd;
    v = (tw >> s.kw) + (go ? s.s_hw : 0) + cy; cy = v >= Tw ? 1 : 0; tw = ((v - (cy ? Tw : 0)) << s.kw) | lw;
    v = (th >> s.kh) + (go ? s.s_hh : 0) + cy; cy = v >= Th ? 1 : 0; th = ((v - (cy ? Th : 0)) << s.kh) | lh;
    v = (td >> s.kd) + (go ? s.s_hd : 0) + cy; cy = v >= Td ? 1 : 0; td = ((v - (cy ? Td : 0)) << s.kd) | ld;
    nb += (go ? s.s_nb : 0) + cy;
}
#endif
