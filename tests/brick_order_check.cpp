// CPU check of elektronn3_amd/csrc/brick_order.h (built and run by tests/test_brick_order.py): the device functions compiled as host code.
//   brick_order_check ntiles tilesW tilesH tilesD N step concurrent w_run  ->  "ok kw kh kd" or a message and exit code 1
#define __HIPCC__ 1
#define __device__
#define __forceinline__ inline
#include "../elektronn3_amd/csrc/brick_order.h"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    if (argc != 9) return 2;
    const int ntiles = atoi(argv[1]), tW = atoi(argv[2]), tH = atoi(argv[3]), tD = atoi(argv[4]), N = atoi(argv[5]);
    const unsigned step = (unsigned)atoi(argv[6]), conc = (unsigned)atoi(argv[7]);
    const BrickStep b = brick_step_make(step, conc, ntiles, tW, tH, tD, atoi(argv[8]));
    if (tW % (1 << b.kw) || tH % (1 << b.kh) || tD % (1 << b.kd)) { printf("block does not divide the grid\n"); return 1; }
    const unsigned total = (unsigned)ntiles * tW * tH * tD * N;
    // decode is a bijection onto the grid
    std::vector<char> seen(total, 0);
    for (unsigned L = 0; L < total; ++L) {
        int nt, tw, th, td, nb;
        brick_decode(L, b, ntiles, tW, tH, tD, nt, tw, th, td, nb);
        if (nt < 0 || nt >= ntiles || tw < 0 || tw >= tW || th < 0 || th >= tH || td < 0 || td >= tD || nb < 0 || nb >= N) { printf("L %u decodes outside the grid\n", L); return 1; }
        const unsigned flat = (((unsigned)nb * tD + td) * tH + th) * tW * ntiles + (unsigned)tw * ntiles + nt;
        if (seen[flat]) { printf("L %u decodes to a brick seen before\n", L); return 1; }
        seen[flat] = 1;
    }
    // a cursor advanced k times from L0 is decode(L0 + k * step); go = false leaves it alone
    for (unsigned L0 = 0; L0 < total && L0 < 97; L0 += 3) {
        int nt, tw, th, td, nb;
        brick_decode(L0, b, ntiles, tW, tH, tD, nt, tw, th, td, nb);
        for (unsigned L = L0 + step; L < total && step; L += step) {
            brick_advance(b, false, ntiles, tW, tH, tD, nt, tw, th, td, nb);
            brick_advance(b, true, ntiles, tW, tH, tD, nt, tw, th, td, nb);
            int n2, w2, h2, d2, b2;
            brick_decode(L, b, ntiles, tW, tH, tD, n2, w2, h2, d2, b2);
            if (nt != n2 || tw != w2 || th != h2 || td != d2 || nb != b2) { printf("cursor from %u differs from decode(%u)\n", L0, L); return 1; }
        }
    }
    // the concurrent set (conc consecutive indices from a multiple of conc) is one block of bricks times the column tiles
    if (conc && total % conc == 0 && (unsigned)ntiles << (b.kw + b.kh + b.kd) == conc) {
        for (unsigned L0 = 0; L0 < total; L0 += conc) {
            int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
            for (unsigned L = L0; L < L0 + conc; ++L) {
                int nt, c[3], nb;
                brick_decode(L, b, ntiles, tW, tH, tD, nt, c[0], c[1], c[2], nb);
                for (int i = 0; i < 3; ++i) { lo[i] = c[i] < lo[i] ? c[i] : lo[i]; hi[i] = c[i] > hi[i] ? c[i] : hi[i]; }
            }
            if (hi[0] - lo[0] + 1 != 1 << b.kw || hi[1] - lo[1] + 1 != 1 << b.kh || hi[2] - lo[2] + 1 != 1 << b.kd) { printf("set at %u is not one block\n", L0); return 1; }
        }
    }
    printf("ok %d %d %d\n", b.kw, b.kh, b.kd);
    return 0;
}
