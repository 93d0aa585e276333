// K1 lab, round 5: sweeps of the window-form kernel (csrc/k1_win.h).  lab_win_launch(variant in 40..79, ..., stats).
//   4x  gather path only (MODE 0): 40 = plane chunks 8 / 8 / 4 (C = 32 / 16 / 8), 41 = 4 / 4 / 4
//   5x  windows loaded ahead of phase A (MODE 1), 6x the same geometry loaded after the fit test (MODE 2):
//       x0 = (DKB, WP, WR) 4,16,8 / 4,32,8 / 4,48,8     x1 = 8,24,8 / 8,32,8 / 4,64,8     x2 = 4,16,6... WR = 6 needs WBYTES % 1024: 4,16,8 -> 4,32,4?
#pragma once
#include <set>
#include <tuple>
// resident blocks per CU of one instantiation, printed once
template <int C, int DKB, int WP, int WR, int MODE, int NCK>
static void lab_win_occupancy() {
    using W = rcmvs::K1Win<C, DKB, 2, WP, WR>;
    static bool done = false;
    if (done) return;
    done = true;
    const size_t lds = (MODE == 0) ? (size_t)W::OFF_WIN : (size_t)W::LDS_BYTES;
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(rcmvs::warp_variance_win_kernel<C, DKB, 2, WP, WR, MODE, NCK>), 256, lds);
    printf("      [k1_win C=%d DKB=%d WP=%d WR=%d MODE=%d NCK=%d: LDS %zu B, %d blocks / CU]\n", C, DKB, WP, WR, MODE, NCK, lds, nb);
}
static int lab_win_launch(int variant, const float* feats, const float* rot, const float* trans, const float* planes, float* var,
                          int B, int V, int C, int D, int h, int w, unsigned* stats, hipStream_t st) {
    using namespace rcmvs;
#define LW(CC, DD, PP, RR, MM) (lab_win_occupancy<CC, DD, PP, RR, MM, 1>(), k1_win_launch_one<CC, DD, 2, PP, RR, MM>(feats, rot, trans, planes, var, B, V, D, h, w, stats, st))
#define LW3(M, D32, P32, R32, D16, P16, R16, D8, P8, R8) (C == 32 ? LW(32, D32, P32, R32, M) : C == 16 ? LW(16, D16, P16, R16, M) : LW(8, D8, P8, R8, M))
    switch (variant) {
        case 40: return LW3(0, 8, 16, 8, 8, 32, 8, 4, 64, 8);
        case 41: return LW3(0, 4, 16, 8, 4, 32, 8, 4, 64, 8);
        case 50: return LW3(1, 4, 16, 8, 4, 32, 8, 4, 64, 8);
        case 60: return LW3(2, 4, 16, 8, 4, 32, 8, 4, 64, 8);
        case 51: return LW3(1, 8, 24, 8, 8, 32, 8, 4, 64, 8);
        case 61: return LW3(2, 8, 24, 8, 8, 32, 8, 4, 64, 8);
        case 52: return LW3(1, 4, 16, 6, 4, 32, 6, 4, 64, 6);       // WBYTES = 12288 for all three
        case 62: return LW3(2, 4, 16, 6, 4, 32, 6, 4, 64, 6);
        case 53: return LW3(1, 4, 24, 8, 4, 48, 8, 4, 64, 8);       // wider windows
        case 63: return LW3(2, 4, 24, 8, 4, 48, 8, 4, 64, 8);
        // chunks per block (NCK) on the 50 / 60 / 41 geometry: 55 = 2, 56 = 3, 57 = 4, 58 = 6, 59 = 12; 67 = MODE 2 x 4; 47 = gather x 4
#define LWN(M, N) (C == 32 ? k1_win_launch_one<32, 4, 2, 16, 8, M, N>(feats, rot, trans, planes, var, B, V, D, h, w, stats, st) : \
                   C == 16 ? k1_win_launch_one<16, 4, 2, 32, 8, M, N>(feats, rot, trans, planes, var, B, V, D, h, w, stats, st) : \
                             k1_win_launch_one<8, 4, 2, 64, 8, M, N>(feats, rot, trans, planes, var, B, V, D, h, w, stats, st))
        case 55: return LWN(1, 2);
        case 56: return LWN(1, 3);
        case 57: return LWN(1, 4);
        case 58: return LWN(1, 6);
        case 59: return LWN(1, 12);
        case 67: return LWN(2, 4);
        case 47: return LWN(0, 4);
#undef LWN
        default: return fail(-1, "lab_win: unknown variant %d", variant);
    }
#undef LW3
#undef LW
}
