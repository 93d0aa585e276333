// K1 lab, round 5: sweeps of the window-form kernel (csrc/k1_win.h).  lab_win_launch(variant in 40..79, ..., stats).
//   (4x, the window kernel's gather path alone with 8 / 4-plane chunks, and 55-59 / 67 / 47, several chunks per block, were measured in
//    round 5 -- profiles/r5_k1_window.txt -- and removed with the kernel's MODE 0 / NCK parameters)
//   5x  windows loaded ahead of phase A (MODE 1), 6x the same geometry loaded after the fit test (MODE 2):
//       x0 = (DKB, WP, WR) 4,16,8 / 4,32,8 / 4,48,8     x1 = 8,24,8 / 8,32,8 / 4,64,8     x2 = 4,16,6... WR = 6 needs WBYTES % 1024: 4,16,8 -> 4,32,4?
#pragma once
#include <set>
#include <tuple>
// resident blocks per CU of one instantiation, printed once
template <int C, int DKB, int WP, int WR, int MODE>
static void lab_win_occupancy() {
    using W = rcmvs::K1Win<C, DKB, 2, WP, WR>;
    static bool done = false;
    if (done) return;
    done = true;
    const size_t lds = (size_t)W::LDS_BYTES;
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(rcmvs::warp_variance_win_kernel<C, DKB, 2, WP, WR, MODE>), 256, lds);
    printf("      [k1_win C=%d DKB=%d WP=%d WR=%d MODE=%d: LDS %zu B, %d blocks / CU]\n", C, DKB, WP, WR, MODE, lds, nb);
}
static int lab_win_launch(int variant, const float* feats, const float* rot, const float* trans, const float* planes, float* var,
                          int B, int V, int C, int D, int h, int w, unsigned* stats, hipStream_t st) {
    using namespace rcmvs;
#define LW(CC, DD, PP, RR, MM) (lab_win_occupancy<CC, DD, PP, RR, MM>(), k1_win_launch_one<CC, DD, 2, PP, RR, MM>(feats, rot, trans, planes, var, B, V, D, h, w, stats, st))
#define LW3(M, D32, P32, R32, D16, P16, R16, D8, P8, R8) (C == 32 ? LW(32, D32, P32, R32, M) : C == 16 ? LW(16, D16, P16, R16, M) : LW(8, D8, P8, R8, M))
    switch (variant) {
        case 50: return LW3(1, 4, 16, 8, 4, 32, 8, 4, 64, 8);
        case 60: return LW3(2, 4, 16, 8, 4, 32, 8, 4, 64, 8);
        case 51: return LW3(1, 8, 24, 8, 8, 32, 8, 4, 64, 8);
        case 61: return LW3(2, 8, 24, 8, 8, 32, 8, 4, 64, 8);
        case 52: return LW3(1, 4, 16, 6, 4, 32, 6, 4, 64, 6);       // WBYTES = 12288 for all three
        case 62: return LW3(2, 4, 16, 6, 4, 32, 6, 4, 64, 6);
        case 53: return LW3(1, 4, 24, 8, 4, 48, 8, 4, 64, 8);       // wider windows
        case 63: return LW3(2, 4, 24, 8, 4, 48, 8, 4, 64, 8);
        // 8x = plane-pipelined gather form (k1_pp.h): 80 = 4 planes per block, 81 = 8, 82 = 16
#define LP(DD) (C == 32 ? k1_pp_launch_one<32, DD, 2>(feats, rot, trans, planes, var, B, V, D, h, w, st) : \
                C == 16 ? k1_pp_launch_one<16, DD, 2>(feats, rot, trans, planes, var, B, V, D, h, w, st) : \
                          k1_pp_launch_one<8, DD, 2>(feats, rot, trans, planes, var, B, V, D, h, w, st))
        case 80: return LP(4);
        case 81: return LP(8);
        case 82: return LP(16);
#undef LP
        default: return fail(-1, "lab_win: unknown variant %d", variant);
    }
#undef LW3
#undef LW
}
