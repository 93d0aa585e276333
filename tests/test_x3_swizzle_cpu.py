"""The LDS bank swizzles of csrc/conv3d_x3.hip, checked exhaustively on the CPU.

ds_read_b128 is served in four 16-lane groups (MI355X_MICROARCH, "LDS": {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32)
and a 16-byte access occupies slot (address / 16) mod 16 of the 64 banks.  A B-fragment read of the kernel has lane (n = lane & 15,
kk = lane >> 4) at byte  hc * VB + ((ci0 * 2) ^ swz(hc))  with hc = tap column + n: the test restates x3_swz() and asserts that
every lane group hits 16 distinct slots for every tap column (and both channel halves at Cin = 64) -- and that without the swizzle
it does not (the 43 % conflict share measured before, profiles/r2_x3_pmc_summary.txt)."""
import pytest

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def swz(cin, hc):                       # x3_swz<C, CIN, KIND> for the stride-1 / planar / transposed kinds
    if cin == 32:
        return ((hc >> 2) & 1) * 32
    if cin == 64:
        return ((0xCB5888 >> (3 * (hc >> 1))) & 7) * 16
    return 0


def conflicts(cin, tap_cols, swizzled):
    vb = cin * 2                        # bytes per voxel per piece plane
    worst = 1
    for q in range(tap_cols):
        for half in range(max(1, cin // 32)):
            for g in GROUPS:
                slots = {}
                for lane in g:
                    n, kk = lane & 15, lane >> 4
                    hc = q + n
                    inner = half * 64 + kk * 16            # ci0 * 2: channel block of this lane inside the voxel
                    addr = hc * vb + (inner ^ (swz(cin, hc) if swizzled else 0))
                    s = (addr // 16) % 16
                    slots[s] = slots.get(s, 0) + 1
                worst = max(worst, max(slots.values()))
    return worst


@pytest.mark.parametrize("cin,tap_cols", [(32, 3), (32, 2), (64, 3)])
def test_swizzle_makes_every_lane_group_conflict_free(cin, tap_cols):
    assert conflicts(cin, tap_cols, swizzled=True) == 1
    assert conflicts(cin, tap_cols, swizzled=False) >= 2


def test_swizzle_stays_inside_the_voxel_and_is_its_own_inverse():
    for cin in (32, 64):
        for hc in range(18):
            m = swz(cin, hc)
            assert m % 16 == 0 and m < cin * 2                      # whole 16-byte slots, inside the voxel's bytes
            for off in range(0, cin * 2, 8):                          # the producer's 8-byte stores
                assert ((off ^ m) ^ m) == off and 0 <= (off ^ m) < cin * 2


def test_conv_roofline_object_of_the_bench():
    """bench.conv_roofline: algorithmic flops of the split-bf16 launches over their event times."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    class Ev:
        def __init__(self, ms): self.ms = ms
        def elapsed_time(self, other): return other.ms

    ev = [(Ev(0), Ev(0.1), ("s1", 1, 32, 256, 320, 16, 8)),          # 18.1 GFLOP in 0.1 ms
          (Ev(0), Ev(0.05), ("s1", 1, 8, 64, 80, 64, 64)),           # not an x3 layer: ignored
          (Ev(0), Ev(0.01), ("s1", 3, 1, 128, 160, 32, 32))]         # planar: 9 taps
    r = bench.conv_roofline(ev, 1)
    flops = 2 * 27 * 16 * 8 * 32 * 256 * 320 + 2 * 9 * 32 * 32 * 3 * 128 * 160
    assert abs(r["achieved"] - flops / 0.11e-3 / 1e12) < 0.06 and r["bound"] == "mfma" and r["peak"] == bench.FP32_PEAK_TFLOPS
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and abs(r["us_per_scene"] - 110.0) < 1e-6
    assert bench.conv_roofline([], 1) is None
