// Shared helpers for librcmvs_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/rcmvs.h"

namespace rcmvs {

constexpr int WAVE = 64;

// thread-local last-error text (rcmvs_last_error_string)
char* err_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// check the launch that was just enqueued (no sync): >0 = hipError_t
inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

// A launch whose start / stop timestamps go to two caller-owned events (hipExtLaunchKernelGGL: the dispatch's own timestamps, what rocprofv3 reports as
// the kernel's duration -- event records around a launch add the marker packets' ~2-3 us each); both events NULL = a plain launch.
#define RCMVS_LAUNCH_TIMED(kernel, grid, block, lds, st, ev0, ev1, ...) do { \
        if ((ev0) || (ev1)) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, ev0, ev1, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__); } while (0)

#define RCMVS_REQUIRE(cond, ...) do { if (!(cond)) return ::rcmvs::fail(-1, __VA_ARGS__); } while (0)

__host__ __device__ inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// XCD-aware remap of a 1-D block id (blocks are dispatched round-robin over the 8 XCDs,
// MI355X_MICROARCH "Workgroup dispatch"): gives each XCD a contiguous chunk of the logical
// tile order so neighbouring tiles share that XCD's L2.  Bijective for any grid size.
__device__ inline unsigned xcd_remap(unsigned bid, unsigned nblk) {
    constexpr unsigned NX = 8;
    unsigned xcd = bid % NX, idx = bid / NX;
    unsigned q = nblk / NX, r = nblk % NX;
    unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace rcmvs
