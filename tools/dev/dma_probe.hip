// tools/dev/dma_probe.hip -- what `buffer_load_dwordx4 ... offen lds` (direct-to-LDS DMA) does on gfx950, checked on the hardware
// before the split-bf16 conv kernel relies on it (csrc/lds_dma.h):
//   T1  lane i writes 16 B at M0 + 16 i, whatever its source offset (a per-lane source permutation = an LDS swizzle)
//   T2  a lane whose voffset is out of range (>= num_records) writes ZEROS (the conv's zero padding)
//   T3  a lane that is masked off in EXEC writes nothing
//   T4  the SGPR offset is outside the bounds check (voffset in range + a large soffset still loads)
//   T5  LDS destinations above 64 KB work (M0 carries more than 16 bits)
//   T6  streaming rate: every CU pulls a large buffer through an LDS ring with 1, 2 or 4 waves issuing, 1 KiB per instruction
// build: hipcc --offload-arch=gfx950 -O3 tools/dev/dma_probe.hip -o tools/dev/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst, int voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// one wave; LDS prefilled with 0xAB; mode selects the test; the first 1 KiB at lds_base is copied out
__global__ void semantics_kernel(const unsigned* src, unsigned* out, int num_records, int mode, unsigned lds_base) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    unsigned* l32 = reinterpret_cast<unsigned*>(smem + lds_base);
    for (int i = lane; i < 512; i += 64) l32[i] = 0xABABABABu;      // 2 KiB: the 1 KiB target and the KiB after it
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), (short)0, num_records, 0x00020000);
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + lds_base);
    int voff = ((lane * 37) & 63) * 16;          // a permutation of the 64 16-byte units
    int soff = 0;
    if (mode == 2 && (lane & 3) == 1) voff = 0x7ffffff0;             // T2: every fourth lane out of range
    if (mode == 4) soff = num_records;                                // T4: the data sits one buffer length further on
    if (mode == 3) { if ((lane & 3) != 2) dma16(rsrc, dst, voff, soff); }      // T3: lanes 2 mod 4 masked off
    else dma16(rsrc, dst, voff, soff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = l32[i];
}

// T6: persistent streaming.  grid = CUs, block = 256; `nw` of the 4 waves issue; each issuing wave walks its share of the block's
// contiguous chunk with `depth` instructions in flight (counted vmcnt), ring of 64 KiB per block.
template <int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned char* src, long long bytes_per_block, int nw, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nw) return;
    const unsigned char* base = src + (long long)blockIdx.x * bytes_per_block;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), (short)0, (int)bytes_per_block, 0x00020000);
    const unsigned ring = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem) + wave * 16384;
    const long long per_wave = bytes_per_block / nw;
    const int n = (int)(per_wave / 1024);
    const int w0 = (int)(wave * per_wave);
    for (int i = 0; i < n; ++i) {
        dma16(rsrc, ring + (i & 15) * 1024, lane * 16, w0 + i * 1024);
        if (i >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(DEPTH) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && smem[wave * 16384] == 0x5a && sink) sink[0] = 1;
}

static bool run_sem(const char* name, int mode, unsigned lds_base, const unsigned* dsrc, const std::vector<unsigned>& hsrc, int num_records) {
    unsigned* dout;
    CK(hipMalloc(&dout, 2048));
    const size_t lds = lds_base + 2048;
    CK(hipFuncSetAttribute((const void*)semantics_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(semantics_kernel, dim3(1), dim3(64), lds, 0, dsrc, dout, num_records, mode, lds_base);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(512);
    CK(hipMemcpy(h.data(), dout, 2048, hipMemcpyDeviceToHost));
    CK(hipFree(dout));
    int bad = 0, zeros = 0, untouched = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int unit = (lane * 37) & 63;
        for (int j = 0; j < 4; ++j) {
            const unsigned got = h[lane * 4 + j];
            unsigned want = hsrc[(mode == 4 ? num_records / 4 : 0) + unit * 4 + j];
            if (mode == 2 && (lane & 3) == 1) { if (got == 0u) ++zeros; else if (got == 0xABABABABu) ++untouched; want = got; }
            if (mode == 3 && (lane & 3) == 2) { if (got == 0xABABABABu) ++untouched; else if (got == 0u) ++zeros; want = got; }
            if (got != want) ++bad;
        }
    }
    int spill = 0;
    for (int i = 256; i < 512; ++i) if (h[i] != 0xABABABABu) ++spill;
    printf("%-44s mismatches %d  special lanes: zeros %d untouched %d (of 64 dwords)  bytes past the KiB touched %d\n", name, bad, zeros, untouched, spill);
    return bad == 0;
}

int main() {
    const int num_records = 4096;
    std::vector<unsigned> hsrc(2 * num_records / 4);
    for (size_t i = 0; i < hsrc.size(); ++i) hsrc[i] = 0x10000000u + (unsigned)i;
    unsigned* dsrc;
    CK(hipMalloc(&dsrc, hsrc.size() * 4));
    CK(hipMemcpy(dsrc, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice));
    run_sem("T1 permuted source, lane-linear destination", 1, 0, dsrc, hsrc, num_records);
    run_sem("T2 out-of-range lanes (1 mod 4)", 2, 0, dsrc, hsrc, num_records);
    run_sem("T3 EXEC-masked lanes (2 mod 4)", 3, 0, dsrc, hsrc, num_records);
    run_sem("T4 soffset beyond num_records", 4, 0, dsrc, hsrc, num_records);
    run_sem("T5 destination at 100 KiB", 1, 100 * 1024, dsrc, hsrc, num_records);
    run_sem("T5 destination at 150 KiB, out-of-range lanes", 2, 150 * 1024, dsrc, hsrc, num_records);

    // T6
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const long long per_block = 8LL << 20;           // 8 MiB per CU -> 2 GiB total: beyond the Infinity Cache
    unsigned char* big;
    CK(hipMalloc(&big, per_block * ncu));
    CK(hipMemset(big, 1, per_block * ncu));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define STREAM(DEPTH) for (int nw = 1; nw <= 4; nw *= 2) { \
        CK(hipFuncSetAttribute((const void*)stream_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); \
        hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(ncu), dim3(256), 65536, 0, big, per_block, nw, (unsigned*)nullptr); \
        CK(hipEventRecord(e0)); \
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(ncu), dim3(256), 65536, 0, big, per_block, nw, (unsigned*)nullptr); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
        printf("T6 stream: %d CUs, %d issuing wave(s), depth %2d: %.1f us per pass, %.2f TB/s\n", ncu, nw, DEPTH, ms * 1e3 / 3, per_block * ncu * 3 / (ms * 1e-3) / 1e12); }
    STREAM(4) STREAM(8) STREAM(14)
    return 0;
}
