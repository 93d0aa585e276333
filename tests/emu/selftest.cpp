// Self-test kernels of the CPU emulation (tests/test_emu_shim_cpu.py): the semantics of every primitive the product kernels
// rely on, plus a deliberately racy kernel that the schedule-permutation check must expose.  TEST INFRASTRUCTURE ONLY.
#include "hip/hip_runtime.h"

typedef float v4f __attribute__((ext_vector_type(4)));

// out[b] = sum of in[b * 256 .. +256) through shuffles, LDS and one barrier
__global__ void k_block_sum(const float* in, float* out, int racy) {
    __shared__ float red[4];
    float v = in[blockIdx.x * 256 + threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    if (!racy) __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// lane-level primitives: out[8][64]
__global__ void k_wave_ops(int* out) {
    const int l = threadIdx.x;
    out[0 * 64 + l] = __shfl_xor(l, 5);
    out[1 * 64 + l] = __shfl_up(l, 3);
    out[2 * 64 + l] = __shfl_down(l, 7, 16);
    const unsigned long long b = __ballot(l % 3 == 0);
    out[3 * 64 + l] = __popcll(b & ((1ull << l) - 1ull));
    out[4 * 64 + l] = __builtin_amdgcn_readlane(l * 10, 17);
    out[5 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x112, 0xf, 0xf, false);      // row_shr:2
    out[6 * 64 + l] = __syncthreads_count(l < 20);
    out[7 * 64 + l] = __builtin_amdgcn_readfirstlane(l + 100);
}

// threads >= n exit before the collectives: exited lanes must not block or contribute
__global__ void k_partial_wave(int* out, int n) {
    const int l = threadIdx.x;
    if (l >= n) return;
    out[l] = (int)__popcll(__ballot(1));
    out[64 + l] = __shfl_down(l, 1);
    __syncthreads();
    out[128 + l] = 7;
}

// D = A (16x4) B (4x16) + C with the hardware's register layout
__global__ void k_mfma(const float* A, const float* B, const float* C, float* D) {
    const int l = threadIdx.x;
    const float a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    v4f c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * (l / 16) + r) * 16 + l % 16];
    const v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = d[r];
}

// raw buffer loads: in range -> data, out of range -> 0; dynamic LDS; 2-D blocks and grids
__global__ void k_buffer_and_lds(const float* src, int nfloats, float* out) {
    float* dyn = reinterpret_cast<float*>(::shim::dyn_lds());
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, nfloats * 4, 0x00020000);
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, t * 16, 0, 0);
    dyn[t] = __builtin_bit_cast(v4f, v)[0];
    __syncthreads();
    out[(blockIdx.y * gridDim.x + blockIdx.x) * 64 + t] = dyn[63 - t];
}

// more lane primitives: out[10][64]
__global__ void k_wave_ops2(int* out, float* fout) {
    const int l = threadIdx.x;
    out[0 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x103, 0xf, 0xf, false);      // row_shl:3
    out[1 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x124, 0xf, 0xf, false);      // row_ror:4
    out[2 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x140, 0xf, 0xf, false);      // row_mirror
    out[3 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    out[4 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    out[5 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x1b, 0xf, 0xf, false);       // quad_perm [3,2,1,0]
    out[6 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x111, 0xf, 0xf, true);       // row_shr:1 with bound_ctrl: 0 at the row start
    out[7 * 64 + l] = __builtin_amdgcn_ds_bpermute(((l * 7) % 64) * 4, l * 2);
    out[8 * 64 + l] = __shfl(l, 5, 16);
    out[9 * 64 + l] = __all(l < 64) * 2 + __any(l == 99);
    fout[l] = __builtin_amdgcn_readlane(1.5f * l, 3);
}

// direct-to-LDS loads: every lane moves 16 bytes, the rows land contiguously; out-of-range lanes write zeros
__global__ void k_dma(const float* src, int nfloats, float* out) {
    __shared__ __attribute__((aligned(16))) float stage[2][256];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, nfloats * 4, 0x00020000);
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, &stage[wave][0], 16, (wave * 64 + lane) * 16, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int k = 0; k < 4; ++k) out[threadIdx.x * 4 + k] = stage[wave][lane * 4 + k];
}

typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k_mfma32(const float* A, const float* B, const float* C, float* D) {
    const int l = threadIdx.x;
    const float a = A[(l % 32) * 2 + l / 32], b = B[(l / 32) * 32 + l % 32];
    v16f c;
    for (int r = 0; r < 16; ++r) c[r] = C[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + l % 32];
    const v16f d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + l % 32] = d[r];
}

extern "C" {
void st_wave_ops2(int* out, float* fout) { hipLaunchKernelGGL(k_wave_ops2, dim3(1), dim3(64), 0, 0, out, fout); }
void st_dma(const float* src, int nfloats, float* out) { hipLaunchKernelGGL(k_dma, dim3(1), dim3(128), 0, 0, src, nfloats, out); }
void st_mfma32(const float* A, const float* B, const float* C, float* D) { hipLaunchKernelGGL(k_mfma32, dim3(1), dim3(64), 0, 0, A, B, C, D); }
void st_block_sum(const float* in, float* out, int nblk, int racy) { hipLaunchKernelGGL(k_block_sum, dim3(nblk), dim3(256), 0, 0, in, out, racy); }
void st_wave_ops(int* out) { hipLaunchKernelGGL(k_wave_ops, dim3(1), dim3(64), 0, 0, out); }
void st_partial_wave(int* out, int n) { hipLaunchKernelGGL(k_partial_wave, dim3(1), dim3(64), 0, 0, out, n); }
void st_mfma(const float* A, const float* B, const float* C, float* D) { hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, A, B, C, D); }
void st_buffer_and_lds(const float* src, int nfloats, float* out) {
    hipLaunchKernelGGL(k_buffer_and_lds, dim3(2, 3), dim3(16, 4), 64 * sizeof(float), 0, src, nfloats, out);
}
}
