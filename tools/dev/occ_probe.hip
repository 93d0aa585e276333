// Developer probe: resident blocks per CU of the marching prob conv as the runtime sees it (LDS + register limits)
#include "../../rc_mvsnet_amd/csrc/conv3d_lds.hip"
namespace rcmvs { char* err_buf() { static char b[512]; return b; }
int fail(int code, const char* fmt, ...) { (void)fmt; return code; } }
int main() {
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)rcmvs::prob_conv_march_kernel, 256, 0);
    printf("prob_conv_march_kernel: %d resident blocks per CU (%s)\n", n, hipGetErrorString(e));
    hipFuncAttributes a;
    e = hipFuncGetAttributes(&a, (const void*)rcmvs::prob_conv_march_kernel);
    printf("  numRegs %d, sharedSizeBytes %zu, maxThreadsPerBlock %d\n", a.numRegs, a.sharedSizeBytes, a.maxThreadsPerBlock);
    return 0;
}
