// stand-in for <hip/hip_ext.h> on the CPU emulation: hipExtLaunchKernelGGL is defined by hip_runtime.h (events ignored)
#pragma once
#include "hip_runtime.h"
