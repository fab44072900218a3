/* tests/emul: hipExtLaunchKernelGGL lives in the stand-in hip_runtime.h (TEST INFRASTRUCTURE ONLY) */
#pragma once
#include <hip/hip_runtime.h>
