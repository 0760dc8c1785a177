// fill.cuh — device memory cleared by a kernel of this library (shared by devops.cuh and msm.hip)
#pragma once
#include <stdint.h>
#include <algorithm>
#include <hip/hip_runtime.h>
#include "ff.cuh"

namespace swm {

// Zero fill as a kernel of this library instead of hipMemsetAsync: the runtime's fill kernel has no issue priority and is starved
// by an accumulation in flight like every other light kernel was (r05 timeline: 1.1 + 0.7 ms for the two 32-MB clears of round 1
// at 2^20, 10 us each alone) — this one carries SWM_LIGHT_KERNEL.  `bytes` is a multiple of 4 (every caller clears whole words).
static __global__ void __launch_bounds__(256) zero_fill_kernel(uint32_t* __restrict__ p, size_t words) {
    SWM_LIGHT_KERNEL();
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < words; i += stride) {
        if (i + 4 <= words && ((uintptr_t)(p + i) & 15) == 0) {
            *reinterpret_cast<uint4*>(p + i) = make_uint4(0u, 0u, 0u, 0u);
        } else {
            for (size_t k = i; k < words && k < i + 4; k++) p[k] = 0u;
        }
    }
}
inline hipError_t zero_fill_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (bytes % 4 != 0 || ((uintptr_t)p & 3) != 0) return hipMemsetAsync(p, 0, bytes, st);
    const size_t words = bytes / 4;
    const unsigned grid = (unsigned)std::min<size_t>((words / 4 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), words);
    return hipGetLastError();
}

}  // namespace swm
