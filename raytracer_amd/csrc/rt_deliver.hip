// rt_deliver.hip -- finished uint8 rows from device memory into pinned host memory (the array the caller of the
// reference's render() converts and saves, ray_tracer.py:436-439), as a kernel with a SMALL footprint.
//
// Why not hipMemcpyAsync: on this platform the runtime performs a device-to-host copy with a blit kernel whose grid
// covers the whole buffer; its waves sit on every CU waiting for PCIe (0.45 ms for a 4K frame) and the kernels of the
// next frame queue up behind them.  PCIe needs only ~100 KB in flight, so a few dozen workgroups that loop over the
// rows reach the same rate, and with at most 8 VGPRs a wave of this kernel fits beside the register-heavy persistent
// waves of the shading kernel (3 x 168 of a SIMD's 512 VGPRs) instead of waiting for one of them to finish.
//
// Optionally gathers rows on the way (dst row y = src row src_row[y]): rank 0 of a multi-GPU job assembles the frame
// from the rank-major stack of gathered tiles and sends it to the host in one pass.
#include "rt_device.h"

#include <algorithm>

namespace {

// one workgroup per row at a time; the row's base addresses are wave-uniform (scalar registers), the lanes supply a
// 32-bit element offset
template <typename T>
__global__ __launch_bounds__(256) void k_deliver(T* __restrict__ dst, const T* __restrict__ src, const int32_t* __restrict__ src_row,
                                                 uint32_t rows, uint32_t row_elems) {
    for (uint32_t y = blockIdx.x; y < rows; y += gridDim.x) {
        const uint32_t sy = src_row ? (uint32_t)__builtin_amdgcn_readfirstlane(src_row[y]) : y;
        const T* __restrict__ s = src + (size_t)sy * row_elems;
        T* __restrict__ d = dst + (size_t)y * row_elems;
        for (uint32_t x = threadIdx.x; x < row_elems; x += 256) d[x] = s[x];
    }
}

// The 16-byte form, written out so that it stays within 8 VGPRs: row bases in scalar registers, one 32-bit byte offset
// per lane (the compiler's own addressing keeps two 64-bit addresses per lane: 10 VGPRs, one granule too many).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_deliver16(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, const int32_t* __restrict__ src_row,
                                                   uint32_t rows, uint32_t row_bytes) {
    for (uint32_t y = blockIdx.x; y < rows; y += gridDim.x) {
        const uint32_t sy = src_row ? (uint32_t)__builtin_amdgcn_readfirstlane(src_row[y]) : y;
        const uint8_t* s = src + (size_t)sy * row_bytes;
        uint8_t* d = dst + (size_t)y * row_bytes;
        for (uint32_t off = threadIdx.x * 16u; off < row_bytes; off += 256u * 16u) {
            u32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(s) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(off), "v"(v), "s"(d) : "memory");
        }
    }
}

}  // namespace

void rt_launch_deliver(void* dst, const void* src, const int32_t* src_row, uint32_t rows, size_t row_bytes, int workgroups, hipStream_t s) {
    if (!rows || !row_bytes) return;
    const dim3 g((unsigned)std::max(1, std::min<int>(workgroups, (int)rows))), b(256);
    const bool a16 = row_bytes % 16 == 0 && ((uintptr_t)dst | (uintptr_t)src) % 16 == 0;
    const bool a4 = row_bytes % 4 == 0 && ((uintptr_t)dst | (uintptr_t)src) % 4 == 0;
    if (a16 && row_bytes < (size_t(1) << 31)) hipLaunchKernelGGL(k_deliver16, g, b, 0, s, (uint8_t*)dst, (const uint8_t*)src, src_row, rows, (uint32_t)row_bytes);
    else if (a4) hipLaunchKernelGGL(k_deliver<uint32_t>, g, b, 0, s, (uint32_t*)dst, (const uint32_t*)src, src_row, rows, (uint32_t)(row_bytes / 4));
    else hipLaunchKernelGGL(k_deliver<uint8_t>, g, b, 0, s, (uint8_t*)dst, (const uint8_t*)src, src_row, rows, (uint32_t)row_bytes);
}
