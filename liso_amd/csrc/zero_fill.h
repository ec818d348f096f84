// Zero-fill as an ordinary kernel launch.
//
// The library does not use hipMemsetAsync: inside a captured hipGraph its memset nodes are not reliably ordered against the
// neighbouring kernel nodes when the launch stream carries a backlog of eager work (ROCm 7.2, MI355X; measured with
// scripts/debug_pillar_graph_fault.py: a graph holding memset -> count(atomicAdd) -> scan -> fill(atomicAdd cursor) replays
// cleanly on an idle stream and ends in a GPU memory fault once a few thousand eager launches sit between replays -- the
// counters are then not zero when the kernels run and the scattered writes leave their buffer; the same graph with this
// kernel in place of the memset survives).  A kernel node is ordered like every other node.
#ifndef LISO_ZERO_FILL_H
#define LISO_ZERO_FILL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace liso_zero {
namespace {  // (one private copy per translation unit)

__global__ __launch_bounds__(256) void zero_fill_kernel(unsigned char* __restrict__ p, size_t bytes) {
    // head up to the first 16-byte boundary and tail after the last one: bytewise by block 0; the middle: uint4 stores
    const uintptr_t a = (uintptr_t)p;
    size_t head = (16 - (a & 15)) & 15;
    if (head > bytes) head = bytes;
    const size_t mid16 = (bytes - head) / 16;
    const size_t tail_begin = head + mid16 * 16;
    uint4* q = reinterpret_cast<uint4*>(p + head);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < mid16; i += (size_t)gridDim.x * blockDim.x) q[i] = z;
    if (blockIdx.x == 0) {
        for (size_t i = threadIdx.x; i < head; i += blockDim.x) p[i] = 0;
        for (size_t i = tail_begin + threadIdx.x; i < bytes; i += blockDim.x) p[i] = 0;
    }
}

// enqueues the fill on `st`; returns hipSuccess or the launch error
static inline hipError_t zero_async(void* ptr, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    size_t blocks = (bytes / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char*)ptr, bytes);
    return hipGetLastError();
}

}  // namespace
}  // namespace liso_zero
#endif
