// Several device-to-device copies in ONE launch for gfx950 (MI355X).  C ABI: include/liso_optim.h.
//
// The training steps stage their inputs into the static buffers of a captured hipGraph tensor by tensor (10 target tensors per detector
// step, 20-30 cloud tensors per SLIM inference replay): each a runtime buffer copy of a few KB that costs a launch (~4 us of stream
// time, more of host time).  Here the (dst, src, bytes) triples travel as kernel arguments and blockIdx.y picks the segment.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_optim.h"

namespace {

struct CopyTable {
    unsigned char* dst[LISO_MULTI_COPY_MAX];
    const unsigned char* src[LISO_MULTI_COPY_MAX];
    unsigned long long bytes[LISO_MULTI_COPY_MAX];
};

__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyTable t) {
    const int s = blockIdx.y;
    unsigned char* __restrict__ d = t.dst[s];
    const unsigned char* __restrict__ a = t.src[s];
    const unsigned long long n = t.bytes[s];
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)d | (uintptr_t)a) & 15) == 0) {
        const unsigned long long n16 = n >> 4;
        for (unsigned long long k = i; k < n16; k += stride) reinterpret_cast<uint4*>(d)[k] = reinterpret_cast<const uint4*>(a)[k];
        for (unsigned long long k = (n16 << 4) + i; k < n; k += stride) d[k] = a[k];
    } else {
        for (unsigned long long k = i; k < n; k += stride) d[k] = a[k];
    }
}

}  // namespace

extern "C" int liso_multi_copy(int n, void* const* dst, const void* const* src, const size_t* bytes, void* stream) {
    if (n < 0 || (n > 0 && (!dst || !src || !bytes))) return LISO_EINVAL;
    for (int base = 0; base < n; base += LISO_MULTI_COPY_MAX) {
        CopyTable t = {};
        const int m = n - base < LISO_MULTI_COPY_MAX ? n - base : LISO_MULTI_COPY_MAX;
        size_t most = 0;
        int used = 0;
        for (int i = 0; i < m; i++) {
            if (bytes[base + i] == 0) continue;
            if (!dst[base + i] || !src[base + i]) return LISO_EINVAL;
            t.dst[used] = (unsigned char*)dst[base + i];
            t.src[used] = (const unsigned char*)src[base + i];
            t.bytes[used] = bytes[base + i];
            most = bytes[base + i] > most ? bytes[base + i] : most;
            used++;
        }
        if (!used) continue;
        size_t bx = (most / 16 + 255) / 256;  // one 16-byte chunk per thread of the largest segment, at most 1024 blocks per segment
        bx = bx < 1 ? 1 : bx > 1024 ? 1024 : bx;
        multi_copy_kernel<<<dim3((unsigned)bx, (unsigned)used), 256, 0, (hipStream_t)stream>>>(t);
    }
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}
