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

// ---- several 2-D copies in one launch: rows of `row_bytes` contiguous bytes, `*_stride` bytes between consecutive rows --------------------
// Block placements of filters into merged filters (block-diagonal, concatenated or channel-permuted: liso/slim/model/update.py's and
// center_head.py's parallel convolutions run as one) and the way back for their gradients: a filter block [co_k, ci_k, kh, kw] inside
// [CO, CI, kh, kw] is co_k rows of ci_k * kh * kw contiguous floats.  One launch instead of one framework copy per block.
namespace {

struct RowsTable {
    unsigned char* dst[LISO_MULTI_COPY_ROWS_MAX];
    const unsigned char* src[LISO_MULTI_COPY_ROWS_MAX];
    unsigned rows[LISO_MULTI_COPY_ROWS_MAX], row_bytes[LISO_MULTI_COPY_ROWS_MAX];
    unsigned long long dst_stride[LISO_MULTI_COPY_ROWS_MAX], src_stride[LISO_MULTI_COPY_ROWS_MAX];
};

__global__ __launch_bounds__(256) void multi_copy_rows_kernel(const RowsTable t) {
    const int s = blockIdx.y;
    unsigned char* __restrict__ d = t.dst[s];
    const unsigned char* __restrict__ a = t.src[s];
    const unsigned rows = t.rows[s], rb = t.row_bytes[s];
    const unsigned long long ds = t.dst_stride[s], ss = t.src_stride[s];
    // 4-byte words (every user copies fp32 blocks); a thread per word
    const unsigned words = rb >> 2;
    const unsigned long long total = (unsigned long long)rows * words;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long r = i / words;
        const unsigned w = (unsigned)(i - r * words);
        reinterpret_cast<unsigned*>(d + r * ds)[w] = reinterpret_cast<const unsigned*>(a + r * ss)[w];
    }
}

}  // namespace

extern "C" int liso_multi_copy_rows(int n, void* const* dst, const void* const* src, const unsigned* rows, const unsigned* row_bytes,
                                    const size_t* dst_stride, const size_t* src_stride, void* stream) {
    if (n < 0 || (n > 0 && (!dst || !src || !rows || !row_bytes || !dst_stride || !src_stride))) return LISO_EINVAL;
    for (int base = 0; base < n; base += LISO_MULTI_COPY_ROWS_MAX) {
        RowsTable t = {};
        const int m = n - base < LISO_MULTI_COPY_ROWS_MAX ? n - base : LISO_MULTI_COPY_ROWS_MAX;
        unsigned long long most = 0;
        int used = 0;
        for (int i = 0; i < m; i++) {
            const int k = base + i;
            if (rows[k] == 0 || row_bytes[k] == 0) continue;
            if (!dst[k] || !src[k] || (row_bytes[k] & 3) || (dst_stride[k] & 3) || (src_stride[k] & 3) ||
                ((((uintptr_t)dst[k]) | ((uintptr_t)src[k])) & 3))
                return LISO_EINVAL;
            t.dst[used] = (unsigned char*)dst[k];
            t.src[used] = (const unsigned char*)src[k];
            t.rows[used] = rows[k];
            t.row_bytes[used] = row_bytes[k];
            t.dst_stride[used] = dst_stride[k];
            t.src_stride[used] = src_stride[k];
            const unsigned long long words = (unsigned long long)rows[k] * (row_bytes[k] >> 2);
            most = words > most ? words : most;
            used++;
        }
        if (!used) continue;
        unsigned long long bx = (most + 255) / 256;
        bx = bx < 1 ? 1 : bx > 512 ? 512 : bx;
        multi_copy_rows_kernel<<<dim3((unsigned)bx, (unsigned)used), 256, 0, (hipStream_t)stream>>>(t);
    }
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

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
