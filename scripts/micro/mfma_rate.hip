// Bare MFMA issue-rate probe for gfx950 (round-5 VERDICT item 1d: "re-measure the bare MFMA issue rate against the guide's 32 cycles").
// One wave per SIMD (256-thread blocks, one block per CU) or two (512-thread blocks); operands in registers; NACC independent
// accumulators; every lane stamps s_memtime / s_memrealtime around the loop: cycles per v_mfma_f32_32x32x16_bf16 and the shader clock
// held during the loop.  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/mfma_rate scripts/micro/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC, int CHAIN>
__global__ __launch_bounds__(512) void mfma_loop(const unsigned* __restrict__ seed, int iters, float* __restrict__ sink,
                                                 unsigned long long* __restrict__ stamps) {
    const int tid = threadIdx.x;
    uint4 av = reinterpret_cast<const uint4*>(seed)[(blockIdx.x * blockDim.x + tid) % 4096];
    uint4 bv = reinterpret_cast<const uint4*>(seed)[(blockIdx.x * blockDim.x + tid + 77) % 4096];
    bf8 a = __builtin_bit_cast(bf8, av), b = __builtin_bit_cast(bf8, bv);
    f16v acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++)
#pragma unroll
            for (int c = 0; c < CHAIN; c++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) s += acc[i][e];
    if (s == 123.456f) sink[0] = s;
    if ((tid & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int NACC, int CHAIN>
void run(const char* name, int blocks, int threads, int iters, const unsigned* seed, float* sink, unsigned long long* stamps, int reps) {
    const int waves = blocks * threads / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) mfma_loop<NACC, CHAIN><<<blocks, threads>>>(seed, iters, sink, stamps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) mfma_loop<NACC, CHAIN><<<blocks, threads>>>(seed, iters, sink, stamps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), clk(waves);
    const double n_mfma = (double)iters * NACC * CHAIN;
    for (int w = 0; w < waves; w++) {
        cyc[w] = (double)h[2 * w] / n_mfma;
        clk[w] = (double)h[2 * w] / ((double)h[2 * w + 1] * 10.0);  // s_memrealtime ticks at 100 MHz -> GHz
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double us = ms * 1e3 / reps;
    const double flops = (double)waves * n_mfma * 32768.0;
    printf("%-34s blocks %4d x %3d thr iters %6d: %8.1f us/launch  %7.1f TFLOP/s  cyc/MFMA median %.2f (min %.2f max %.2f)  clock median %.3f GHz\n",
           name, blocks, threads, iters, us, flops / (us * 1e-6) / 1e12, cyc[waves / 2], cyc[0], cyc[waves - 1], clk[waves / 2]);
}

int main(int argc, char** argv) {
    const int random = argc > 1 ? atoi(argv[1]) : 1;
    unsigned* seed;
    float* sink;
    unsigned long long* stamps;
    hipMalloc(&seed, 4096 * 16);
    hipMalloc(&sink, 64);
    hipMalloc(&stamps, 2 * 8192 * 8);
    std::vector<unsigned> hs(4096 * 4);
    for (size_t i = 0; i < hs.size(); i++) {
        // bf16 pairs in [-2, 2): random mantissas, exponents around 0
        unsigned lo = 0x3f00u + (rand() & 0xff) + ((rand() & 1) << 15), hi = 0x3f00u + (rand() & 0xff) + ((rand() & 1) << 15);
        hs[i] = random ? (lo | (hi << 16)) : 0u;
    }
    hipMemcpy(seed, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    printf("operands: %s\n", random ? "random bf16" : "zeros");
    // long launches (clock settled) and ~70-us launches (the size of one convolution layer)
    for (int iters : {200000, 2700 / 6}) {
        const int reps = iters > 10000 ? 3 : 200;
        run<1, 6>("1 acc, chain 6 (1 wave/SIMD)", 256, 256, iters, seed, sink, stamps, reps);
        run<2, 3>("2 acc x3 (F32X3 pattern, 1 w/SIMD)", 256, 256, iters, seed, sink, stamps, reps);
        run<6, 1>("6 independent acc (1 wave/SIMD)", 256, 256, iters, seed, sink, stamps, reps);
        run<2, 3>("2 acc x3, 2 waves/SIMD", 256, 512, iters / 2, seed, sink, stamps, reps);
        run<2, 3>("2 acc x3, 192 CUs", 192, 256, iters, seed, sink, stamps, reps);
        run<2, 3>("2 acc x3, 2 blocks/CU", 512, 256, iters / 2, seed, sink, stamps, reps);
    }
    return 0;
}
