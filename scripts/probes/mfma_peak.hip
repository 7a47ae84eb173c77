// MFMA-only peak on the live device ("Peak denominators"): independent accumulate chains of
// v_mfma_f32_32x32x16_bf16, v_mfma_f32_32x32x64_f8f6f4 (fp8 e4m3) and v_mfma_f32_32x32x2_f32 -- no memory traffic in the
// loop -- on every SIMD (2 waves each), once with zero operands and once with pseudo-random ones (the chip is
// power-limited: the sustained rate depends on the data).  Prints one JSON line with the device properties and rates.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/probes/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int MODE> __global__ __launch_bounds__(512, 2) void peak(float *out, int iters, unsigned seed) {
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    unsigned x = seed ? (threadIdx.x * 2654435761u + blockIdx.x * 40503u + seed) : 0u;
    auto next = [&]() { x = seed ? x * 1664525u + 1013904223u : 0u; return x; };
    if (MODE == 0) {
        bf16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = (__bf16)(seed ? ((int)(next() >> 24) - 128) / 64.0f : 0.0f);
            b[j] = (__bf16)(seed ? ((int)(next() >> 24) - 128) / 64.0f : 0.0f);
        }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    } else if (MODE == 1) {
        i32x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = (int)(next() & 0x3f3f3f3fu);  // e4m3 bytes with small exponents (no NaN pattern)
            b[j] = (int)(next() & 0x3f3f3f3fu);
        }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[k], 0, 0, 0, 0, 0, 0);
    } else if (MODE == 3) {  // bf16 on the other matrix shape, 16x16x32 (four accumulators of 4 registers)
        typedef __attribute__((ext_vector_type(4))) float f32x4;
        bf16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = (__bf16)(seed ? ((int)(next() >> 24) - 128) / 64.0f : 0.0f);
            b[j] = (__bf16)(seed ? ((int)(next() >> 24) - 128) / 64.0f : 0.0f);
        }
        f32x4 c4[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c4[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k & 3][k >> 2] += c4[k][0] + c4[k][3];
    } else if (MODE == 4) {  // f16 32x32x16
        typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
        f16x8 a, b;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = (_Float16)(seed ? ((int)(next() >> 24) - 128) / 64.0f : 0.0f);
            b[j] = (_Float16)(seed ? ((int)(next() >> 24) - 128) / 64.0f : 0.0f);
        }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
    } else {
        float a = seed ? ((int)(next() >> 24) - 128) / 64.0f : 0.0f, b = seed ? ((int)(next() >> 24) - 128) / 64.0f : 0.0f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE> double run(int cus, unsigned seed, double flop_per_mfma, int per_iter = 4) {
    float *out;
    (void)hipMalloc(&out, 4);
    const int iters = 20000, wgs = cus;  // one 8-wave workgroup per CU = 2 waves per SIMD
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(peak<MODE>, dim3(wgs), dim3(512), 0, 0, out, iters, seed);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep >= 1 && ms < best) best = ms;  // rep 0 = clock ramp
    }
    (void)hipFree(out);
    return (double)wgs * 8 * iters * per_iter * flop_per_mfma / (best * 1e-3) * 1e-12;
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double bf = 2.0 * 32 * 32 * 16, f8 = 2.0 * 32 * 32 * 64, f32 = 2.0 * 32 * 32 * 2;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"lds_per_cu_kib\": %d, "
           "\"vendor_peak_tflops\": {\"bf16\": 2500, \"fp8\": 5000, \"f32_matrix\": 157.3}, "
           "\"mfma_only_tflops\": {\"bf16_zeros\": %.0f, \"bf16_random\": %.0f, \"fp8_zeros\": %.0f, \"fp8_random\": %.0f, "
           "\"f32_zeros\": %.1f, \"f32_random\": %.1f, \"bf16_16x16x32_zeros\": %.0f, \"bf16_16x16x32_random\": %.0f, "
           "\"f16_zeros\": %.0f, \"f16_random\": %.0f}}\n",
           p.name, p.gcnArchName, cus, p.clockRate / 1000, (int)(p.maxSharedMemoryPerMultiProcessor / 1024),
           run<0>(cus, 0, bf), run<0>(cus, 7, bf), run<1>(cus, 0, f8), run<1>(cus, 7, f8), run<2>(cus, 0, f32), run<2>(cus, 7, f32),
           run<3>(cus, 0, 2.0 * 16 * 16 * 32, 8), run<3>(cus, 7, 2.0 * 16 * 16 * 32, 8), run<4>(cus, 0, bf), run<4>(cus, 7, bf));
    return 0;
}
