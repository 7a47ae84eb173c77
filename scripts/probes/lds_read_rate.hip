// LDS read throughput per CU for the operand reads the kernels use: ds_read_b128, ds_read_b64,
// ds_read_b64_tr_b16, ds_read_b64_tr_b8.  One workgroup of 8 waves per CU, conflict-free addresses (lane * size),
// ITER x 8 independent reads per wave between lgkmcnt(0) waits; prints bytes / clock / CU at the measured SCLK-free
// rate (s_memtime runs at 100 MHz, so the rate is derived from wall time and a fixed 2.4 GHz nominal -- compare rows).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_read_rate scripts/probes/lds_read_rate.hip && /tmp/lds_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define LDS_PTR(T) __attribute__((address_space(3))) T *

template <int MODE> __global__ __launch_bounds__(512) void probe(unsigned *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_PTR(char) lds = (LDS_PTR(char))smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 512) ((LDS_PTR(unsigned))lds)[i] = i;
    __syncthreads();
    unsigned acc = 0;
    const int base = wave * 8192 + lane * (MODE == 0 ? 16 : 8);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int off = base + k * (MODE == 0 ? 1024 : 512);  // eight distinct addresses: nothing to merge
            if (MODE == 0) {
                const u32x4 v = *(LDS_PTR(u32x4))(lds + off);
                acc += v[0] ^ v[3];
            } else if (MODE == 1) {
                const u32x2 v = *(LDS_PTR(u32x2))(lds + off);
                acc += v[0] ^ v[1];
            } else if (MODE == 2) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + off));
                acc += (unsigned)v[0] ^ (unsigned)v[3];
            } else {
                const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((LDS_PTR(i32x2))(lds + off));
                acc += (unsigned)v[0] ^ (unsigned)v[1];
            }
        }
        asm volatile("" ::: "memory");
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE> void run(const char *name, int bytes_per_lane) {
    unsigned *out;
    (void)hipMalloc(&out, 4);
    const int iters = 20000, wgs = 256;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), 65536, 0, out, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
    }
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes_per_cu = (double)iters * 8 * 8 * 64 * bytes_per_lane;  // 8 reads x 8 waves x 64 lanes
    printf("%-22s %8.3f ms  %7.1f GB/s per CU  = %6.1f B/clk at 2.4 GHz\n", name, ms, bytes_per_cu / ms * 1e-6, bytes_per_cu / (ms * 1e-3) / 2.4e9);
    (void)hipFree(out);
}

int main() {
    run<0>("ds_read_b128", 16);
    run<1>("ds_read_b64", 8);
    run<2>("ds_read_b64_tr_b16", 8);
    run<3>("ds_read_b64_tr_b8", 8);
    return 0;
}
