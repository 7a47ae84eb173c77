// v_cvt_scalef32_pk_fp8_f32 (gfx950) with scale 1.0 against v_cvt_pk_fp8_f32: the same bytes?  (and the bf8 pair)
//   hipcc --offload-arch=gfx950 -O2 -o cvt_scale cvt_scale.hip && ./cvt_scale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(const float *x, uint32_t *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float a = x[2 * i], b = x[2 * i + 1];
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b));
    asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, 1.0" : "+v"(r1) : "v"(a), "v"(b));
    asm volatile("v_cvt_pk_bf8_f32 %0, %1, %2" : "+v"(r2) : "v"(a), "v"(b));
    asm volatile("v_cvt_scalef32_pk_bf8_f32 %0, %1, %2, 1.0" : "+v"(r3) : "v"(a), "v"(b));
    o[4 * i] = r0; o[4 * i + 1] = r1; o[4 * i + 2] = r2; o[4 * i + 3] = r3;
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    uint32_t s = 12345;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        uint32_t bits;
        if (i < n / 2) { bits = s; }                              // any bit pattern (NaN, inf, denormals, huge)
        else { float f = (float)((s >> 8) & 0xFFFF) / 65536.0f * 600.0f; memcpy(&bits, &f, 4); if (s & 1) bits |= 0x80000000u; }   // [0, 600): the fp8 range and beyond
        memcpy(&h[i], &bits, 4);
    }
    h[0] = 448.0f; h[1] = 449.0f; h[2] = 464.0f; h[3] = 480.0f; h[4] = 1e30f; h[5] = 0.001f; h[6] = 0.0009765625f; h[7] = 57344.0f;
    float *dx; uint32_t *d_o;
    hipMalloc(&dx, n * 4); hipMalloc(&d_o, n * 8);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(dx, d_o, n);
    std::vector<uint32_t> o(2 * n);
    hipMemcpy(o.data(), d_o, n * 8, hipMemcpyDeviceToHost);
    long d8 = 0, db = 0;
    for (int i = 0; i < n / 2; ++i) {
        if ((o[4 * i] & 0xFFFF) != (o[4 * i + 1] & 0xFFFF)) { if (d8 < 8) printf("fp8 differs: %g %g -> %04x vs %04x\n", h[2 * i], h[2 * i + 1], o[4 * i] & 0xFFFF, o[4 * i + 1] & 0xFFFF); ++d8; }
        if ((o[4 * i + 2] & 0xFFFF) != (o[4 * i + 3] & 0xFFFF)) { if (db < 8) printf("bf8 differs: %g %g -> %04x vs %04x\n", h[2 * i], h[2 * i + 1], o[4 * i + 2] & 0xFFFF, o[4 * i + 3] & 0xFFFF); ++db; }
    }
    printf("pairs %d: fp8 differing %ld, bf8 differing %ld\n", n / 2, d8, db);
    for (int i = 0; i < 4; ++i) printf("  %g %g -> fp8 %04x / %04x  bf8 %04x / %04x\n", h[2 * i], h[2 * i + 1], o[4 * i] & 0xFFFF, o[4 * i + 1] & 0xFFFF, o[4 * i + 2] & 0xFFFF, o[4 * i + 3] & 0xFFFF);
    return 0;
}
