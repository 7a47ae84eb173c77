// Probe: can buffer_load ... lds (LDS-DMA) write LDS addresses >= 64 KiB on gfx950 (is M0's LDS offset wider than 16 bits)?
// Each test DMAs 1 KiB of a known pattern to LDS address `dst`, then reads it back with ds_read and reports mismatches;
// it also checks where the data landed if not at `dst` (dst & 0xFFFF).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
__global__ __launch_bounds__(64) void probe(const unsigned* src, unsigned* out, unsigned dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 160 * 1024 / 4; i += 64) smem[i] = 0xEEEE0000u;
    __syncthreads();
    const unsigned long long ba = (unsigned long long)src;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ba);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba >> 32) & 0xffffu));
    r[2] = 4096; r[3] = 0x00020000;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(dst);
    int voff = lane * 16;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(m0v), "v"(voff), "s"(r) : "memory");
    __syncthreads();
    // report: matches at dst, matches at dst & 0xffff, matches at dst & 0x3ffff
    int at_dst = 0, at_lo = 0;
    for (int k = 0; k < 4; ++k) {
        at_dst += smem[dst / 4 + lane * 4 + k] == src[lane * 4 + k];
        at_lo += smem[(dst & 0xffff) / 4 + lane * 4 + k] == src[lane * 4 + k];
    }
    out[lane] = at_dst;
    out[64 + lane] = at_lo;
}
int main() {
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0xABC00000u + i;
    unsigned *src, *out;
    hipMalloc(&src, 4096); hipMalloc(&out, 512);
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (unsigned dst : {1024u, 65536u - 1024u, 65536u, 65536u + 4096u, 131072u, 131072u + 8192u, 163840u - 1024u}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, src, out, dst);
        unsigned r[128];
        hipError_t e = hipMemcpy(r, out, 512, hipMemcpyDeviceToHost);
        int a = 0, b = 0;
        for (int i = 0; i < 64; ++i) { a += r[i]; b += r[64 + i]; }
        printf("dst=%6u : %3d/256 dwords at dst, %3d/256 at (dst & 0xffff)  [%s]\n", dst, a, b, hipGetErrorString(e));
    }
    return 0;
}
