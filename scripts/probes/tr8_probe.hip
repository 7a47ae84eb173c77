// Probe: lane mapping of ds_read_b64_tr_b8 on gfx950 (needed for an fp8 P.V operand read; the ISA document that
// defines it is not in this image).  Lane l supplies LDS address l*8; LDS byte x holds its own index.  Prints, for
// every lane and result byte, which (source lane, byte) it received.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) int i32x2;
__global__ void k(int *out, int hi) {
    __shared__ __attribute__((aligned(16))) unsigned char s[1024];
    for (int x = threadIdx.x; x < 1024; x += 64) s[x] = hi ? (unsigned char)(x >> 8) : (unsigned char)(x & 255);
    __syncthreads();
    i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2 *)(s + threadIdx.x * 8));
    out[threadIdx.x * 2] = v[0];
    out[threadIdx.x * 2 + 1] = v[1];
}
int main() {
    int *d, lo[128], hi[128];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d, 0); hipMemcpy(lo, d, 512, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d, 1); hipMemcpy(hi, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 8; ++e) {
            int pos = ((((unsigned)hi[l * 2 + e / 4] >> (8 * (e % 4))) & 255) << 8) | (((unsigned)lo[l * 2 + e / 4] >> (8 * (e % 4))) & 255);
            printf(" (%2d,%d)", pos / 8, pos % 8);
        }
        printf("\n");
    }
    return 0;
}
