// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands: which lane's scale byte applies to which part of the product?
//   hipcc --offload-arch=gfx950 -O2 -o mfma_scale mfma_scale.hip && ./mfma_scale
// Operand layout (measured in round 1, fa2_mfma8x.hip): lane (i = lane & 31, h = lane >> 5) supplies row / column i, registers 0-3 =
// k 16 h .. 16 h + 15, registers 4-7 = k 32 + 16 h .. 32 + 16 h + 15.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// mode 0: A = B = 1 everywhere, scale_a = 127, scale_b = sb[lane]          -> D[i][j] tells the lanes whose scale reaches column j
// mode 1: as 0 with B = 1 only in 16-byte chunk `chunk` (k 16 chunk ..)     -> which block (scale) that chunk belongs to
// mode 2: scale_a = sa[lane], scale_b = 127, A chunk only                   -> the same for A rows
// mode 3: scale registers with different bytes, op_sel = sel                 -> which byte is read
__global__ void probe(float *out, const int *sa, const int *sb, int mode, int chunk, int sel) {
    const int lane = threadIdx.x, h = lane >> 5;
    i32x8 a, b;
    const int one4 = 0x38383838;   // four e4m3 1.0
    for (int r = 0; r < 8; ++r) { a[r] = one4; b[r] = one4; }
    if (mode == 1 || mode == 2) {
        for (int r = 0; r < 8; ++r) {
            const int c = (r < 4) ? h : 2 + h;      // the chunk registers r belong to
            if (c != chunk) { if (mode == 1) b[r] = 0; else a[r] = 0; }
        }
    }
    f32x16 d = {0};
    int va = sa[lane], vb = sb[lane];
    if (sel == 0) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d, 0, 0, 0, va, 0, vb);
    if (sel == 1) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d, 0, 0, 1, va, 1, vb);
    if (sel == 2) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d, 0, 0, 2, va, 2, vb);
    if (sel == 3) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d, 0, 0, 3, va, 3, vb);
    // D layout (32x32 C): lane (j = lane & 31, h): register 4 q + r <-> row 8 q + 4 h + r, column j
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) out[(8 * q + 4 * h + r) * 32 + (lane & 31)] = d[4 * q + r];
}

int main() {
    float *out; int *sa, *sb;
    hipMalloc(&out, 1024 * 4); hipMalloc(&sa, 256); hipMalloc(&sb, 256);
    std::vector<int> ha(64), hb(64);
    std::vector<float> ho(1024);
    auto run = [&](int mode, int chunk, int sel) {
        hipMemcpy(sa, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(sb, hb.data(), 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(out, sa, sb, mode, chunk, sel);
        hipMemcpy(ho.data(), out, 4096, hipMemcpyDeviceToHost);
    };
    // mode 0: scale_b = 127 + (lane & 3) + 4 (lane >> 5)
    for (int l = 0; l < 64; ++l) { ha[l] = 127; hb[l] = 127 + (l & 3) + 4 * (l >> 5); }
    run(0, 0, 0);
    printf("mode 0 (sb = 127 + (lane & 3) + 4 (lane >> 5)):  D[0][j], j = 0..7:");
    for (int j = 0; j < 8; ++j) printf(" %g", ho[j]);
    printf("\n   expected if lane j -> k block 0 of column j and lane j + 32 -> k block 1: 32 * 2^(j & 3) * (1 + 16):");
    for (int j = 0; j < 8; ++j) printf(" %g", 32.0 * (1 << (j & 3)) * 17);
    printf("\n   rows equal? D[5][3] = %g D[31][3] = %g\n", ho[5 * 32 + 3], ho[31 * 32 + 3]);
    // mode 1: which block does chunk c (k 16 c .. 16 c + 15) of B belong to?  scale_b = 127 + 3 h
    for (int l = 0; l < 64; ++l) hb[l] = 127 + 3 * (l >> 5);
    for (int c = 0; c < 4; ++c) { run(1, c, 0); printf("mode 1 B chunk %d (k %d..%d): D[0][0] = %g  (16 = the lower lane half's scale, 128 = the upper's)\n", c, 16 * c, 16 * c + 15, ho[0]); }
    for (int l = 0; l < 64; ++l) { hb[l] = 127; ha[l] = 127 + 3 * (l >> 5); }
    for (int c = 0; c < 4; ++c) { run(2, c, 0); printf("mode 2 A chunk %d: D[0][0] = %g\n", c, ho[0]); }
    // mode 3: bytes: scale register = 127 | 128 << 8 | 129 << 16 | 130 << 24
    for (int l = 0; l < 64; ++l) { ha[l] = 127 | 127 << 8 | 127 << 16 | 127 << 24; hb[l] = 127 | 128 << 8 | 129 << 16 | 130u << 24; }
    for (int s = 0; s < 4; ++s) { run(0, 0, s); printf("mode 3 op_sel %d: D[0][0] = %g (64 x 2^byte index if the selector picks that byte)\n", s, ho[0]); }
    // scale_a per row
    for (int l = 0; l < 64; ++l) { hb[l] = 127; ha[l] = 127 + (l & 3); }
    run(0, 0, 0);
    printf("scale_a = 127 + (lane & 3): D[i][0], i = 0..7:");
    for (int i = 0; i < 8; ++i) printf(" %g", ho[i * 32]);
    printf("\n");
    return 0;
}
