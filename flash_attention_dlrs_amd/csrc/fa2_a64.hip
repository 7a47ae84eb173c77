// fa2_a64.hip -- host side of the generated assembly kernels `fa2_fwd_a64_<dtype>_<c|n>` (variant "a64"):
// f16 / bf16, d = 128, N >= 256 (a multiple of 256: the plain kernels; else the "ragged" ones: range-checked descriptors,
// masked key tail); 4 waves x 64 query rows, one wave per SIMD with the whole register file,
// persistent grid.  The kernels are produced by asm/fa2_a64_gen.py (instruction stream, register map and kernel-argument
// layout are documented there), assembled into a gfx950 code object and embedded in this library (fa2_a64_blob.S); they
// are loaded once per device with hipModuleLoadData and launched with hipModuleLaunchKernel.
// Reference arithmetic: /root/reference/src/flash_attention_kernels.py:84-108.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "fa2_common.h"

extern "C" const unsigned char fa2_a64_hsaco_start[];
extern "C" const unsigned char fa2_a64_hsaco_end[];

namespace {

// kernel-argument block: the layout of Gen.k_setup() in asm/fa2_a64_gen.py (192 bytes)
struct __attribute__((packed)) A64Args {
    const void *Q, *K, *V;
    void *O, *L;
    int64_t qs_b, qs_h, ks_b, ks_h, vs_b, vs_h, os_b, os_h, ls_b, ls_h;  // bytes
    int32_t qs_n, ks_n, vs_n, os_n;                                        // bytes
    int32_t N, H, nq, total;
    float c_log2e, thr;
    int32_t nunit, group;
    int32_t nbh, nwg;
    void *dbg;
    uint32_t lg;   // lgH | lgG << 8 | lg(G * nunit) << 16 | 1 << 24 when those are powers of two and B * H % 8 == 0; bit 25: light causal jobs walk downwards
    uint32_t pad;
};
static_assert(sizeof(A64Args) == 192, "kernel-argument layout");

constexpr int kMaxDev = 64;
struct DevState {
    bool ready = false, failed = false;
    hipModule_t mod = nullptr;
    hipFunction_t fn[2][2][2][2] = {};  // [a64 / a16][bf16 / f16][non-causal / causal][N % 256 == 0 / ragged]
    hipFunction_t fn8[2][2][2] = {};    // a8: [e4m3 / e5m2][non-causal / causal][N % 256 == 0 / ragged]
    hipFunction_t fnd[2][2][2] = {};    // a64d (head size 64): [bf16 / f16][non-causal / causal][N % 256 == 0 / ragged]
    int cus = 0;
};
DevState g_dev[kMaxDev];
std::mutex g_mu;

DevState *dev_state() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
        fa2_set_error("a64: hipGetDevice failed or device ordinal out of range");
        return nullptr;
    }
    DevState &d = g_dev[dev];
    std::lock_guard<std::mutex> lk(g_mu);
    if (d.ready) return &d;
    if (d.failed) {
        fa2_set_error("a64: code object could not be loaded on device %d", dev);
        return nullptr;
    }
    // The first launch on a device loads the code object.  That launch may sit inside a stream capture (a caller building a
    // HIP graph without a warm-up call): loading a module is not a stream operation, but under the default (global) capture
    // mode the runtime refuses "unsafe" calls from a capturing thread -- so this thread is relaxed for the duration of the load.
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    const bool exchanged = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    hipError_t e = hipModuleLoadData(&d.mod, (const void *)fa2_a64_hsaco_start);
    if (exchanged) (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (e != hipSuccess) {
        d.failed = true;
        fa2_set_error("a64: hipModuleLoadData failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    for (int m = 0; m < 2; ++m)
        for (int t = 0; t < 2; ++t)
            for (int c = 0; c < 2; ++c)
                for (int r = 0; r < 2; ++r) {
                    char nm[64];
                    snprintf(nm, sizeof(nm), "fa2_fwd_%s_%s_%s%s", m ? "a16" : "a64", t ? "f16" : "bf16", c ? "c" : "n", r ? "r" : "");
                    e = hipModuleGetFunction(&d.fn[m][t][c][r], d.mod, nm);
                    if (e != hipSuccess) d.fn[m][t][c][r] = nullptr;  // a kernel the generator did not emit: reported at launch
                }
    for (int t = 0; t < 2; ++t)
        for (int c = 0; c < 2; ++c)
            for (int r = 0; r < 2; ++r) {
                char nm[64];
                snprintf(nm, sizeof(nm), "fa2_fwd_a8_%s_%s%s", t ? "e5m2" : "e4m3", c ? "c" : "n", r ? "r" : "");
                e = hipModuleGetFunction(&d.fn8[t][c][r], d.mod, nm);
                if (e != hipSuccess) d.fn8[t][c][r] = nullptr;
            }
    for (int t = 0; t < 2; ++t)
        for (int c = 0; c < 2; ++c)
            for (int r = 0; r < 2; ++r) {
                char nm[64];
                snprintf(nm, sizeof(nm), "fa2_fwd_a64d_%s_%s%s", t ? "f16" : "bf16", c ? "c" : "n", r ? "r" : "");
                e = hipModuleGetFunction(&d.fnd[t][c][r], d.mod, nm);
                if (e != hipSuccess) d.fnd[t][c][r] = nullptr;
            }
    (void)hipGetLastError();  // a failed lookup must not surface in another launcher's hipGetLastError()
    d.cus = fa2_device_cus();
    d.ready = true;
    return &d;
}

}  // namespace

bool fa2_a64_supports(const Fa2Problem &p) {
    if (p.dtype != FA2_DTYPE_BF16 && p.dtype != FA2_DTYPE_F16) return false;
    if (p.d != 128 || p.N < 256) return false;   // N a multiple of 256: the plain kernels; any other N above 256: the ragged ones
    if (p.qs[3] != 1 || p.ks[3] != 1 || p.vs[3] != 1 || p.os[3] != 1) return false;
    if (!(p.scale > 0.0f) || !isfinite(p.scale)) return false;
    const int64_t rows[4] = {p.qs[2], p.ks[2], p.vs[2], p.os[2]};
    for (int k = 0; k < 4; ++k)
        if (rows[k] < 128 || (rows[k] & 7) || (int64_t)(p.N + 512) * rows[k] * 2 >= (1LL << 31)) return false;
    const uintptr_t ptrs[4] = {(uintptr_t)p.Q, (uintptr_t)p.K, (uintptr_t)p.V, (uintptr_t)p.O};
    for (int k = 0; k < 4; ++k)
        if (ptrs[k] & 15) return false;
    if (((p.qs[0] | p.qs[1] | p.ks[0] | p.ks[1] | p.vs[0] | p.vs[1] | p.os[0] | p.os[1]) & 7) != 0) return false;
    if ((uintptr_t)p.L & 1) return false;
    const int64_t nq = (p.N + 255) / 256, jobs = (int64_t)p.B * p.H * nq;
    if (jobs >= (1 << 22) || p.H >= (1 << 22) || p.ls[1] < p.N) return false;
    return true;
}

namespace {
int launch(const Fa2Problem &p, int shape16);
}

// fp8: one byte per element, rows of 128 bytes; otherwise the conditions of the 16-bit kernels (16-byte aligned rows and bases,
// N * row stride below 2 GiB); N >= 256 (a multiple of 256: the plain kernels, else the ragged ones; every other fp8 shape: fa2_mfma8x.hip)
bool fa2_a8_supports(const Fa2Problem &p) {
    if (p.dtype != FA2_DTYPE_F8E4M3 && p.dtype != FA2_DTYPE_F8E5M2) return false;
    if (p.d != 128 || p.N < 256) return false;
    if (p.qs[3] != 1 || p.ks[3] != 1 || p.vs[3] != 1 || p.os[3] != 1) return false;
    if (!(p.scale > 0.0f) || !isfinite(p.scale)) return false;
    const int64_t rows[4] = {p.qs[2], p.ks[2], p.vs[2], p.os[2]};
    for (int k = 0; k < 4; ++k)
        if (rows[k] < 128 || (rows[k] & 15) || (int64_t)(p.N + 512) * rows[k] >= (1LL << 31)) return false;
    const uintptr_t ptrs[4] = {(uintptr_t)p.Q, (uintptr_t)p.K, (uintptr_t)p.V, (uintptr_t)p.O};
    for (int k = 0; k < 4; ++k)
        if (ptrs[k] & 15) return false;
    if (((p.qs[0] | p.qs[1] | p.ks[0] | p.ks[1] | p.vs[0] | p.vs[1] | p.os[0] | p.os[1]) & 15) != 0) return false;
    const int64_t nq = (p.N + 255) / 256, jobs = (int64_t)p.B * p.H * nq;
    if (jobs >= (1 << 22) || p.H >= (1 << 22) || p.ls[1] < p.N) return false;
    return true;
}
int fa2_launch_a8(const Fa2Problem &p) { return launch(p, 2); }

// head size 64, f16 / bf16: rows of 128 bytes; N a multiple of 256 (no ragged form); otherwise the conditions of a64
bool fa2_a64d_supports(const Fa2Problem &p) {
    if (p.dtype != FA2_DTYPE_BF16 && p.dtype != FA2_DTYPE_F16) return false;
    if (p.d != 64 || p.N < 256) return false;    // N a multiple of 256: the plain kernels; any other N above 256: the ragged ones
    if (p.qs[3] != 1 || p.ks[3] != 1 || p.vs[3] != 1 || p.os[3] != 1) return false;
    if (!(p.scale > 0.0f) || !isfinite(p.scale)) return false;
    const int64_t rows[4] = {p.qs[2], p.ks[2], p.vs[2], p.os[2]};
    for (int k = 0; k < 4; ++k)
        if (rows[k] < 64 || (rows[k] & 7) || (int64_t)(p.N + 512) * rows[k] * 2 >= (1LL << 31)) return false;
    const uintptr_t ptrs[4] = {(uintptr_t)p.Q, (uintptr_t)p.K, (uintptr_t)p.V, (uintptr_t)p.O};
    for (int k = 0; k < 4; ++k)
        if (ptrs[k] & 15) return false;
    if (((p.qs[0] | p.qs[1] | p.ks[0] | p.ks[1] | p.vs[0] | p.vs[1] | p.os[0] | p.os[1]) & 7) != 0) return false;
    if ((uintptr_t)p.L & 1) return false;
    const int64_t nq = (p.N + 255) / 256, jobs = (int64_t)p.B * p.H * nq;
    if (jobs >= (1 << 22) || p.H >= (1 << 22) || p.ls[1] < p.N) return false;
    return true;
}
int fa2_launch_a64d(const Fa2Problem &p) { return launch(p, 3); }



// The a16 kernels take what the a64 kernels take (same argument block, same job stream).
bool fa2_a16_supports(const Fa2Problem &p) { return fa2_a64_supports(p); }
int fa2_launch_a64(const Fa2Problem &p) { return launch(p, 0); }
int fa2_launch_a16(const Fa2Problem &p) { return launch(p, 1); }

namespace {
int launch(const Fa2Problem &p, int shape16) {
    const bool f8 = shape16 == 2, d64 = shape16 == 3;
    if (d64) {
        if (!fa2_a64d_supports(p)) {
            fa2_set_error("a64d kernel: needs f16/bf16, d = 64, N >= 256, unit d-stride, 16-byte aligned rows, scale > 0");
            return FA2_ERR_UNSUPPORTED;
        }
    } else if (f8) {
        if (!fa2_a8_supports(p)) {
            fa2_set_error("a8 kernel: needs fp8 (e4m3fn / e5m2), d = 128, N >= 256, unit d-stride, 16-byte aligned rows");
            return FA2_ERR_UNSUPPORTED;
        }
    } else if (!fa2_a64_supports(p)) {
        fa2_set_error("a64 kernel: needs f16/bf16, d = 128, N >= 256, unit d-stride, 16-byte aligned rows, "
                      "scale > 0, N * row stride < 2 GiB");
        return FA2_ERR_UNSUPPORTED;
    }
    DevState *d = dev_state();
    if (!d) return FA2_ERR_LAUNCH;
    hipFunction_t fn = f8 ? d->fn8[p.dtype == FA2_DTYPE_F8E5M2 ? 1 : 0][p.causal ? 1 : 0][(p.N & 255) ? 1 : 0]
                     : d64 ? d->fnd[p.dtype == FA2_DTYPE_F16 ? 1 : 0][p.causal ? 1 : 0][(p.N & 255) ? 1 : 0]
                           : d->fn[shape16][p.dtype == FA2_DTYPE_F16 ? 1 : 0][p.causal ? 1 : 0][(p.N & 255) ? 1 : 0];
    if (!fn) {
        fa2_set_error("%s kernel: this (dtype, causal, ragged N) form is not in the code object", f8 ? "a8" : d64 ? "a64d" : shape16 ? "a16" : "a64");
        return FA2_ERR_UNSUPPORTED;
    }
    A64Args a;
    memset(&a, 0, sizeof(a));
    a.Q = p.Q; a.K = p.K; a.V = p.V; a.O = p.O; a.L = p.L;
    const int es = f8 ? 1 : 2;    // bytes per element
    a.qs_b = p.qs[0] * es; a.qs_h = p.qs[1] * es; a.ks_b = p.ks[0] * es; a.ks_h = p.ks[1] * es;
    a.vs_b = p.vs[0] * es; a.vs_h = p.vs[1] * es; a.os_b = p.os[0] * es; a.os_h = p.os[1] * es;
    a.ls_b = p.ls[0] * es; a.ls_h = p.ls[1] * es;
    a.qs_n = (int32_t)(p.qs[2] * es); a.ks_n = (int32_t)(p.ks[2] * es); a.vs_n = (int32_t)(p.vs[2] * es); a.os_n = (int32_t)(p.os[2] * es);
    a.N = p.N; a.H = p.H; a.nq = (p.N + 255) / 256;
    a.nunit = p.causal ? (a.nq + 1) / 2 : a.nq;
    a.nbh = p.B * p.H;
    a.total = a.nunit * a.nbh;
    a.c_log2e = (float)((double)p.scale * FA2_LOG2E);
    // Deferred running-max threshold (log2 units): P may reach 2^thr before the running maximum is raised (and O, l are
    // rescaled: ~2 500 cycles during which the other three waves wait at the barrier).  bf16 P has the fp32 exponent range:
    // 2^60 * N * |V| stays far below fp32 overflow in l and O, and on N(0,1) inputs at scale 1 (score sigma ~ 16 log2 units)
    // 24 still fired a few times per job and wave -- measured 3 443 vs 2 946 cycles per tile step.  f16 P must stay below 65504.
    // (2^15.875 = 60 097 < 65 504; 12 -> 15.875: +3..4 % on the reference bench's fp16 shape, fewer rescales)
    a.thr = p.dtype == FA2_DTYPE_F16 ? 15.875f : 60.0f;
    if (f8) a.thr = p.dtype == FA2_DTYPE_F8E4M3 ? 8.5f : 15.0f;      // (P <= 2^8.5 inside e4m3's 448, 2^15 inside e5m2's 57 344; fa2_mfma8x.hip's)
    a.group = 1;
    if (p.causal && (a.nbh & 7) == 0) {
        const int per_xcd = a.nbh / 8;
        int g = 2 > per_xcd ? per_xcd : 2;
        while (per_xcd % g) --g;
        a.group = g;
    }
    // Causal: the LIGHT job of a unit walks its non-diagonal key tiles downwards (asm/fa2_a64_gen.py, Gen.pairs): the light jobs of
    // a head then form one stream in lockstep that meets the tiles in the reverse of the order the heavy jobs left them in L2,
    // instead of each starting again at tile 0 at a time of its own.  The order is a function of (query block, nq) alone, so a
    // head's result does not depend on the launch it is part of (bit-identical head sharding).  Same-device A/B against
    // FA2_A64_PAIRS=0 (profiles/r03/pairs_ab.jsonl): c3 +1 %, N = 2048 +2.8 %, N >= 8192 0 .. -0.6 % (left alone there).
    bool down = p.causal && (p.N & 255) == 0 && a.nq >= 2 && a.nq <= 16 && (shape16 == 0 || d64 || f8);
#ifdef FA2_A64_STAMPS
    down = false;      // (the diagnostic kernels use the registers for the debug pointer)
#endif
#ifdef FA2_A64_VARIANTS
    {
        const char *pv = getenv("FA2_A64_PAIRS");     // A/B runs: FA2_A64_PAIRS=0: every job walks upwards
        if (pv && *pv == '0') down = false;
        const char *gv = getenv("FA2_A64_GROUP");     // A/B runs: heads per XCD group of the causal unit order
        if (gv && *gv && p.causal && (a.nbh & 7) == 0) {
            int g = atoi(gv);
            const int per_xcd = a.nbh / 8;
            g = g < 1 ? 1 : (g > per_xcd ? per_xcd : g);
            while (per_xcd % g) --g;
            a.group = g;
        }
    }
#endif
    // persistent grid: one workgroup per CU, a multiple of 8 when there is more work than CUs (XCD affinity of the units)
    int slots = d->cus - d->cus % 8;
    if (slots < 8) slots = 8;
    a.nwg = a.total < slots ? a.total : slots;
    a.dbg = nullptr;
    {
        auto pow2 = [](int x) { return x > 0 && (x & (x - 1)) == 0; };
        auto lg2 = [](int x) { int l = 0; while ((1 << l) < x) ++l; return l; };
        const int gn = a.group * a.nunit;
        a.lg = ((a.nbh & 7) == 0 && pow2(p.H) && pow2(a.group) && pow2(gn))
                   ? (uint32_t)(lg2(p.H) | (lg2(a.group) << 8) | (lg2(gn) << 16) | (1 << 24)) : 0u;
        if (down) a.lg |= 1u << 25;
        a.pad = 0;
    }
#ifdef FA2_A64_VARIANTS
    // experiments library only (make experiments): a named variant of the kernel from the same code object, for A/B runs in
    // one process (benchmarks/variants.py: "c3:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_lean")
    {
        const char *kn = getenv("FA2_A64_KERNEL");
        if (kn && *kn && hipModuleGetFunction(&fn, d->mod, kn) != hipSuccess) {
            fa2_set_error("a64 experiments build: kernel %s not found", kn);
            return FA2_ERR_BAD_ARG;
        }
        const char *th = getenv("FA2_A64_THR");    // the deferral threshold (log2 units), for A/B runs
        if (th && *th) a.thr = (float)atof(th);
    }
#endif
#ifdef FA2_A64_STAMPS
    // diagnostic library only (make stamps): the kernels carry s_memtime stamps and write them to the buffer whose device
    // address the harness passes in FA2_A64_DBG (benchmarks/a64_stamps.py)
    {
        const char *v = getenv("FA2_A64_DBG");
        a.dbg = v ? (void *)strtoull(v, nullptr, 0) : nullptr;
        if (!a.dbg) {
            fa2_set_error("a64 stamps build: FA2_A64_DBG is not set");
            return FA2_ERR_BAD_ARG;
        }
        const char *kn = getenv("FA2_A64_KERNEL");  // a timing-only ablation kernel of the diagnostic code object
        if (kn && *kn && hipModuleGetFunction(&fn, d->mod, kn) != hipSuccess) {
            fa2_set_error("a64 stamps build: kernel %s not found", kn);
            return FA2_ERR_BAD_ARG;
        }
    }
#endif
    size_t size = sizeof(a);
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    const hipError_t e = hipModuleLaunchKernel(fn, (unsigned)a.nwg, 1, 1, 256, 1, 1, 0, p.stream, nullptr, extra);
    if (e != hipSuccess) {
        fa2_set_error("a64 kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}
}  // namespace
