// Shared declarations of the gfx950 FA-2 forward kernels (internal; the public ABI is include/fa2_fwd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fa2_fwd.h"

// One forward problem, as handed over by fa2_fwd().  Strides in elements (reference convention,
// src/flash_attention_torch.py:53-57).
struct Fa2Problem {
    const void *Q, *K, *V;
    void *O, *L;
    int64_t qs[4], ks[4], vs[4], os[4], ls[2];
    int32_t B, H, N, d;
    int32_t dtype, causal;
    float scale;
    hipStream_t stream;
};

// Launchers, one per translation unit.  Return FA2_OK / FA2_ERR_*; set_error() on failure.
int fa2_launch_generic(const Fa2Problem &p);
int fa2_launch_mfma16(const Fa2Problem &p, int waves);
int fa2_launch_mfma32(const Fa2Problem &p);
int fa2_launch_mfma16d(const Fa2Problem &p, int waves);
int fa2_launch_mfma8x(const Fa2Problem &p, int waves);
int fa2_launch_mfma16k(const Fa2Problem &p, int shape);
#ifdef FA2_EXPERIMENTS   // libfa2_hip_exp.so only (fa2_experiments.h)
#include "fa2_experiments.h"
int fa2_launch_mfma16p(const Fa2Problem &p, int waves, int opt);
int fa2_launch_mfma16x(const Fa2Problem &p, int abl);
int fa2_launch_mfma8(const Fa2Problem &p, int waves);
int fa2_launch_mfma16s(const Fa2Problem &p, int waves);
#endif
int fa2_launch_mfma16h(const Fa2Problem &p, int waves);  // dispatches to the two translation units below
int fa2_launch_mfma16h_causal(const Fa2Problem &p, int waves);
int fa2_launch_mfma16h_noncausal(const Fa2Problem &p, int waves);
int fa2_launch_a64(const Fa2Problem &p);  // generated assembly kernel (asm/fa2_a64_gen.py)
bool fa2_a64_supports(const Fa2Problem &p);
int fa2_launch_a16(const Fa2Problem &p);  // the same structure on v_mfma_f32_16x16x32 (asm/fa2_a16_gen.py)
bool fa2_a16_supports(const Fa2Problem &p);
int fa2_launch_a8(const Fa2Problem &p);   // ... and on the fp8 matrix path, v_mfma_f32_32x32x64_f8f6f4 (asm/fa2_a8_gen.py)
bool fa2_a8_supports(const Fa2Problem &p);
int fa2_launch_a64d(const Fa2Problem &p); // ... and at head size 64 (asm/fa2_a64d_gen.py)
bool fa2_a64d_supports(const Fa2Problem &p);
bool fa2_mfma8_supports(const Fa2Problem &p);
bool fa2_mfma8x_supports(const Fa2Problem &p);
bool fa2_mfma16_supports(const Fa2Problem &p);
bool fa2_mfma16_supports_dp(const Fa2Problem &p);   // + the other multiples of 8 up to 128 (fa2_mfma16.hip only)
bool fa2_mfma32_supports(const Fa2Problem &p);

void fa2_set_error(const char *fmt, ...);
// Tuning knobs for A/B runs.  The product build returns `dflt` without touching the environment (no getenv on the launch
// path); `make experiments` (-DFA2_TUNING_ENV) reads the variable per call.
int fa2_env_int(const char *name, int dflt);

// Per-DEVICE launch state (a process may drive several GPUs): the CU count of the current device, and a once-per-device
// latch for hipFuncSetAttribute(MaxDynamicSharedMemorySize) -- function attributes are per device.
int fa2_device_cus();
struct Fa2DeviceLatch {
    unsigned long long done = 0;  // bit d: applied on device d (a racing second application is harmless: idempotent)
    bool need() const;            // true if the current device has not been marked yet
    void mark();
};

static inline int fa2_dtype_size(int dt) {
    switch (dt) {
    case FA2_DTYPE_F64: return 8;
    case FA2_DTYPE_F32: return 4;
    case FA2_DTYPE_F16:
    case FA2_DTYPE_BF16: return 2;
    case FA2_DTYPE_F8E5M2:
    case FA2_DTYPE_F8E4M3: return 1;
    default: return 0;
    }
}

#define FA2_LOG2E 1.4426950408889634  // np.log2(np.e), src/flash_attention_kernels.py:9
