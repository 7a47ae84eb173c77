// Shared declarations of the gfx950 FA-2 backward kernels (internal; the public ABI is include/fa2_bwd.h).
#pragma once
#include "fa2_common.h"

#include "../../include/fa2_bwd.h"

// One backward problem, as handed over by fa2_bwd().  Strides in elements (src/flash_attention_torch.py:110-121).
struct Fa2BwdProblem {
    const void *Q, *K, *V, *O, *dO, *L;
    void *dQ, *dK, *dV, *D;
    int64_t qs[4], ks[4], vs[4], os[4], dos[4], dqs[4], dks[4], dvs[4], ls[2];
    int32_t B, H, N, d;
    int32_t dtype, causal;
    float scale;
    hipStream_t stream;
};

int fa2_bwd_launch_generic(const Fa2BwdProblem &p);
int fa2_bwd_launch_mfma16(const Fa2BwdProblem &p);
bool fa2_bwd_mfma16_supports(const Fa2BwdProblem &p);
int fa2_bwd_launch_mfma32(const Fa2BwdProblem &p);
bool fa2_bwd_mfma32_supports(const Fa2BwdProblem &p);
