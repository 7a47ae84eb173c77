#!/usr/bin/env python3
"""Job timeline of the generated assembly kernel from in-kernel stamps (diagnostic library `make -C ... stamps`).

    FA2_HIP_LIB=flash_attention_dlrs_amd/libfa2_hip_stamps.so python benchmarks/a64_stamps.py c3_noncausal

Slots per (workgroup, wave): 0 job start (loop entry), 3 steady loop end, 4 seam body end, 5 epilogue end (all of the LAST
job of the workgroup); 6 / 7 s_memrealtime (100 MHz) at kernel start / end, 8 / 9 s_memtime there; 10..12 cycles summed
over the last job's steady loop: phase A, barrier wait, phase B.  The stamped build's run time is not the product's: read shares, not lengths.
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, TORCH_DTYPE, flops  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3_noncausal"
c = CONFIGS[cfg]
dev = torch.device("cuda:0")
NSLOT = 24
nwg = 256
dbg = torch.zeros(nwg * 4 * NSLOT, dtype=torch.int64, device=dev)
os.environ["FA2_A64_DBG"] = hex(dbg.data_ptr())
from flash_attention_dlrs_amd import flash_attention_forward  # noqa: E402

torch.manual_seed(42)
data = sys.argv[2] if len(sys.argv) > 2 else "randn"
spread = 0.5 if c["dtype"] == "fp8" else 1.0        # (fp8 inputs as bench.py / benchmarks/variants.py draw them)
Q, K, V = (torch.randn(c["B"], c["H"], c["N"], c["d"], device=dev) * spread for _ in range(3))
if data == "small":        # the usual 1/sqrt(d) softmax scale folded into Q (scores of unit variance: the deferred maximum never moves)
    Q = Q * 128 ** -0.5 / spread ** 2
Q, K, V = (t.to(TORCH_DTYPE[c["dtype"]]) for t in (Q, K, V))
if data == "zeros":
    Q.zero_(); K.zero_(); V.zero_()
elif data == "kzero":      # scores all zero, V random
    K.zero_()
elif data == "vzero":
    V.zero_()
elif data == "small":
    pass
elif data == "ones":
    Q.fill_(0.1); K.fill_(0.1); V.fill_(1.0)
VARIANT = os.environ.get("A64_STAMPS_VARIANT", "a64")     # "a8": the fp8 kernels (their stamped forms: FA2_A64_KERNEL=fa2_fwd_a8_...)
for _ in range(20):
    flash_attention_forward(Q, K, V, dev, causal=c["causal"], variant=VARIANT)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    flash_attention_forward(Q, K, V, dev, causal=c["causal"], variant=VARIANT)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
d = dbg.cpu().view(nwg, 4, NSLOT).double()
tiles = c["N"] // 64
out = {"config": cfg, "data": data, "kernel": os.environ.get("FA2_A64_KERNEL", ""), "ms": round(ms, 4), "tflops": round(flops(c) / ms / 1e9, 1)}
seg = {"loop": d[..., 3] - d[..., 0], "seam": d[..., 4] - d[..., 3], "epilogue": d[..., 5] - d[..., 4], "kernel": d[..., 9] - d[..., 8]}
for k, v in seg.items():
    out[k + "_cyc_median"] = float(v.median())
if not c["causal"]:
    out["cyc_per_step_loop"] = round(float((seg["loop"] / (tiles - 4)).median()), 1)
    out["cyc_per_mfma_loop"] = round(out["cyc_per_step_loop"] / 64, 2)
lo = dbg.cpu().view(nwg, 4, NSLOT) & 0xFFFFFFFF
steps = tiles - 4
for k, nm in ((10, "phaseA"), (11, "sync"), (12, "phaseB")):
    out[nm + "_cyc_per_step"] = [round(float(lo[:, w, k].double().median()) / steps, 1) for w in range(4)]
out["seam_steps_cyc"] = [float((d[..., 17] - d[..., 16]).median()), float((d[..., 18] - d[..., 17]).median()),
                         float((d[..., 19] - d[..., 18]).median()), float((d[..., 4] - d[..., 19]).median()), float((d[..., 16] - d[..., 3]).median())]
# the same per wave class of the split causal kernel (waves 0-1 take the "low" bodies, waves 2-3 the "high" ones): a wave that is
# through a step early waits at the next step's barrier, so the class with the LONGER steps is the critical one
if c["causal"]:
    for nm, sl in (("low", slice(0, 2)), ("high", slice(2, 4))):
        w = d[:, sl, :]
        out["seam_steps_" + nm] = [float((w[..., 17] - w[..., 16]).median()), float((w[..., 18] - w[..., 17]).median()),
                                   float((w[..., 19] - w[..., 18]).median()), float((w[..., 4] - w[..., 19]).median())]
# seam steps 2 and 3 of the split causal kernel at their mid-step barrier (slots 22, 23; waves 2, 3 carry them): phase A + barrier | phase B
if c["causal"]:
    hi = d[:, 2:, :]
    out["seam_step2_AB"] = [float((hi[..., 22] - hi[..., 18]).median()), float((hi[..., 19] - hi[..., 22]).median())]
    out["seam_step3_AB"] = [float((hi[..., 23] - hi[..., 19]).median()), float((hi[..., 4] - hi[..., 23]).median())]
# inside the epilogue (slot 4 = its start): L + descriptors | block 0 scale/pack/write + read-back issue | block 1 first half |
# block 0 stores + block 1 second half + read-back issue | O := 0 | block 1 stores
pts = [d[..., 4], d[..., 13], d[..., 14], d[..., 15], d[..., 20], d[..., 21], d[..., 5]]
out["epilogue_parts_cyc"] = [float((b - a).median()) for a, b in zip(pts[:-1], pts[1:])]
if "lite" in os.environ.get("FA2_A64_KERNEL", ""):   # sums over all jobs of a workgroup: steady loops, seam bodies, epilogues, first fill
    out["all_jobs_cyc"] = {nm: float(lo[:, 0, k].double().median()) for nm, k in (("steady", 10), ("seam", 11), ("epilogue", 12))}
# spread over the workgroups (wave 0): start skew, length, end skew, in ns
st, en = d[:, 0, 6] * 10.0, d[:, 0, 7] * 10.0     # s_memrealtime ticks (100 MHz, common to the XCDs) -> ns
q = lambda x: [float(x.min()), float(x.median()), float(x.max())]
out["wg_start_rel"] = q(st - st.min())
out["wg_len"] = q(en - st)
out["wg_end_rel"] = q(en - st.min())
out["wg_len_by_xcd_us"] = [round(float((en - st)[x::8].mean()) / 1e3, 1) for x in range(8)]   # workgroup id % 8 = XCD
out["wg_len_spread_in_xcd_us"] = [round(float(((en - st)[x::8].max() - (en - st)[x::8].min())) / 1e3, 1) for x in range(8)]
# pipeline fill of the workgroup's first job: kernel start -> first tiles landed (set-up, descriptors, DMA issue, HBM latency) ->
# end of step -1 (Q / K fragment reads, first QK^T, start of the first softmax)
out["fill_cyc"] = [float((d[..., 1] - d[..., 8]).median()), float((d[..., 2] - d[..., 1]).median())]
real = (d[..., 7] - d[..., 6]).median().item()  # 100 MHz ticks
out["clock_ghz"] = round(float(seg["kernel"].median()) / real / 10.0, 3) if real > 0 else None
out["job_cyc"] = float((d[..., 5] - d[..., 0]).median())
print(json.dumps(out))
