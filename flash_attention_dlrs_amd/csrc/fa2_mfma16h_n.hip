// Non-causal instantiations of fa2_mfma16h.hip, alone in their translation unit.
#define FA2_H_INST 2
#define FA2_H_ENTRY fa2_launch_mfma16h_noncausal
#include "fa2_mfma16h.hip"
