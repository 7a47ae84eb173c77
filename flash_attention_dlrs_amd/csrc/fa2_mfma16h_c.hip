// Causal instantiations of fa2_mfma16h.hip, alone in their translation unit (see the note at the launch site there).
#define FA2_H_INST 1
#define FA2_H_ENTRY fa2_launch_mfma16h_causal
#include "fa2_mfma16h.hip"
