/* the gfx950 code object of asm/fa2_a64_gen.py, embedded as read-only data (see fa2_a64.hip) */
    .section .rodata
    .global fa2_a64_hsaco_start
    .global fa2_a64_hsaco_end
    .hidden fa2_a64_hsaco_start
    .hidden fa2_a64_hsaco_end
    .balign 4096
fa2_a64_hsaco_start:
#ifndef FA2_A64_HSACO
#define FA2_A64_HSACO "fa2_a64.hsaco"
#endif
    .incbin FA2_A64_HSACO
fa2_a64_hsaco_end:
    .byte 0
    .section .note.GNU-stack,"",@progbits
