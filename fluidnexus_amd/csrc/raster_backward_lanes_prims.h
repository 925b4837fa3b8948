// Cross-lane primitives of the entries-as-lanes blend backward (raster_backward_lanes.h); tools/micro/lanes_prims.hip checks
// them against their definitions on the device.
#pragma once
#include <hip/hip_runtime.h>

namespace fnx {

#define FNX_SCAN4(op)                                                          \
    "s_nop 1\n\t"                                                             \
    op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                 \
    op " %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"

// Inclusive scans over the 16 lanes of every row, four independent values at a time (Hillis-Steele: shifts 1, 2, 4, 8).
// A lane whose source lies outside its row keeps its value (bound_ctrl off: the lane is disabled for the instruction).
// The four scans are interleaved, so an instruction never reads a register written less than three instructions ago
// (a DPP operand needs two wait states behind the VALU write); s_nop 1 covers the writes in front of the block.
__device__ __forceinline__ void row_scan_mul4(float (&x)[4]) {
    asm(FNX_SCAN4("v_mul_f32_dpp") : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
}
__device__ __forceinline__ void row_scan_add4(float (&x)[4]) {
    asm(FNX_SCAN4("v_add_f32_dpp") : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
}
// value of the previous lane of the row; lane 0 of a row gets `first`
__device__ __forceinline__ float row_prev(float v, float first) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, v),
                                                                  0x111, 0xf, 0xf, false));
}
// lane 15 of every row to all lanes of the row (row_newbcast:15)
__device__ __forceinline__ float row_last(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x15F, 0xf, 0xf, false));
}
// Sums over the four rows of a wave, four values at a time: afterwards row 0 holds the rows' total of a, row 1 of c,
// row 2 of b, row 3 of d (per lane of the row).  v_permlane32_swap exchanges the upper half of its first operand with the
// lower half of its second, v_permlane16_swap the odd rows of the first with the even rows of the second (gfx950).
__device__ __forceinline__ float rows_fold4(float a, float b, float c, float d) {
#ifdef FNX_EXP_FOLD_BUILTIN  // timing experiment, WRONG sums (see below): the builtin form
    const auto ab = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const auto cd = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d), false, false);
    const float s1 = __builtin_bit_cast(float, ab[0]) + __builtin_bit_cast(float, ab[1]);
    const float s2 = __builtin_bit_cast(float, cd[0]) + __builtin_bit_cast(float, cd[1]);
    const auto t = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s2), false, false);
    return __builtin_bit_cast(float, t[0]) + __builtin_bit_cast(float, t[1]);
#endif
    // (written as instructions: with hipcc 7.2 both elements of __builtin_amdgcn_permlane32_swap's result came out as the
    //  FIRST one -- v_add v1, v1, v1 behind the swap, tools/micro/lanes_prims.hip.  A swap reads its operands two wait
    //  states behind a VALU write at the earliest.)
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %1\n\t"   // a: a.r0 a.r1 b.r0 b.r1   b: a.r2 a.r3 b.r2 b.r3
        "v_permlane32_swap_b32 %2, %3\n\t"
        "v_add_f32 %0, %0, %1\n\t"           // a02 a13 b02 b13
        "v_add_f32 %2, %2, %3\n\t"           // c02 c13 d02 d13
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %2\n\t"   // a: a02 c02 b02 d02       c: a13 c13 b13 d13
        "v_add_f32 %0, %0, %2\n\t"           // a c b d
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    return a;
}

}  // namespace fnx
