// Shared device helpers for the pepper_amd gfx950 kernels.
//
// Two families of kernels share these helpers: the f32 ones (gemm.hip, rnn.hip) contract on
// v_mfma_f32_32x32x2_f32 with exact f32 operands (157.3 TFLOP/s dense peak), the default "h2" ones
// (gemm_h2.hip, rnn_h2.hip, mlp_h2.hip) on v_mfma_f32_32x32x16_f16 with every operand split into
// f16 hi/lo halves and three MFMAs per product (f32-level accuracy at 5.3x the rate; format and
// rationale at the top of gemm_h2.hip).  The parity bar is 1e-4 on softmax outputs after 66
// dependent recurrent steps and a K=16896 reduction.
//
// Fragment conventions of the f32 instruction (wave = 64 lanes, lane l):
//   A operand: one f32 = A[i = l & 31][k = l >> 5]
//   B operand: one f32 = B[k = l >> 5][j = l & 31]
//   C/D      : 16 f32,  reg r -> row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col l & 31
// A "k-block" is 8 consecutive k: lane l holds k = 8*kb + 4*(l>>5) + s for s = 0..3 in one
// 16-byte register quad, so 4 MFMAs consume one ds_read_b128 / global_load_dwordx4 per operand.
// For v_mfma_f32_32x32x16_f16: A operand 8 halves = A[i = l & 31][k = 8*(l>>5) + 0..7], B likewise
// with j = l & 31; C/D as above.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define PA_DEV __device__ __forceinline__

PA_DEV f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

PA_DEV int crow32(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

PA_DEV float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Branch-free gate nonlinearities on v_exp_f32 / v_rcp_f32 (both ~1 ulp).  Absolute error is
// ~1e-7 (cancellation in tanh near 0 included), far inside the 1e-4 parity budget; the libm
// expf/tanhf expand to range-split code with exec-mask branches in the recurrent hot loop.
PA_DEV float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
PA_DEV float fast_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// torch.nn.SELU constants (torch/nn/functional.py selu; alpha, scale as published)
PA_DEV float selu_f(float x) {
    const float alpha = 1.6732632423543772848170429916717f;
    const float scale = 1.0507009873554804934193349852946f;
    return scale * (x > 0.0f ? x : alpha * expm1f(x));
}

// One element of a recurrent state -> its word of the h2 row image (even lane: [hi(self), hi(odd neighbour)], odd lane:
// [lo(even neighbour), lo(self)]) in four VALU operations: v_cvt_f16_f32 (hi, round to nearest even), v_fma_mixhi_f16
// (lo = f16(v - hi), one rounding, into the upper half of the same register), a quad-permute DPP move (the neighbour's
// pair) and v_perm_b32 with the lane's byte selector (h2_select).  Same values as the separate cvt / sub / cvt / select
// sequence it replaces, a third of its instructions -- the gate phases are VALU-issue bound.
PA_DEV unsigned h2_select(bool odd) { return odd ? 0x03020706u : 0x05040100u; }
PA_DEV unsigned h2_pair(float v) {       // [f16 hi | f16 lo << 16]
    unsigned p;
    asm("v_cvt_f16_f32 %0, %1\n\tv_fma_mixhi_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]" : "=&v"(p) : "v"(v));
    return p;
}
// Same image written as two 16-bit LDS stores from the lane's own pair (hi at dst, lo 16 bytes on; dst = the lane's
// column in the hi half of its k chunk): no cross-lane exchange, two VALU operations per element instead of four -- the
// LDS port is nearly idle in the gate phases.
PA_DEV void h2_store16(unsigned short* dst, float v) {
    const unsigned p = h2_pair(v);
    dst[0] = (unsigned short)p;
    dst[8] = (unsigned short)(p >> 16);
}
PA_DEV unsigned h2_word_of(float v, unsigned select) {
    const unsigned p = h2_pair(v);
    const unsigned q = (unsigned)__builtin_amdgcn_mov_dpp((int)p, 0xB1, 0xF, 0xF, true);
    return __builtin_amdgcn_perm(q, p, select);
}

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  Unlike
// __syncthreads() it does not drain vmcnt, so global loads / stores issued before it (next-step
// accumulator seeds, y stores) stay in flight across the barrier.
PA_DEV void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Bijective XCD-aware remap of a 1-D grid: workgroup b runs on XCD b % 8 (observed, speed
// only); give every XCD a contiguous run of logical tiles so neighbours share L2 lines.
PA_DEV int xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
