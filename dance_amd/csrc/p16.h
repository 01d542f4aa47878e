// p16.h -- the "split-half" activation format of the fast precision mode.
//
// An fp32 value x is carried as two fp16 numbers  hi = rn16(x), lo = rn16(x - hi)
// (22 significant bits; hi*hi products are exact in the fp32 accumulator).  A GEMM on such
// operands is three f16 MFMAs  A_lo*B_hi + A_hi*B_lo + A_hi*B_hi  accumulated in fp32:
// fp32-level accuracy (measured: as close to an fp64 evaluation as the fp32 reference is)
// at 16/3 of the fp32-MFMA rate.  Weights are pre-scaled by a power of two so that their lo
// halves stay normal fp16 numbers; the scale is undone exactly in the epilogue.
//
// Layout "P16" of a (C, V) activation block, C a multiple of 128 ("slots" of 128 channels):
//   slot s at  base + s * slot_stride                       (halfs)
//   plane p (0 = hi, 1 = lo) at  + p * 128 * V              (halfs)
//   channel c, vertex v at  + ((c / 8) * V + v) * 8 + c % 8 (halfs)
// i.e. [slot][plane][c/8][v][8]: the 8 consecutive channels one MFMA lane needs for one
// vertex are one 16-byte word, consecutive vertices are consecutive 16-byte words (bank
// conflict free ds_read_b128), and the bytes per element equal fp32's.
//
// Activation scales.  rn16(x - hi) is a NORMAL fp16 number -- the pair carries 22 bits -- only while
// |x| >= 2^-3; below, lo is subnormal (absolute resolution 2^-25) and the pair degrades gracefully;
// above 65504 hi is inf (loud: NaN coordinates).  So every split is done on x * 2^a with a power
// of two that puts the data's largest magnitudes at 2^8 .. 2^15, and 2^-a is undone exactly where
// the fp32 accumulator is read:
//   * the head block's input (gathered features | de_location): per-contour exponents from the
//     image's measured max |feature| (absmax words written by the layout / feature-prep pass)
//     and the contour's de_location range -- head_input_scales below;
//   * the 8 state slots, prediction[0]'s output: per-slot exponents fixed when the net is packed,
//     from its BatchNorm parameters / weight norms (dance_amd/snake.py), folded into the packed
//     constants (bias, BN affine, inverse weight scales; one residual multiplier per block).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace dance {

// hi = rn16(x) of the fp32 VALUE x.  The empty asm pins x to a register first: without it the
// compiler may fold the fp32 operation that produced x into the conversion (v_fma_mixlo_f16 /
// v_fma_mixhi_f16: one rounding of the unrounded result instead of rn16(rn32(.))), for SOME elements
// of a tile and differently in every kernel instantiation -- on a tie the pair then splits the other
// way (same sum, other hi / lo), and kernel forms that must give identical bits no longer do
// (seen: 3 elements of 600 k between the 4- and 8-wave residual stacks, tools/stack_forms_probe.py).
__device__ __forceinline__ void split_half(float x, _Float16 &hi, _Float16 &lo)
{
    asm("" : "+v"(x));
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// The same without the pin, for values that do not come out of an fma / mul the conversion could
// absorb (the gather loops: sums of separately rounded products under fp contract(off)) -- the asm
// statement keeps the compiler from unrolling those loops.
__device__ __forceinline__ void split_half_sum(float x, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// 2^e as a float, e in [-126, 127]
__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

// floor(log2 |x|) from x's bit pattern; `none` for zero, subnormals, inf and NaN
__device__ __forceinline__ int exponent_of(unsigned bits, int none)
{
    const unsigned e = (bits >> 23) & 0xffu;
    return (e == 0u || e == 0xffu) ? none : (int)e - 127;
}

// Exponent of the power of two that puts the largest |feature| of an image into [2^14, 2^15)
// (absmax_img: DANCE_ABSMAX_WORDS partial maxima as float bit patterns).  A bilinear blend never
// exceeds the image's maximum, so nothing overflows.  Clamped to +-40 (the accumulators carry the
// scale too); 40 for an all-zero image (it does not bind), 0 for inf / NaN features (the result is
// NaN either way).
__device__ __forceinline__ int feature_scale_exp(const uint32_t *absmax_img)
{
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        m = max(m, absmax_img[k]);
    if (m >= 0x7f800000u)
        return 0;
    return max(-40, min(40, 14 - exponent_of(m, m ? -126 : -1000)));
}

// Exponent of the scale of a magnitude bound `amax` (a float >= 0): amax * 2^f in [2^14, 2^15);
// 14 when amax is zero, subnormal, inf or NaN.
__device__ __forceinline__ int bound_scale_exp(float amax)
{
    return max(-40, min(40, 14 - exponent_of(__builtin_bit_cast(unsigned, amax), 0)));
}

// 2^e, or 0 when the factor (and with it the whole term it scales) lies more than 2^120 below its
// partner's: that term's contribution is then far below the fp32 resolution of the sum
__device__ __forceinline__ float pow2f_or_zero(int e) { return e < -120 ? 0.0f : pow2f(min(e, 120)); }

// The head block's input has two groups of channels of unrelated magnitude -- gathered features and
// the two de_location channels -- multiplied by weight columns of unrelated magnitude as well
// (packed separately: head_w at 2^ew, head_wd at 2^ewd).  Both groups accumulate into the same fp32
// tile, so the products must share one exponent E = (weight exponent) + (operand exponent).  Each
// group has an ideal operand exponent that puts its largest magnitude into [2^14, 2^15) (f_feat: from
// the image's measured max |feature|; f_deloc: from the contour's de_location range); E is the
// smaller of the two sums -- the group that contributes MORE to the sum in true units keeps its ideal
// scale, the other one sits lower by exactly the factor its contribution is smaller.
struct HeadInputScales {
    float fs;       // factor of the gathered features (folded into the bilinear tap weights)
    float ds;       // factor of the de_location values
    float inv;      // 2^-E: accumulator -> true units (and: bias * 2^ew * fs = bias * 2^E)
};
// winv / wdinv: the inverse weight scales 2^-ew, 2^-ewd (powers of two).  scaled == false: the features
// stay as they are (fs = 1), the de_location values only make up for the two weight scales.
__device__ __forceinline__ HeadInputScales head_input_scales(float winv, float wdinv, bool scaled, int f_feat,
                                                             bool has_deloc, int f_deloc)
{
    const int ew = -exponent_of(__builtin_bit_cast(unsigned, winv), 0);
    const int ewd = -exponent_of(__builtin_bit_cast(unsigned, wdinv), 0);
    int e = ew + (scaled ? f_feat : 0);
    if (scaled && has_deloc)
        e = min(e, ewd + f_deloc);
    HeadInputScales s;
    s.fs = pow2f_or_zero(e - ew);
    s.ds = has_deloc ? pow2f_or_zero(e - ewd) : 0.0f;
    s.inv = pow2f(max(-120, min(120, -e)));
    return s;
}

// Store the 4 consecutive-channel values one lane holds for one vertex (C/D tile: lane =
// vertex column, register group g = 4 rows) as hi/lo 8-byte words of a P16 slot.
//   slot : base of the 128-channel slot (halfs);  cb : channel block inside the slot (0..15)
//   half4: 0/1 = which 4 of the block's 8 channels (lane >> 5)
__device__ __forceinline__ void store_p16(_Float16 *slot, int v_total, int cb, int v, int half4,
                                          const float (&val)[4])
{
    h4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        _Float16 a, b;
        split_half(val[i], a, b);
        hi[i] = a;
        lo[i] = b;
    }
    _Float16 *p = slot + ((long long)cb * v_total + v) * 8 + half4 * 4;
    *reinterpret_cast<h4 *>(p) = hi;
    *reinterpret_cast<h4 *>(p + 128ll * v_total) = lo;
}

}  // namespace dance
