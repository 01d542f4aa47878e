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
#pragma once
#include <hip/hip_runtime.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace dance {

__device__ __forceinline__ void split_half(float x, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
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
