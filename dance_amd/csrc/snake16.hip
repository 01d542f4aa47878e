// snake16.hip -- the snake blocks in the fast precision mode: split-half (P16) activations and
// weights, three f16 MFMAs per product term (see p16.h), fp32 accumulation.
//
// Same decomposition as snake_mfma.hip (one workgroup = one contour, a wave = 32-channel row
// tiles x all vertex tiles, weights from L2 as the A operand, the contour's LDS image as the
// B operand, ring taps as index rotations), with v_mfma_f32_32x32x16_f16:
//   A lane (r = lane & 31, h = lane >> 5): W[o0 + r][channels 8h .. 8h+7]  = one 16-B word of
//     the packed weights [plane][tap][c/8][o][8];
//   B lane: X[channels 8h .. 8h+7][vertex r] = one 16-B word of the P16 LDS image
//     [plane][c/8][VP][8]  ->  ds_read_b128, consecutive lanes on consecutive words.
//
// At f16 MFMA rates (32 cycles per 32x32x16) the vector ALU, not the matrix pipe, is what a
// naive loop saturates: a wave64 VALU instruction costs 4 cycles, so a 12-MFMA step has room
// for a few dozen of them per SIMD.  The inner loop (mfma16_core) is therefore built so that a
// step issues only MFMAs, LDS reads with immediate offsets (compile-time row stride VP),
// weight loads from a scalar base + a constant lane offset, and scalar pointer bumps; ring
// positions advance incrementally once per tap.  Activations of the next chunk are staged
// global -> registers one load per step, right AFTER the step's weight prefetch, so the
// weight wait (in-order vmcnt) never waits for an activation load issued before it.
//
// Kernels: k_conv16 (residual SnakeBlock / k=1 Conv1d), k_fuse16 (fusion + max over the ring
// + row constant + prediction[0], the 1024-channel state read once), k_gather_head16 (gather
// + de_location + head block).
#ifndef DANCE_FUSE_DMA
// 1: k_fuse16 / k_fuse16r stage their chunk images by LDS-direct loads (stage_dma, inline asm with
// hand-placed waits) instead of registers + ds_write.  Built, bit-identical results (127 GPU
// tests), measured same box twice: fusion kernel 0.611 / 0.613 ms through registers, 0.622 /
// 0.624 ms with the DMA (1600 contours x 128 vertices) -- the staging registers and ds_writes
// are not what the K loop waits for.  Off by default; -DDANCE_FUSE_DMA=1 builds it.
#define DANCE_FUSE_DMA 0
#endif
#include "common.h"
#include "mfma16.h"
#include "p16.h"
#include "sampling.h"

using dance::launch_check;

namespace {

struct Conv16Args {
    const _Float16 *x;          // P16 input
    long long x_stride;         // halfs per contour
    long long x_slot_stride;    // halfs per 128-channel slot
    const h8 *w;                // [plane][tap][cin/8][cout] words
    const float *bias_s;        // bias * 2^s (accumulator start)
    const float *scale, *shift; // folded BatchNorm (or null)
    float winv;                 // 2^-s
    _Float16 *y;                // P16 output (or null)
    long long y_stride, y_slot_stride;
    float *yf;                  // fp32 channel-major output (or null)
    long long yf_stride;
    const float *rowinit;       // per-contour accumulator start (n, cout), unscaled, or null
    int v, cin, cout, taps, dilation, relu, residual, maxv;
};

// Residual SnakeBlock / k=1 Conv1d.  4 waves, one 32-channel tile each (128 outputs per
// workgroup, grid.y tiles of 128); chunks of 64 (V <= 128) or 32 channels double-buffered in
// LDS: while chunk k is multiplied, chunk k+1 -- of this contour or, when the grid is smaller
// than n, of the workgroup's next contour -- is staged into registers and committed to the
// other buffer at the chunk boundary.  With two chunks per contour (128 channels) both stay
// resident, and the residual is read back from LDS in the epilogue.
template <int RT, int TERMS>
__global__ __launch_bounds__(256, 2) void k_conv16(Conv16Args a, int n)
{
    constexpr int CBC = RT <= 4 ? 8 : 4, NST = CBC / 2, AD = NST % 4 == 0 ? 4 : 2, NW = 4;
    constexpr int NB = RT <= 4 ? 4 : 2;
    using G = Geo<RT, CBC>;
    using S = StageGeo<RT, CBC, NW>;
    extern __shared__ __attribute__((aligned(16))) char xs[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int o0 = blockIdx.y * 128 + wave * 32;
    const bool active = o0 < a.cout;
    const int v = a.v;
    const long long wbs_b = (long long)a.cout * 16, wts_b = (long long)(a.cin >> 3) * wbs_b,
                    wps_b = (long long)a.taps * wts_b;
    const unsigned wlane = (unsigned)(h * a.cout + r) * 16u;
    const int nch = a.cin / (CBC * 8);
    const bool res_lds = a.residual && nch <= 2;
    const int ostep = gridDim.x;

    // byte pointer to (plane 0, first block) of chunk c of contour m
    auto chunk_src = [&](long long m, int c) __attribute__((always_inline)) {
        const int c0 = c * CBC * 8;
        return reinterpret_cast<const char *>(a.x + m * a.x_stride + (c0 >> 7) * a.x_slot_stride) +
               (long long)((c0 & 127) >> 3) * v * 16;
    };

    u32x4 sreg[S::N];
    long long nn = blockIdx.x;
    stage_load<RT, CBC, NW>(sreg, 0, S::N, chunk_src(nn, 0), v, wave, lane);
    stage_commit<RT, CBC, NW>(sreg, xs, 0, v, wave, lane);
    __syncthreads();

    int k = 0;
    for (; nn < n; nn += ostep) {
        f32x16 acc[1][RT];
        if (active) {
            // accumulators carry 2^s-scaled sums; a per-contour row start is scaled to match
            if (a.rowinit)
                init_acc16<RT, 1>(acc, a.rowinit + nn * a.cout, o0, h, 1.0f / a.winv);
            else
                init_acc16<RT, 1>(acc, a.bias_s, o0, h, 1.0f);
        }
        for (int c = 0; c < nch; ++c, ++k) {
            const unsigned cur = (k & 1) * G::BYTES, nxt = ((k + 1) & 1) * G::BYTES;
            const bool last = c + 1 == nch;
            const bool has_next = !last || nn + ostep < n;
            // nothing left to stage: re-read the current chunk (valid memory), never committed
            const char *ssrc = has_next ? chunk_src(last ? nn + ostep : nn, last ? 0 : c + 1) : chunk_src(nn, c);
            if (active) {
                auto hook = [&](int s) __attribute__((always_inline)) {
                    stage_load<RT, CBC, NW>(sreg, s * S::N / NST, (s + 1) * S::N / NST, ssrc, v, wave, lane);
                };
                mfma16_core<RT, 1, NST, AD, NB, TERMS>(acc, xs, cur, v,
                                            reinterpret_cast<const char *>(a.w) + ((long long)(c * CBC) * a.cout + o0) * 16,
                                            wlane, wps_b, wts_b, 2 * wbs_b, a.taps, a.dilation, r, h, hook);
            } else {
                stage_load<RT, CBC, NW>(sreg, 0, S::N, ssrc, v, wave, lane);
            }
            if (last && active) {
                // epilogue: lane = vertex column, register group g = channels 8g+4h .. +3.
                // (re, he) are r, h made opaque so that the compiler does not hoist the
                // epilogue's addresses and per-channel constants above the MFMA chunks,
                // where they would be live across them and spill.
                int re = r, he = h;
                asm volatile("" : "+v"(re), "+v"(he));
                _Float16 *yslot = a.y ? a.y + nn * a.y_stride + (o0 >> 7) * a.y_slot_stride : nullptr;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ob = o0 + 8 * g + 4 * he;
                    float4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
                    if (a.scale) {
                        sc4 = *reinterpret_cast<const float4 *>(a.scale + ob);
                        sh4 = *reinterpret_cast<const float4 *>(a.shift + ob);
                    }
                    const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
                    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    // residual source: input rows (o0 / 8 + g): LDS image (chunk = row / CBC) or global
                    const int xrow = (o0 >> 3) + g;
                    const int xchunk = xrow / CBC, xcbl = xrow - xchunk * CBC;
                    const unsigned rbuf = ((k - (nch - 1) + xchunk) & 1) * G::BYTES;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int vv = rt * 32 + re;
                        if (vv >= v)
                            continue;
                        float val[4];
                        h4 rh = {0, 0, 0, 0}, rl = {0, 0, 0, 0};
                        if (a.residual) {
                            if (res_lds) {
                                const char *q = xs + rbuf + (xcbl * G::VP + vv) * 16 + 8 * he;
                                rh = *reinterpret_cast<const h4 *>(q);
                                rl = *reinterpret_cast<const h4 *>(q + G::PLANE);
                            } else {
                                const _Float16 *q = a.x + nn * a.x_stride + ((long long)xrow * v + vv) * 8 + 4 * he;
                                rh = *reinterpret_cast<const h4 *>(q);
                                rl = *reinterpret_cast<const h4 *>(q + 128ll * v);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float x = acc[0][rt][4 * g + i] * a.winv;
                            if (a.relu)
                                x = relu_nan(x);
                            if (a.scale)
                                x = x * sc[i] + sh[i];
                            if (a.residual)
                                x += (float)rh[i] + (float)rl[i];
                            val[i] = x;
                        }
                        if (a.maxv) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                mx[i] = dance::nan_max(val[i], mx[i]);
                        } else if (yslot) {
                            dance::store_p16(yslot, v, ((o0 & 127) >> 3) + g, vv, he, val);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                a.yf[nn * a.yf_stride + (long long)(ob + i) * v + vv] = val[i];
                        }
                    }
                    if (a.maxv) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float m = mx[i];
#pragma unroll
                            for (int s = 16; s >= 1; s >>= 1)
                                m = dance::nan_max(__shfl_xor(m, s, 64), m);
                            if (re == 0)
                                a.yf[nn * a.cout + ob + i] = m;
                        }
                    }
                }
            }
            __syncthreads();                    // every wave is done with chunk k (and the residual)
            if (has_next)
                stage_commit<RT, CBC, NW>(sreg, xs, nxt, v, wave, lane);
            __syncthreads();
        }
    }
}

// k_conv16 on 16x16x32 MFMAs (V <= 128).  4 waves, one 32-channel tile each (128 outputs per
// workgroup, grid.y tiles of 128); chunks of 64 (V <= 128) or 32 channels double-buffered in
// LDS: while chunk k is multiplied, chunk k+1 -- of this contour or, when the grid is smaller
// than n, of the workgroup's next contour -- is staged into registers and committed to the
// other buffer at the chunk boundary.  With two chunks per contour (128 channels) both stay
// resident, and the residual is read back from LDS in the epilogue.
template <int RT, int TERMS>
__global__ __launch_bounds__(256, 2) void k_conv16s(Conv16Args a, int n)
{
    static_assert(RT <= 4, "larger rings use k_conv16");
    constexpr int CBC = 8, NST = CBC / 4, AD = 2, NW = 4, NB = 4, CT = 2 * RT;
    using G = Geo<RT, CBC>;
    using S = StageGeo<RT, CBC, NW>;
    extern __shared__ __attribute__((aligned(16))) char xs[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const int o0 = blockIdx.y * 128 + wave * 32;
    const bool active = o0 < a.cout;
    const int v = a.v;
    const long long wbs_b = (long long)a.cout * 16, wts_b = (long long)(a.cin >> 3) * wbs_b,
                    wps_b = (long long)a.taps * wts_b;
    const unsigned wlane = (unsigned)(q4 * a.cout + c16) * 16u;
    const int nch = a.cin / (CBC * 8);
    const bool res_lds = a.residual && nch <= 2;
    const int ostep = gridDim.x;

    // byte pointer to (plane 0, first block) of chunk c of contour m
    auto chunk_src = [&](long long m, int c) __attribute__((always_inline)) {
        const int c0 = c * CBC * 8;
        return reinterpret_cast<const char *>(a.x + m * a.x_stride + (c0 >> 7) * a.x_slot_stride) +
               (long long)((c0 & 127) >> 3) * v * 16;
    };

    u32x4 sreg[S::N];
    long long nn = blockIdx.x;
    stage_load<RT, CBC, NW>(sreg, 0, S::N, chunk_src(nn, 0), v, wave, lane);
    stage_commit<RT, CBC, NW>(sreg, xs, 0, v, wave, lane);
    __syncthreads();

    int k = 0;
    for (; nn < n; nn += ostep) {
        f32x4 acc[2][CT];
        if (active) {
            // accumulators carry 2^s-scaled sums; a per-contour row start is scaled to match
            const float *bsrc = a.rowinit ? a.rowinit + nn * a.cout : a.bias_s;
            const float mul = a.rowinit ? 1.0f / a.winv : 1.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 b4 = *reinterpret_cast<const float4 *>(bsrc + o0 + 16 * i + 4 * q4);
                const f32x4 bv = {b4.x * mul, b4.y * mul, b4.z * mul, b4.w * mul};
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    acc[i][j] = bv;
            }
        }
        for (int c = 0; c < nch; ++c, ++k) {
            const unsigned cur = (k & 1) * G::BYTES, nxt = ((k + 1) & 1) * G::BYTES;
            const bool last = c + 1 == nch;
            const bool has_next = !last || nn + ostep < n;
            // nothing left to stage: re-read the current chunk (valid memory), never committed
            const char *ssrc = has_next ? chunk_src(last ? nn + ostep : nn, last ? 0 : c + 1) : chunk_src(nn, c);
            if (active) {
                auto hook = [&](int s) __attribute__((always_inline)) {
                    stage_load<RT, CBC, NW>(sreg, s * S::N / NST, (s + 1) * S::N / NST, ssrc, v, wave, lane);
                };
                mfma16_core_s<RT, NST, AD, NB, TERMS>(acc, xs, cur, v,
                                               reinterpret_cast<const char *>(a.w) + ((long long)(c * CBC) * a.cout + o0) * 16,
                                               wlane, wps_b, wts_b, 4 * wbs_b, a.taps, a.dilation, c16, q4, hook);
            } else {
                stage_load<RT, CBC, NW>(sreg, 0, S::N, ssrc, v, wave, lane);
            }
            if (last && active) {
                // epilogue: lane = vertex column 16 j + c16, row tile i = channels 16 i + 4 q4 .. + 3.
                // (ce, qe) are made opaque so that the compiler does not hoist the epilogue's
                // addresses and per-channel constants above the MFMA chunks.
                int ce = c16, qe = q4;
                asm volatile("" : "+v"(ce), "+v"(qe));
                _Float16 *yslot = a.y ? a.y + nn * a.y_stride + (o0 >> 7) * a.y_slot_stride : nullptr;
#pragma unroll
                for (int i2 = 0; i2 < 2; ++i2) {
                    const int ob = o0 + 16 * i2 + 4 * qe;
                    float4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
                    if (a.scale) {
                        sc4 = *reinterpret_cast<const float4 *>(a.scale + ob);
                        sh4 = *reinterpret_cast<const float4 *>(a.shift + ob);
                    }
                    const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
                    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    // residual source: input row (ob / 8), half (ob / 4) & 1: LDS image or global
                    const int xrow = ob >> 3, xhalf = (ob >> 2) & 1;
                    const int xchunk = xrow / CBC, xcbl = xrow - xchunk * CBC;
                    const unsigned rbuf = ((k - (nch - 1) + xchunk) & 1) * G::BYTES;
#pragma unroll
                    for (int j = 0; j < CT; ++j) {
                        const int vv = j * 16 + ce;
                        if (vv >= v)
                            continue;
                        float val[4];
                        h4 rh = {0, 0, 0, 0}, rl = {0, 0, 0, 0};
                        if (a.residual) {
                            if (res_lds) {
                                const char *q = xs + rbuf + (xcbl * G::VP + vv) * 16 + 8 * xhalf;
                                rh = *reinterpret_cast<const h4 *>(q);
                                rl = *reinterpret_cast<const h4 *>(q + G::PLANE);
                            } else {
                                const _Float16 *q = a.x + nn * a.x_stride + ((long long)xrow * v + vv) * 8 + 4 * xhalf;
                                rh = *reinterpret_cast<const h4 *>(q);
                                rl = *reinterpret_cast<const h4 *>(q + 128ll * v);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float x = acc[i2][j][i] * a.winv;
                            if (a.relu)
                                x = relu_nan(x);
                            if (a.scale)
                                x = x * sc[i] + sh[i];
                            if (a.residual)
                                x += (float)rh[i] + (float)rl[i];
                            val[i] = x;
                        }
                        if (a.maxv) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                mx[i] = dance::nan_max(val[i], mx[i]);
                        } else if (yslot) {
                            dance::store_p16(yslot, v, (ob & 127) >> 3, vv, xhalf, val);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                a.yf[nn * a.yf_stride + (long long)(ob + i) * v + vv] = val[i];
                        }
                    }
                    if (a.maxv) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float m = mx[i];
#pragma unroll
                            for (int s = 8; s >= 1; s >>= 1)
                                m = dance::nan_max(__shfl_xor(m, s, 64), m);
                            if (ce == 0)
                                a.yf[nn * a.cout + ob + i] = m;
                        }
                    }
                }
            }
            __syncthreads();                    // every wave is done with chunk k (and the residual)
            if (has_next)
                stage_commit<RT, CBC, NW>(sreg, xs, nxt, v, wave, lane);
            __syncthreads();
        }
    }
}

// All residual SnakeBlocks of one _SnakeNet in ONE launch (V <= 128): a workgroup takes a
// contour through the 7 blocks with the activations resident in LDS.  Block l reads the image,
// every wave finishes its MFMAs (barrier), then the epilogue adds the residual from the image,
// writes block l's output to its state slot in HBM (fusion / prediction read it later) AND
// back into the same LDS words as block l + 1's input (barrier).  Against 7 launches of
// k_conv16s: one launch gap and one kernel tail instead of 7, no activation re-read, no chunk
// staging inside the MFMA loops.
struct Stack16Args {
    _Float16 *state;                // (n, 1 + nlayers slots) P16: slot l = input of block l
    long long stride, slot;         // halfs per contour / per slot
    const h8 *w[7];
    const float *bias_s[7], *scale[7], *shift[7];
    float winv[7];
    int dilation[7];
    int v, nlayers;
};

// The layers of the stack for one contour whose 128-channel input image already sits in LDS
// (xs: chunk images 0 | 1) behind a barrier.
// CT: 16-wide vertex tiles computed (2 * RT = the whole image width; fewer when the ring is
// shorter: 33..48 vertices = 3 tiles in the 64-column image)
template <int RT, int TERMS, int CT = 2 * RT>
__device__ __forceinline__ void res_stack16_layers(const Stack16Args &a, long long nn, char *xs)
{
    static_assert(RT <= 4 && CT * 16 <= RT * 32, "rings of at most 128 vertices, the tiles inside the image");
    constexpr int CBC = 8, NST = CBC / 4, AD = 2, NB = (NST * CT) % 4 == 0 ? 4 : 2;
    using G = Geo<RT, CBC>;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const int o0 = wave * 32;
    const int v = a.v;
    const long long wbs_b = 128 * 16, wts_b = 16 * wbs_b, wps_b = 9 * wts_b;
    const unsigned wlane = (unsigned)(q4 * 128 + c16) * 16u;
    auto nohook = [](int) __attribute__((always_inline)) {};
    _Float16 *sn = a.state + nn * a.stride;
    for (int l = 0; l < a.nlayers; ++l) {
        f32x4 acc[2][CT];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 b4 = *reinterpret_cast<const float4 *>(a.bias_s[l] + o0 + 16 * i + 4 * q4);
            const f32x4 bv = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < CT; ++j)
                acc[i][j] = bv;
        }
        const char *wl = reinterpret_cast<const char *>(a.w[l]) + (long long)o0 * 16;
        const int dil = a.dilation[l];
#pragma unroll 1
        for (int c = 0; c < 2; ++c)
            mfma16_core_sw<RT, CT, NST, AD, NB, TERMS>(acc, xs, c * G::BYTES, v, wl + (long long)c * CBC * wbs_b, wlane,
                                                       wps_b, wts_b, 4 * wbs_b, 9, dil, c16, q4, 0, nohook);
        __syncthreads();                    // every wave has read the whole image
        // epilogue: lane = vertex 16 j + c16, row tile i = channels 16 i + 4 q4 .. + 3; indices
        // made opaque so the compiler does not hoist the address math over the MFMAs
        int ce = c16, qe = q4;
        asm volatile("" : "+v"(ce), "+v"(qe));
        _Float16 *yslot = sn + (l + 1) * a.slot;
        const float winv = a.winv[l];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int ob = o0 + 16 * i2 + 4 * qe;
            const float4 sc4 = *reinterpret_cast<const float4 *>(a.scale[l] + ob);
            const float4 sh4 = *reinterpret_cast<const float4 *>(a.shift[l] + ob);
            const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            const int xrow = ob >> 3, xhalf = (ob >> 2) & 1;
            const int xchunk = xrow / CBC, xcbl = xrow - xchunk * CBC;
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int vv = j * 16 + ce;
                if (vv >= v)
                    continue;
                char *q = xs + xchunk * G::BYTES + (xcbl * G::VP + vv) * 16 + 8 * xhalf;
                const h4 rh = *reinterpret_cast<const h4 *>(q);
                const h4 rl = *reinterpret_cast<const h4 *>(q + G::PLANE);
                h4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = relu_nan(acc[i2][j][i] * winv) * sc[i] + sh[i];
                    x += (float)rh[i] + (float)rl[i];
                    _Float16 xh, xl;
                    dance::split_half(x, xh, xl);
                    hi[i] = xh;
                    lo[i] = xl;
                }
                // next block's input, in place (this lane owns these two half-words)
                *reinterpret_cast<h4 *>(q) = hi;
                *reinterpret_cast<h4 *>(q + G::PLANE) = lo;
                // written once, read once by the fusion kernel: non-temporal, the L2 keeps the weights
                _Float16 *d = yslot + ((long long)xrow * v + vv) * 8 + 4 * xhalf;
                __builtin_nontemporal_store(hi, reinterpret_cast<h4 *>(d));
                __builtin_nontemporal_store(lo, reinterpret_cast<h4 *>(d + 128ll * v));
            }
        }
        __syncthreads();                    // image = block l + 1's input
    }
}

template <int RT, int TERMS, int CT = 2 * RT>
__global__ __launch_bounds__(256, 2) void k_res_stack16s(Stack16Args a, int n)
{
    constexpr int CBC = 8, NW = 4;
    using G = Geo<RT, CBC>;
    using S = StageGeo<RT, CBC, NW>;
    extern __shared__ __attribute__((aligned(16))) char xs[];   // chunk images 0 | 1 = 128 channels
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int v = a.v;
    for (long long nn = blockIdx.x; nn < n; nn += gridDim.x) {
        const _Float16 *sn = a.state + nn * a.stride;
        {
            u32x4 sreg[S::N];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                stage_load<RT, CBC, NW>(sreg, 0, S::N, reinterpret_cast<const char *>(sn) + (long long)c * CBC * v * 16,
                                        v, wave, lane);
                stage_commit<RT, CBC, NW>(sreg, xs, c * G::BYTES, v, wave, lane);
            }
        }
        __syncthreads();
        res_stack16_layers<RT, TERMS, CT>(a, nn, xs);
    }
}

// The residual stack for rings of 129..256 vertices: 8 waves = 4 output-channel tiles x 2 halves
// of the ring's 16-wide vertex tiles, the 128-channel image (115 / 132 KB) resident in LDS, one
// workgroup per CU.  Same arithmetic per output element as k_conv16 / k_conv16s.
// A wave's share of the layers: CTW vertex tiles starting at tile j0 (the image sits in LDS behind
// a barrier; every wave of the workgroup runs this with the same number of barriers).
template <int RT, int CTW, int TERMS>
__device__ __forceinline__ void res_stack16w_layers(const Stack16Args &a, _Float16 *sn, char *xs, int o0, int j0,
                                                    int c16, int q4)
{
    constexpr int CBC = 8, NST = CBC / 4, AD = 2, NB = (NST * CTW) % 4 == 0 ? 4 : 2;
    using G = Geo<RT, CBC>;
    const int v = a.v;
    const long long wbs_b = 128 * 16, wts_b = 16 * wbs_b, wps_b = 9 * wts_b;
    const unsigned wlane = (unsigned)(q4 * 128 + c16) * 16u;
    auto nohook = [](int) __attribute__((always_inline)) {};
    for (int l = 0; l < a.nlayers; ++l) {
        f32x4 acc[2][CTW];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 b4 = *reinterpret_cast<const float4 *>(a.bias_s[l] + o0 + 16 * i + 4 * q4);
            const f32x4 bv = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < CTW; ++j)
                acc[i][j] = bv;
        }
        const char *wl = reinterpret_cast<const char *>(a.w[l]) + (long long)o0 * 16;
        const int dil = a.dilation[l];
#pragma unroll 1
        for (int c = 0; c < 2; ++c)
            mfma16_core_sw<RT, CTW, NST, AD, NB, TERMS>(acc, xs, c * G::BYTES, v, wl + (long long)c * CBC * wbs_b,
                                                        wlane, wps_b, wts_b, 4 * wbs_b, 9, dil, c16, q4, j0, nohook);
        __syncthreads();                    // every wave has read the whole image
        int ce = c16, qe = q4;
        asm volatile("" : "+v"(ce), "+v"(qe));
        _Float16 *yslot = sn + (l + 1) * a.slot;
        const float winv = a.winv[l];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int ob = o0 + 16 * i2 + 4 * qe;
            const float4 sc4 = *reinterpret_cast<const float4 *>(a.scale[l] + ob);
            const float4 sh4 = *reinterpret_cast<const float4 *>(a.shift[l] + ob);
            const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            const int xrow = ob >> 3, xhalf = (ob >> 2) & 1;
            const int xchunk = xrow / CBC, xcbl = xrow - xchunk * CBC;
#pragma unroll
            for (int j = 0; j < CTW; ++j) {
                const int vv = (j0 + j) * 16 + ce;
                if (vv >= v)
                    continue;
                char *q = xs + xchunk * G::BYTES + (xcbl * G::VP + vv) * 16 + 8 * xhalf;
                const h4 rh = *reinterpret_cast<const h4 *>(q);
                const h4 rl = *reinterpret_cast<const h4 *>(q + G::PLANE);
                h4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = relu_nan(acc[i2][j][i] * winv) * sc[i] + sh[i];
                    x += (float)rh[i] + (float)rl[i];
                    _Float16 xh, xl;
                    dance::split_half(x, xh, xl);
                    hi[i] = xh;
                    lo[i] = xl;
                }
                *reinterpret_cast<h4 *>(q) = hi;
                *reinterpret_cast<h4 *>(q + G::PLANE) = lo;
                // written once, read once by the fusion kernel: non-temporal, the L2 keeps the weights
                _Float16 *d = yslot + ((long long)xrow * v + vv) * 8 + 4 * xhalf;
                __builtin_nontemporal_store(hi, reinterpret_cast<h4 *>(d));
                __builtin_nontemporal_store(lo, reinterpret_cast<h4 *>(d + 128ll * v));
            }
        }
        __syncthreads();                    // image = block l + 1's input
    }
}

// CTB: vertex tiles of the second group of waves (RT for an even split of 2 * RT tiles; RT - 1
// when the ring needs an odd number of tiles: 196 vertices = 13 tiles = 7 + 6, and since waves w
// and w + 4 share a SIMD every SIMD still carries 13 tiles' worth of MFMAs)
template <int RT, int CTB, int TERMS>
__device__ __forceinline__ void res_stack16w_body(const Stack16Args &a, int n)
{
    static_assert(RT >= 4 && RT <= 8, "rings of 97..128 (few contours) or 129..256 vertices");
    static_assert(CTB == RT || CTB == RT - 1, "the second group of waves takes as many tiles or one fewer");
    constexpr int CBC = 8, NW = 8;
    using G = Geo<RT, CBC>;
    using S = StageGeo<RT, CBC, NW>;
    extern __shared__ __attribute__((aligned(16))) char xs[];   // chunk images 0 | 1 = 128 channels
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const int o0 = (wave & 3) * 32, half = wave >> 2;
    const int v = a.v;

    for (long long nn = blockIdx.x; nn < n; nn += gridDim.x) {
        _Float16 *sn = a.state + nn * a.stride;
        {
            u32x4 sreg[S::N];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                stage_load<RT, CBC, NW>(sreg, 0, S::N, reinterpret_cast<const char *>(sn) + (long long)c * CBC * v * 16,
                                        v, wave, lane);
                stage_commit<RT, CBC, NW>(sreg, xs, c * G::BYTES, v, wave, lane);
            }
        }
        __syncthreads();
        if (CTB != RT && half == 1)
            res_stack16w_layers<RT, CTB, TERMS>(a, sn, xs, o0, RT, c16, q4);
        else
            res_stack16w_layers<RT, RT, TERMS>(a, sn, xs, o0, half * RT, c16, q4);
    }
}

template <int RT, int TERMS>
__global__ __launch_bounds__(512) void k_res_stack16w(Stack16Args a, int n)
{
    res_stack16w_body<RT, RT, TERMS>(a, n);
}

// an odd number of tiles: 13 (193..208 vertices, the shipped yamls' 196) as 7 + 6, 9 as 5 + 4, 11 as 6 + 5
template <int RT, int TERMS>
__global__ __launch_bounds__(512) void k_res_stack16w13(Stack16Args a, int n)
{
    res_stack16w_body<RT, RT - 1, TERMS>(a, n);
}

// fusion (1024 -> 256, max over the ring) + row constant + prediction[0] (1280 -> 256, ReLU)
// in one pass over the 1024-channel state: 8 waves, two 32-channel tiles each; waves 0-3 own
// the fusion outputs, waves 4-7 the prediction outputs (weights packed as one 512-output GEMM).
// Chunks of 128 channels, double-buffered like k_conv16 (one staged load per step).
//
// Two forms of the same K loop:
//   fused (k_fuse16, rings of at most 128 vertices): max -> row constant (a 256 x 256 GEMV per
//     contour) -> relu(prediction[0]) inside the kernel; the fusion and the prediction waves take
//     turns waiting for each other at the end.
//   raw (k_fuse16r, any ring up to 256 vertices): both k = 1 convolutions are pointwise over
//     the ring, so the ring may be split into PARTS of at most 128 vertices (blockIdx.y), each
//     a workgroup of its own.  A part writes the max of the fusion output over ITS vertices
//     (gpart) and prediction[0]'s state part WITHOUT row constant and ReLU; k_tail16 (tail16.hip)
//     reduces the parts, forms the row constant and goes on with relu -> prediction[2] ->
//     prediction[4] -> offset update.  No cross-wave hand-over at the end of this kernel.
struct Fuse16Args {
    const _Float16 *x;
    long long x_stride, x_slot_stride;
    const h8 *w;                  // [plane][1][128][512]
    const float *bias_s;          // (512): fusion bias * 2^s, then zeros
    const float *wg;              // (256, 256) [g][o] fp32: global-state columns of prediction[0] (fused form)
    const float *pb;              // (256) prediction[0] bias (fused form)
    float winv;
    _Float16 *y;                  // P16, 2 slots
    long long y_stride, y_slot_stride;
    int v;                        // ring length = vertices per row of x and y
    float *gpart;                 // raw form: (n, parts, 256) max of the fusion output over each part
    int part_v;                   // raw form: vertices of every part but the last
};

// v0 / v: first vertex and vertex count of this workgroup's part of the ring (0 / a.v: whole ring)
// NW = 8: one 8-wave workgroup per contour part and CU (all 512 outputs, 128-channel chunks).
// NW = 4 (raw form only): the 512 outputs as two 4-wave workgroups (blockIdx.z = fusion half /
// prediction half, 64-channel chunks, two workgroups per CU): each reads the state part itself,
// but the two barrier domains are independent -- one's chunk barriers, prologue and tail overlap
// the other's MFMAs.
template <int RT, int TERMS, bool RAW, int NW = 8>
__device__ __forceinline__ void fuse16_body(const Fuse16Args &a, int n, int v0, int v, int part, int parts)
{
    static_assert(NW == 8 || (NW == 4 && RAW), "the 4-wave split exists in the raw form only");
    // K = 1024 as one flat sequence of 64 steps of 16 channels: weights rotate through two
    // buffers over the whole sequence (no refill bubble at chunk boundaries), activations come
    // from 128-channel chunk images double-buffered in LDS, one staged load per step, ONE
    // barrier per chunk (the other image was last read two chunks ago).
    constexpr int CBC = 2 * NW, NST = CBC / 2, NCH = 1024 / (CBC * 8), NT = NST * RT;
    using G = Geo<RT, CBC>;
    using S = StageGeo<RT, CBC, NW>;
    constexpr int SSTEP = 2 * G::VP * 16;
    extern __shared__ __attribute__((aligned(16))) char xs[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int o0 = wave * 64 + (NW == 4 ? (int)blockIdx.z * 256 : 0);
    const int vs = a.v;
    const long long wbs_b = 512 * 16, wps_b = 128 * wbs_b, wstep_b = 2 * wbs_b;
    const unsigned wlane = (unsigned)(h * 512 + r) * 16u;
    const int ostep = gridDim.x;
    const char *wbase = reinterpret_cast<const char *>(a.w) + (long long)o0 * 16;

    auto chunk_src = [&](long long m, int c) __attribute__((always_inline)) {
        const int cb0 = c * CBC;                // first channel block of the chunk: slot cb0 / 16, block cb0 % 16
        return reinterpret_cast<const char *>(a.x + m * a.x_stride + (cb0 >> 4) * a.x_slot_stride) +
               (long long)(cb0 & 15) * vs * 16 + v0 * 16;
    };

    long long nn = blockIdx.x;
    // the 4-wave form stages by LDS-direct loads: it has no registers to spare for staging
    constexpr bool DMA = DANCE_FUSE_DMA || NW == 4;
    const unsigned lds_base = __builtin_amdgcn_groupstaticsize();   // the dynamic segment follows the static one
    u32x4 sreg[DMA ? 1 : S::N];
    if constexpr (DMA) {
        stage_dma<RT, CBC, NW>(0, S::N, chunk_src(nn, 0), lds_base, v, wave, lane, vs);
        stage_dma_wait();
    } else {
        stage_load<RT, CBC, NW>(sreg, 0, S::N, chunk_src(nn, 0), v, wave, lane, vs);
        stage_commit<RT, CBC, NW>(sreg, xs, 0, v, wave, lane);
    }
    __syncthreads();

    unsigned pl[RT];                            // this lane's word in each vertex tile (no ring shift: k = 1)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
        pl[rt] = (unsigned)(h * G::VP + min(rt * 32 + r, v - 1)) * 16u;

    int k = 0;
    for (; nn < n; nn += ostep) {
        f32x16 acc[2][RT];
        init_acc16<RT, 2>(acc, a.bias_s, o0, h, 1.0f);
        h8 ah[2][2], al[2][2];
#define F_LOAD_A(SLOT, PTR)                                                                       \
    do {                                                                                          \
        const char *pa_ = (PTR);                                                                  \
        _Pragma("unroll") for (int ot_ = 0; ot_ < 2; ++ot_) {                                     \
            ah[SLOT][ot_] = *reinterpret_cast<const h8 *>(pa_ + wlane + ot_ * 512);               \
            if constexpr (TERMS == 3)                                                             \
                al[SLOT][ot_] = *reinterpret_cast<const h8 *>(pa_ + wps_b + wlane + ot_ * 512);   \
        }                                                                                         \
    } while (0)
        F_LOAD_A(0, wbase);
        for (int c = 0; c < NCH; ++c, ++k) {
            const unsigned cur = (k & 1) * G::BYTES, nxt = ((k + 1) & 1) * G::BYTES;
            const bool last = c + 1 == NCH;
            const bool has_next = !last || nn + ostep < n;
            const char *ssrc = has_next ? chunk_src(last ? nn + ostep : nn, last ? 0 : c + 1) : chunk_src(nn, c);
            const char *wchunk = wbase + (long long)c * NST * wstep_b;
            h8 bh[2], bl[2];
#define F_LOAD_B(Q)                                                                               \
    do {                                                                                          \
        constexpr int s_ = (Q) / RT, rt_ = (Q) % RT;                                              \
        bh[(Q) % 2] = *reinterpret_cast<const h8 *>(xs + cur + pl[rt_] + s_ * SSTEP);             \
        if constexpr (TERMS == 3)                                                                 \
            bl[(Q) % 2] = *reinterpret_cast<const h8 *>(xs + cur + pl[rt_] + s_ * SSTEP + G::PLANE); \
    } while (0)
            F_LOAD_B(0);
            dance::static_for<0, NT>([&](auto qq) __attribute__((always_inline)) {
                constexpr int q = decltype(qq)::value, s = q / RT, rt = q % RT;
                if constexpr (rt == 0) {
                    // next step's weights: the sequence runs on linearly into the next chunk
                    if (s + 1 < NST || !last)
                        F_LOAD_A((s + 1) % 2, wchunk + (s + 1) * wstep_b);
                    if constexpr (DMA) {
                        if (has_next)
                            stage_dma<RT, CBC, NW>(s * S::N / NST, (s + 1) * S::N / NST, ssrc, lds_base + nxt, v, wave, lane, vs);
                    } else {
                        stage_load<RT, CBC, NW>(sreg, s * S::N / NST, (s + 1) * S::N / NST, ssrc, v, wave, lane, vs);
                    }
                }
                if constexpr (q + 1 < NT)
                    F_LOAD_B(q + 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (TERMS == 3) {
                    acc[0][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s % 2][0], bh[q % 2], acc[0][rt], 0, 0, 0);
                    acc[1][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s % 2][1], bh[q % 2], acc[1][rt], 0, 0, 0);
                    acc[0][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s % 2][0], bl[q % 2], acc[0][rt], 0, 0, 0);
                    acc[1][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s % 2][1], bl[q % 2], acc[1][rt], 0, 0, 0);
                }
                acc[0][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s % 2][0], bh[q % 2], acc[0][rt], 0, 0, 0);
                acc[1][rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s % 2][1], bh[q % 2], acc[1][rt], 0, 0, 0);
            });
#undef F_LOAD_B
            if constexpr (DMA) {
                stage_dma_wait();
            } else {
                if (has_next)
                    stage_commit<RT, CBC, NW>(sreg, xs, nxt, v, wave, lane);
            }
            __syncthreads();
        }
#undef F_LOAD_A
        int re = r, he = h;
        asm volatile("" : "+v"(re), "+v"(he));
        // max over the (part of the) ring of the fusion output (snake_head.py:271-272).
        // The 32 cross-lane reductions run side by side, one butterfly round at a time
        // (32 independent shuffles per round instead of 32 serial five-round chains).
        auto ring_max = [&](float *dst) __attribute__((always_inline)) {
            float m[2][16];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float mm = -INFINITY;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        if (rt * 32 + re < v)
                            mm = dance::nan_max(acc[ot][rt][q] * a.winv, mm);
                    m[ot][q] = mm;
                }
#pragma unroll
            for (int s = 16; s >= 1; s >>= 1)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        m[ot][q] = dance::nan_max(__shfl_xor(m[ot][q], s, 64), m[ot][q]);
            if (re == 0) {
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        dst[o0 + ot * 32 + (q & 3) + 8 * (q >> 2) + 4 * he] = m[ot][q];
            }
        };
        if constexpr (RAW) {
            if (o0 < 256) {
                ring_max(a.gpart + (nn * parts + part) * 256);
            } else {
                const int p0 = o0 - 256;                    // prediction output channel of tile 0
                _Float16 *yn = a.y + nn * a.y_stride;
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ob = p0 + ot * 32 + 8 * g;
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const int vv = rt * 32 + re;
                            if (vv >= v)
                                continue;
                            float val[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                val[i] = acc[ot][rt][4 * g + i] * a.winv;
                            dance::store_p16(yn + (ob >> 7) * a.y_slot_stride, vs, (ob & 127) >> 3, v0 + vv, he, val);
                        }
                    }
            }
        } else {
            __shared__ float gs[256];
            __shared__ float hp[8 * 256];
            if (wave < 4)
                ring_max(gs);
            __syncthreads();
            {   // row constant h[o] = pb[o] + sum_g wg[g][o] * G[g]: 8 partial sums per output, each
                // thread 4 consecutive outputs x 32 values of g, 8 16-byte loads in flight per round
                // (a 128-step dependent loop per thread was latency-bound: ~25 us per contour)
                const int o4 = (t & 63) * 4, prt = t >> 6;
                const float *wgp = a.wg + (long long)prt * 32 * 256 + o4;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 1
                for (int k0 = 0; k0 < 32; k0 += 8) {            // 8 loads in flight per round
                    float4 w4[8];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk)
                        w4[kk] = *reinterpret_cast<const float4 *>(wgp + (k0 + kk) * 256);
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        const float gk = gs[prt * 32 + k0 + kk];
                        s0 = fmaf(w4[kk].x, gk, s0);
                        s1 = fmaf(w4[kk].y, gk, s1);
                        s2 = fmaf(w4[kk].z, gk, s2);
                        s3 = fmaf(w4[kk].w, gk, s3);
                    }
                }
                *reinterpret_cast<float4 *>(hp + prt * 256 + o4) = float4{s0, s1, s2, s3};
            }
            __syncthreads();
            if (wave >= 4) {
                const int p0 = o0 - 256;                        // prediction output channel of tile 0
                _Float16 *yn = a.y + nn * a.y_stride;
#pragma unroll
                for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float hc[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int o = p0 + ot * 32 + 8 * g + 4 * he + i;
                            float hs = a.pb[o];
#pragma unroll
                            for (int pp = 0; pp < 8; ++pp)
                                hs += hp[pp * 256 + o];
                            hc[i] = hs;
                        }
                        const int ob = p0 + ot * 32 + 8 * g;
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const int vv = rt * 32 + re;
                            if (vv >= v)
                                continue;
                            float val[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                val[i] = relu_nan(acc[ot][rt][4 * g + i] * a.winv + hc[i]);
                            dance::store_p16(yn + (ob >> 7) * a.y_slot_stride, vs, (ob & 127) >> 3, vv, he, val);
                        }
                    }
            }
            // gs / hp are rewritten only after the next contour's chunk barriers
        }
    }
}

template <int RT, int TERMS>
__global__ __launch_bounds__(512) void k_fuse16(Fuse16Args a, int n)
{
    fuse16_body<RT, TERMS, false>(a, n, 0, a.v, 0, 1);
}

// raw form: grid (contours, parts); every part but the last has RTA * 32 = a.part_v vertices,
// the last one what remains (at most RTB * 32)
template <int RTA, int RTB, int TERMS>
__global__ __launch_bounds__(512) void k_fuse16r(Fuse16Args a, int n)
{
    const int part = blockIdx.y, parts = gridDim.y;
    const int v0 = part * a.part_v, vn = min(a.part_v, a.v - v0);
    if (RTB != RTA && part + 1 == parts)
        fuse16_body<RTB, TERMS, true>(a, n, v0, vn, part, parts);
    else
        fuse16_body<RTA, TERMS, true>(a, n, v0, vn, part, parts);
}

// the two-workgroup split: grid (contours, parts, 2)
template <int RTA, int RTB, int TERMS>
__global__ __launch_bounds__(256, 2) void k_fuse16r4(Fuse16Args a, int n)
{
    const int part = blockIdx.y, parts = gridDim.y;
    const int v0 = part * a.part_v, vn = min(a.part_v, a.v - v0);
    if (RTB != RTA && part + 1 == parts)
        fuse16_body<RTB, TERMS, true, 4>(a, n, v0, vn, part, parts);
    else
        fuse16_body<RTA, TERMS, true, 4>(a, n, v0, vn, part, parts);
}

// Fused gather + de_location + head SnakeBlock, P16 flavour: chunks of 64 feature channels
// are gathered (4 taps x 16 B per lane, 16 lanes per vertex), blended in fp32, split into
// hi/lo halves and written straight into the LDS image the MFMA loop reads.
struct Gather16Args {
    const float *feat, *att, *loc;
    const int *image_idx;
    float *att_out;
    int b, hf, wf, c, reflection, deloc;
    const float *x_in;          // XIN flavour: precomputed (c + 2, v) fp32 input per contour instead of
    long long x_stride;         // the gather (Dsnake init: fused features + de_location channels)
    const h8 *wd;               // de_location channels as ONE 32-deep K step (head16.hip: the 2 x 9
                                // (channel, tap) pairs as an im2col column per vertex, weights
                                // [plane][4][128] words), or null: zero-padded 32-channel chunk, 9 taps
};

// TO_LDS: the workgroup goes on with the residual stack (k_head_stack16s): two chunk images are
// reserved, and the epilogue writes the block's output into them as well as to its HBM slot.
template <int RT, int TERMS, bool XIN, bool TO_LDS = false>
__device__ __forceinline__ void gather_head16_body(const Conv16Args &a, const Gather16Args &g)
{
    constexpr int CBC = 8;                                      // 64 channels per chunk
    using G = Geo<RT, CBC>;
    using GD = Geo<RT, 4>;                                      // de_location chunk: 32 channels
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int v = a.v;
    char *xs = smem;                                            // one chunk image
    float *tw = reinterpret_cast<float *>(smem + (TO_LDS ? 2 : 1) * G::BYTES);     // [v][4]
    int *toff = reinterpret_cast<int *>(tw + 4 * v);            // [v][4]
    float *dl = reinterpret_cast<float *>(toff + 4 * v);        // [2][v]
    float *red = dl + 2 * v;                                    // [16]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long long nn = blockIdx.x;
    const int o0 = wave * 32;
    const float *loc_n = g.loc + nn * v * 2;
    int img = g.image_idx ? g.image_idx[nn] : g.b - 1;
    const bool img_ok = img >= 0 && img < g.b;
    img = img_ok ? img : 0;
    const long long plane = (long long)g.hf * g.wf;

    const float *xn = XIN ? g.x_in + nn * g.x_stride : nullptr;
    float xmin = 0.f, xmax = 0.f, ymin = 0.f, ymax = 0.f;
    if (!XIN && g.deloc != DANCE_DELOC_NONE)
        dance::contour_bounds(loc_n, v, red, xmin, xmax, ymin, ymax);
    if constexpr (XIN) {
        for (int vi = t; vi < v; vi += 256) {
            dl[vi] = xn[(long long)g.c * v + vi];
            dl[v + vi] = xn[(long long)(g.c + 1) * v + vi];
        }
    } else
    for (int vi = t; vi < v; vi += 256) {
        const float lx = loc_n[2 * vi], ly = loc_n[2 * vi + 1];
        dance::Taps tp = dance::bilinear_taps(lx, ly, g.hf, g.wf, g.reflection);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tw[vi * 4 + k] = img_ok ? tp.w[k] : 0.0f;
            toff[vi * 4 + k] = tp.off[k];
        }
        if (g.att_out != nullptr) {
            float s = 0.0f;
            if (g.att != nullptr) {
#pragma clang fp contract(off)
                const float *ap = g.att + img * plane;
                s = ap[tp.off[0]] * tp.w[0] + ap[tp.off[1]] * tp.w[1] + ap[tp.off[2]] * tp.w[2] +
                    ap[tp.off[3]] * tp.w[3];
                s = img_ok ? s : 0.0f;
            }
            g.att_out[nn * v + vi] = s;
        }
        if (g.deloc != DANCE_DELOC_NONE) {
            dl[vi] = dance::deloc_value(lx, xmin, xmax, g.deloc);
            dl[v + vi] = dance::deloc_value(ly, ymin, ymax, g.deloc);
        }
    }

    f32x16 acc[1][RT];
    init_acc16<RT, 1>(acc, a.bias_s, o0, h, 1.0f);
    const long long wbs_b = (long long)a.cout * 16, wts_b = (long long)(a.cin >> 3) * wbs_b,
                    wps_b = (long long)a.taps * wts_b;
    const unsigned wlane = (unsigned)(h * a.cout + r) * 16u;
    const float *fimg = g.feat + img * plane * g.c;
    const int lc = t & 15, vs = t >> 4;                         // 16 lanes x float4 = 64 channels
    auto nohook = [](int) __attribute__((always_inline)) {};
    for (int c0 = 0; c0 < g.c; c0 += 64) {
        __syncthreads();
        if constexpr (XIN) {
            for (int s = vs; s < v; s += 16) {
                h4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    _Float16 p, q;
                    dance::split_half(xn[(long long)(c0 + 4 * lc + i) * v + s], p, q);
                    hi[i] = p;
                    lo[i] = q;
                }
                char *d = xs + ((lc >> 1) * G::VP + s) * 16 + 8 * (lc & 1);
                *reinterpret_cast<h4 *>(d) = hi;
                *reinterpret_cast<h4 *>(d + G::PLANE) = lo;
            }
        } else
#pragma unroll 2
        for (int s = vs; s < v; s += 16) {
#pragma clang fp contract(off)
            const float w0 = tw[s * 4 + 0], w1 = tw[s * 4 + 1], w2 = tw[s * 4 + 2], w3 = tw[s * 4 + 3];
            const int ch = c0 + 4 * lc;
            const float4 a0 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 0] * g.c + ch);
            const float4 a1 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 1] * g.c + ch);
            const float4 a2 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 2] * g.c + ch);
            const float4 a3 = *reinterpret_cast<const float4 *>(fimg + (long long)toff[s * 4 + 3] * g.c + ch);
            float val[4];
            val[0] = a0.x * w0 + a1.x * w1 + a2.x * w2 + a3.x * w3;
            val[1] = a0.y * w0 + a1.y * w1 + a2.y * w2 + a3.y * w3;
            val[2] = a0.z * w0 + a1.z * w1 + a2.z * w2 + a3.z * w3;
            val[3] = a0.w * w0 + a1.w * w1 + a2.w * w2 + a3.w * w3;
            h4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                _Float16 p, q;
                dance::split_half(val[i], p, q);
                hi[i] = p;
                lo[i] = q;
            }
            char *d = xs + ((lc >> 1) * G::VP + s) * 16 + 8 * (lc & 1);
            *reinterpret_cast<h4 *>(d) = hi;
            *reinterpret_cast<h4 *>(d + G::PLANE) = lo;
        }
        __syncthreads();
        mfma16_core<RT, 1, 4, 4, 4, TERMS>(acc, xs, 0, v,
                                 reinterpret_cast<const char *>(a.w) + ((long long)(c0 >> 3) * a.cout + o0) * 16,
                                 wlane, wps_b, wts_b, 2 * wbs_b, a.taps, a.dilation, r, h, nohook);
    }
    if (g.deloc != DANCE_DELOC_NONE && g.wd != nullptr) {
        // word (block kb, vertex vi) = entries 8 kb .. 8 kb + 7 of
        // [dx(vi - 4 .. vi + 4), dy(vi - 4 .. vi + 4), 0 x 14]; no ring taps left
        __syncthreads();
        for (int i = t; i < 4 * v; i += 256) {
            const int kb = i / v, vi = i - kb * v;
            h8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb * 8 + e;
                if (k < 18) {
                    const int ch = k >= 9, tap = k - 9 * ch;
                    _Float16 p, q;
                    dance::split_half(dl[ch * v + ring16(vi, tap - 4, v)], p, q);
                    hi[e] = p;
                    lo[e] = q;
                }
            }
            *reinterpret_cast<h8 *>(xs + (kb * GD::VP + vi) * 16) = hi;
            *reinterpret_cast<h8 *>(xs + GD::PLANE + (kb * GD::VP + vi) * 16) = lo;
        }
        __syncthreads();
        mfma16_core<RT, 1, 2, 2, 2, TERMS>(acc, xs, 0, v, reinterpret_cast<const char *>(g.wd) + (long long)o0 * 16,
                                 wlane, 4 * wbs_b, 4 * wbs_b, 2 * wbs_b, 1, 1, r, h, nohook);
    } else if (g.deloc != DANCE_DELOC_NONE) {
        // de_location chunk: 4 blocks (weights zero-padded to 32 channels);
        // block 0 = {dx, dy, 0 x 6}, blocks 1..3 = zeros
        __syncthreads();
        for (int i = t; i < 4 * v; i += 256) {
            const int blk = i / v, vi = i - blk * v;
            h8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
            if (blk == 0) {
                _Float16 p, q;
                dance::split_half(dl[vi], p, q);
                hi[0] = p; lo[0] = q;
                dance::split_half(dl[v + vi], p, q);
                hi[1] = p; lo[1] = q;
            }
            *reinterpret_cast<h8 *>(xs + (blk * GD::VP + vi) * 16) = hi;
            *reinterpret_cast<h8 *>(xs + GD::PLANE + (blk * GD::VP + vi) * 16) = lo;
        }
        __syncthreads();
        mfma16_core<RT, 1, 2, 2, 2, TERMS>(acc, xs, 0, v,
                                 reinterpret_cast<const char *>(a.w) + ((long long)(g.c >> 3) * a.cout + o0) * 16,
                                 wlane, wps_b, wts_b, 2 * wbs_b, a.taps, a.dilation, r, h, nohook);
    }
    _Float16 *yslot = a.y + nn * a.y_stride;
    if constexpr (TO_LDS)
        __syncthreads();                        // every wave is done reading the gather buffer
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const float4 sc4 = *reinterpret_cast<const float4 *>(a.scale + o0 + 8 * gq + 4 * h);
        const float4 sh4 = *reinterpret_cast<const float4 *>(a.shift + o0 + 8 * gq + 4 * h);
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int vv = rt * 32 + r;
            if (vv >= v)
                continue;
            float val[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                val[i] = relu_nan(acc[0][rt][4 * gq + i] * a.winv) * sc[i] + sh[i];
            dance::store_p16(yslot, v, (o0 >> 3) + gq, vv, h, val);
            if constexpr (TO_LDS) {
                // the same words into the LDS image (row = channel block, chunks of 8 blocks)
                const int row = (o0 >> 3) + gq;
                h4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    _Float16 p, q;
                    dance::split_half(val[i], p, q);
                    hi[i] = p;
                    lo[i] = q;
                }
                char *d = xs + (row >> 3) * G::BYTES + ((row & 7) * G::VP + vv) * 16 + 8 * h;
                *reinterpret_cast<h4 *>(d) = hi;
                *reinterpret_cast<h4 *>(d + G::PLANE) = lo;
            }
        }
    }
    if constexpr (TO_LDS)
        __syncthreads();                        // the image is complete
}

template <int RT, int TERMS>
__global__ __launch_bounds__(256, 2) void k_gather_head16(Conv16Args a, Gather16Args g)
{
    gather_head16_body<RT, TERMS, false>(a, g);
}

// The same head block on a precomputed fp32 input (n, c + 2, v): Dsnake's init stage, whose
// input is the fused [sampled | centre] features plus the de_location channels.
template <int RT, int TERMS>
__global__ __launch_bounds__(256, 2) void k_xin_head16(Conv16Args a, Gather16Args g)
{
    gather_head16_body<RT, TERMS, true>(a, g);
}

// Fused gather + head block + the 7 residual blocks (V <= 128): the head's output goes to its
// state slot in HBM and straight into the LDS image the residual stack works on -- no second
// launch, no re-read of slot 0, and one workgroup's gather phases (HBM latency) overlap the
// other resident workgroup's residual-block MFMAs.
template <int RT, int TERMS>
__global__ __launch_bounds__(256, 2) void k_head_stack16s(Conv16Args a, Gather16Args g, Stack16Args st)
{
    extern __shared__ __attribute__((aligned(16))) char smem_hs[];
    gather_head16_body<RT, TERMS, false, true>(a, g);
    res_stack16_layers<RT, TERMS>(st, blockIdx.x, smem_hs);
}

int pick_rt16(int v) { return v <= 64 ? 2 : (v <= 128 ? 4 : (v <= 224 ? 7 : 8)); }

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

template <typename K>
int big_lds(K kernel, size_t bytes, const char *fn)
{
    return dance::allow_lds((const void *)kernel, bytes, fn);      // once per kernel and device
}

#define LAUNCH_ONE(KERNEL, RTV, terms, grid, block, lds, s, fn, ...)                          \
    do {                                                                                      \
        if ((terms) == 1) {                                                                   \
            rc_ = big_lds(KERNEL<RTV, 1>, lds, fn);                                           \
            if (!rc_)                                                                         \
                hipLaunchKernelGGL((KERNEL<RTV, 1>), grid, block, lds, s, __VA_ARGS__);       \
        } else {                                                                              \
            rc_ = big_lds(KERNEL<RTV, 3>, lds, fn);                                           \
            if (!rc_)                                                                         \
                hipLaunchKernelGGL((KERNEL<RTV, 3>), grid, block, lds, s, __VA_ARGS__);       \
        }                                                                                     \
    } while (0)

#define LAUNCH_RT(KERNEL, rt, terms, grid, block, lds, s, fn, ...)                           \
    do {                                                                                      \
        int rc_ = 0;                                                                          \
        switch (rt) {                                                                         \
        case 2: LAUNCH_ONE(KERNEL, 2, terms, grid, block, lds, s, fn, __VA_ARGS__); break;    \
        case 4: LAUNCH_ONE(KERNEL, 4, terms, grid, block, lds, s, fn, __VA_ARGS__); break;    \
        case 7: LAUNCH_ONE(KERNEL, 7, terms, grid, block, lds, s, fn, __VA_ARGS__); break;    \
        default: LAUNCH_ONE(KERNEL, 8, terms, grid, block, lds, s, fn, __VA_ARGS__); break;   \
        }                                                                                     \
        if (rc_)                                                                              \
            return rc_;                                                                       \
    } while (0)

// rings of at most 128 vertices only (RT 2 / 4)
#define LAUNCH_RT24(KERNEL, rt, terms, grid, block, lds, s, fn, ...)                         \
    do {                                                                                      \
        int rc_ = 0;                                                                          \
        if ((rt) == 2)                                                                        \
            LAUNCH_ONE(KERNEL, 2, terms, grid, block, lds, s, fn, __VA_ARGS__);               \
        else                                                                                  \
            LAUNCH_ONE(KERNEL, 4, terms, grid, block, lds, s, fn, __VA_ARGS__);               \
        if (rc_)                                                                              \
            return rc_;                                                                       \
    } while (0)

}  // namespace

extern "C" int dance_conv16(const void *x, int64_t x_stride, int64_t x_slot_stride, const void *w16,
                            const float *bias_scaled, float w_inv_scale, const float *scale,
                            const float *shift, const float *rowinit, int residual, int relu,
                            int maxv, void *y16, int64_t y_stride, int64_t y_slot_stride, float *y_f32,
                            int64_t yf_stride, int n, int v, int cin, int cout, int taps, int dilation,
                            int mfma_terms, void *stream)
{
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256);
    DANCE_REQUIRE(cin >= 32 && cin % 32 == 0 && (cin <= 128 || cin % 128 == 0));
    DANCE_REQUIRE(cout >= 32 && cout % 32 == 0);
    DANCE_REQUIRE(taps == 1 || taps == 9);
    DANCE_REQUIRE(dilation >= 1 && dilation * (taps / 2) <= 16);
    DANCE_REQUIRE((y16 != nullptr) != (y_f32 != nullptr));
    DANCE_REQUIRE(!y16 || cout <= 128 || cout % 128 == 0);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(x && w16);
    Conv16Args a;
    a.x = (const _Float16 *)x; a.x_stride = x_stride; a.x_slot_stride = x_slot_stride;
    a.w = (const h8 *)w16; a.bias_s = bias_scaled; a.scale = scale; a.shift = scale ? shift : nullptr;
    a.winv = w_inv_scale; a.y = (_Float16 *)y16; a.y_stride = y_stride; a.y_slot_stride = y_slot_stride;
    a.yf = y_f32; a.yf_stride = yf_stride; a.v = v; a.cin = cin; a.cout = cout; a.taps = taps;
    a.dilation = dilation; a.relu = relu; a.residual = residual; a.rowinit = rowinit; a.maxv = maxv;
    DANCE_REQUIRE(!maxv || (y_f32 && !residual));
    DANCE_REQUIRE(rowinit || bias_scaled);
    DANCE_REQUIRE(!scale || shift);
    DANCE_REQUIRE(!residual || (cin == cout && cin <= 128));
    const int rt = pick_rt16(v);
    const int cbc = rt <= 4 ? 8 : 4;                           // k_conv16's CBC
    DANCE_REQUIRE(cin % (cbc * 8) == 0);
    const size_t lds = (size_t)2 * 2 * cbc * (rt * 32 + 1) * 16;   // two chunk images
    // one workgroup per contour by default (the dispatcher balances them); a smaller grid
    // makes workgroups persistent over contours nn, nn + grid, ... (tuning knob)
    static const int slots = env_int("DANCE_CONV16_SLOTS", 0);
    dim3 grid(slots > 0 && slots < n ? slots : n, dance::ceil_div(cout, 128));
    // 16x16x32 tiles by default for V <= 128 (measured 2-5 % faster at equal MFMA cycles: the
    // chip holds a higher clock on that shape); DANCE_MFMA_SHAPE=32 selects the 32x32x16 kernel
    static const int shape16 = env_int("DANCE_MFMA_SHAPE", 16) == 16;
    if (shape16 && rt <= 4) {
        LAUNCH_RT24(k_conv16s, rt, mfma_terms, grid, dim3(256), lds, (hipStream_t)stream, "dance_conv16", a, n);
        return launch_check(__func__);
    }
    LAUNCH_RT(k_conv16, rt, mfma_terms, grid, dim3(256), lds, (hipStream_t)stream, "dance_conv16", a, n);
    return launch_check(__func__);
}

extern "C" int dance_res_stack16(const dance_snake_net16 *net, void *state, int64_t stride, int64_t slot,
                                 int n, int v, int mfma_terms, void *stream)
{
    DANCE_REQUIRE(net != nullptr && n >= 0 && v >= 16 && v <= 256);
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(state != nullptr);
    Stack16Args a;
    a.state = (_Float16 *)state; a.stride = stride; a.slot = slot; a.v = v; a.nlayers = 7;
    for (int i = 0; i < 7; ++i) {
        DANCE_REQUIRE(net->res_w[i] && net->res_bs[i] && net->res_scale[i] && net->res_shift[i]);
        DANCE_REQUIRE(net->dilation[i] >= 1 && net->dilation[i] * 4 <= 16);
        a.w[i] = (const h8 *)net->res_w[i]; a.bias_s[i] = net->res_bs[i]; a.scale[i] = net->res_scale[i];
        a.shift[i] = net->res_shift[i]; a.winv[i] = net->res_winv[i]; a.dilation[i] = net->dilation[i];
    }
    int rt = pick_rt16(v);
    // 129..192 vertices: images of 160 / 192 columns (5 + 5, 6 + 6 vertex tiles, or one fewer in the
    // second group of waves) instead of 224 (DANCE_STACK_MID=0: 224 columns, 7 + 7 tiles)
    static const int mid = env_int("DANCE_STACK_MID", 1);
    if (mid && v > 128 && v <= 192)
        rt = v <= 160 ? 5 : 6;
    const size_t lds = (size_t)2 * 2 * 8 * (rt * 32 + 1) * 16;     // the whole 128-channel image
    // few contours (the reference's one-image inference: ~100 instances on 256 CUs): a contour per
    // CU either way, so it gets 8 waves instead of 4 (-9 % on the stack at n = 100; same products
    // and summation order as k_res_stack16s, bit-identical).  DANCE_STACK_SMALL_N = largest such n.
    static const int small_n = env_int("DANCE_STACK_SMALL_N", 256);
    // 193..208 vertices: 13 vertex tiles as 7 + 6 instead of 14 (DANCE_STACK_ODD_SPLIT=0: 7 + 7)
    static const int odd_split = env_int("DANCE_STACK_ODD_SPLIT", 1);
    if (rt > 4 || (rt == 4 && n <= small_n)) {                     // 129..256 vertices: 8 waves, one workgroup per CU
        int rc_ = 0;
        if (rt == 4)
            LAUNCH_ONE(k_res_stack16w, 4, mfma_terms, dim3(n), dim3(512), lds, (hipStream_t)stream, "dance_res_stack16", a, n);
        else if (rt == 7 && v > 192 && v <= 208 && odd_split)
            LAUNCH_ONE(k_res_stack16w13, 7, mfma_terms, dim3(n), dim3(512), lds, (hipStream_t)stream, "dance_res_stack16", a, n);
        else if (rt == 5 && v <= 144 && odd_split)                 // 9 tiles: 5 + 4
            LAUNCH_ONE(k_res_stack16w13, 5, mfma_terms, dim3(n), dim3(512), lds, (hipStream_t)stream, "dance_res_stack16", a, n);
        else if (rt == 5)
            LAUNCH_ONE(k_res_stack16w, 5, mfma_terms, dim3(n), dim3(512), lds, (hipStream_t)stream, "dance_res_stack16", a, n);
        else if (rt == 6 && v <= 176 && odd_split)                 // 11 tiles: 6 + 5
            LAUNCH_ONE(k_res_stack16w13, 6, mfma_terms, dim3(n), dim3(512), lds, (hipStream_t)stream, "dance_res_stack16", a, n);
        else if (rt == 6)
            LAUNCH_ONE(k_res_stack16w, 6, mfma_terms, dim3(n), dim3(512), lds, (hipStream_t)stream, "dance_res_stack16", a, n);
        else if (rt == 7)
            LAUNCH_ONE(k_res_stack16w, 7, mfma_terms, dim3(n), dim3(512), lds, (hipStream_t)stream, "dance_res_stack16", a, n);
        else if (mid && v <= 240 && odd_split)                     // 15 tiles: 8 + 7
            LAUNCH_ONE(k_res_stack16w13, 8, mfma_terms, dim3(n), dim3(512), lds, (hipStream_t)stream, "dance_res_stack16", a, n);
        else
            LAUNCH_ONE(k_res_stack16w, 8, mfma_terms, dim3(n), dim3(512), lds, (hipStream_t)stream, "dance_res_stack16", a, n);
        if (rc_)
            return rc_;
        return launch_check(__func__);
    }
    if (v > 32 && v <= 48) {                // 3 of the 64-column image's 4 vertex tiles (Dsnake's 40-point init ring)
        if (mfma_terms == 1)
            hipLaunchKernelGGL((k_res_stack16s<2, 1, 3>), dim3(n), dim3(256), lds, (hipStream_t)stream, a, n);
        else
            hipLaunchKernelGGL((k_res_stack16s<2, 3, 3>), dim3(n), dim3(256), lds, (hipStream_t)stream, a, n);
        return launch_check(__func__);
    }
    if (mid && v > 64 && v <= 112) {
        // 65..112 vertices: 5, 6 or 7 of the 16-wide vertex tiles instead of 8 (up to 96 vertices in a
        // 96-column image: 50 KB of LDS, three workgroups per CU)
        const int tiles = dance::ceil_div(v, 16);
        const size_t lds3 = (size_t)2 * 2 * 8 * (3 * 32 + 1) * 16;
        int rc_ = 0;
#define STACK_S(RTV, CTV, LDS)                                                                     \
    do {                                                                                           \
        if (mfma_terms == 1) {                                                                     \
            rc_ = big_lds(k_res_stack16s<RTV, 1, CTV>, LDS, "dance_res_stack16");                  \
            if (!rc_)                                                                              \
                hipLaunchKernelGGL((k_res_stack16s<RTV, 1, CTV>), dim3(n), dim3(256), LDS, (hipStream_t)stream, a, n); \
        } else {                                                                                   \
            rc_ = big_lds(k_res_stack16s<RTV, 3, CTV>, LDS, "dance_res_stack16");                  \
            if (!rc_)                                                                              \
                hipLaunchKernelGGL((k_res_stack16s<RTV, 3, CTV>), dim3(n), dim3(256), LDS, (hipStream_t)stream, a, n); \
        }                                                                                          \
    } while (0)
        if (tiles <= 5)
            STACK_S(3, 5, lds3);
        else if (tiles == 6)
            STACK_S(3, 6, lds3);
        else
            STACK_S(4, 7, lds);
#undef STACK_S
        if (rc_)
            return rc_;
        return launch_check(__func__);
    }
    LAUNCH_RT24(k_res_stack16s, rt, mfma_terms, dim3(n), dim3(256), lds, (hipStream_t)stream, "dance_res_stack16", a, n);
    return launch_check(__func__);
}

extern "C" int dance_head_stack16(const dance_snake_net16 *net, const float *feat, const float *att,
                                  const float *loc, const int32_t *image_idx, void *state, int64_t stride,
                                  int64_t slot, float *att_out, int n, int v, int b, int hf, int wf,
                                  int padding_mode, int deloc_type, int mfma_terms, void *stream)
{
    DANCE_REQUIRE(net != nullptr && n >= 0 && v >= 16 && v <= 128);
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    const int c = net->feat_channels;
    DANCE_REQUIRE(b >= 1 && hf >= 1 && wf >= 1 && (long long)hf * wf < (1ll << 30));
    DANCE_REQUIRE(c >= 64 && c % 64 == 0);
    DANCE_REQUIRE(padding_mode == DANCE_PAD_ZEROS || padding_mode == DANCE_PAD_REFLECTION);
    DANCE_REQUIRE(deloc_type >= DANCE_DELOC_DERANGE && deloc_type <= DANCE_DELOC_NONE);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(feat && loc && state && net->head_w && net->head_bs && net->head_scale && net->head_shift);
    Conv16Args a;
    a.x = nullptr; a.x_stride = 0; a.x_slot_stride = 0; a.w = (const h8 *)net->head_w; a.bias_s = net->head_bs;
    a.scale = net->head_scale; a.shift = net->head_shift; a.winv = net->head_winv; a.y = (_Float16 *)state;
    a.y_stride = stride; a.y_slot_stride = 0; a.yf = nullptr; a.yf_stride = 0; a.v = v;
    a.cin = (c + (deloc_type != DANCE_DELOC_NONE ? 2 : 0) + 31) & ~31; a.cout = 128; a.taps = 9;
    a.dilation = 1; a.relu = 1; a.residual = 0; a.rowinit = nullptr; a.maxv = 0;
    Gather16Args g;
    g.feat = feat; g.att = att; g.loc = loc; g.image_idx = image_idx; g.att_out = att_out; g.b = b;
    g.hf = hf; g.wf = wf; g.c = c; g.reflection = padding_mode == DANCE_PAD_REFLECTION;
    g.deloc = deloc_type; g.x_in = nullptr; g.x_stride = 0; g.wd = nullptr;
    Stack16Args st;
    st.state = (_Float16 *)state; st.stride = stride; st.slot = slot; st.v = v; st.nlayers = 7;
    for (int i = 0; i < 7; ++i) {
        DANCE_REQUIRE(net->res_w[i] && net->res_bs[i] && net->res_scale[i] && net->res_shift[i]);
        DANCE_REQUIRE(net->dilation[i] >= 1 && net->dilation[i] * 4 <= 16);
        st.w[i] = (const h8 *)net->res_w[i]; st.bias_s[i] = net->res_bs[i]; st.scale[i] = net->res_scale[i];
        st.shift[i] = net->res_shift[i]; st.winv[i] = net->res_winv[i]; st.dilation[i] = net->dilation[i];
    }
    const int rt = pick_rt16(v);
    const size_t lds = (size_t)2 * 2 * 8 * (rt * 32 + 1) * 16 + ((size_t)10 * v + 16) * sizeof(float);
    LAUNCH_RT24(k_head_stack16s, rt, mfma_terms, dim3(n), dim3(256), lds, (hipStream_t)stream,
                "dance_head_stack16", a, g, st);
    return launch_check(__func__);
}

extern "C" int dance_fuse16(const void *x, int64_t x_stride, int64_t x_slot_stride, const void *w16,
                            const float *bias_scaled, float w_inv_scale, const float *wg,
                            const float *pred_bias, void *y16, int64_t y_stride, int64_t y_slot_stride,
                            int n, int v, int mfma_terms, void *stream)
{
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(x && w16 && bias_scaled && wg && pred_bias && y16);
    Fuse16Args a;
    a.x = (const _Float16 *)x; a.x_stride = x_stride; a.x_slot_stride = x_slot_stride;
    a.w = (const h8 *)w16; a.bias_s = bias_scaled; a.wg = wg; a.pb = pred_bias; a.winv = w_inv_scale;
    a.y = (_Float16 *)y16; a.y_stride = y_stride; a.y_slot_stride = y_slot_stride; a.v = v;
    DANCE_REQUIRE(v <= 128);        // larger rings: unfused dance_conv16 sequence (accumulators)
    const int rt = v <= 64 ? 2 : 4;
    const size_t lds = (size_t)2 * 2 * 16 * (rt * 32 + 1) * 16;    // two 128-channel chunk images
    static const int slots = env_int("DANCE_FUSE16_SLOTS", 0);
    dim3 grid(slots > 0 && slots < n ? slots : n);
    LAUNCH_RT24(k_fuse16, rt, mfma_terms, grid, dim3(512), lds, (hipStream_t)stream, "dance_fuse16", a, n);
    return launch_check(__func__);
}

extern "C" int dance_fuse16_parts(int n, int v)
{
    // few contours (the reference's one-image inference: ~100 instances on 256 CUs): rings of
    // 97..128 vertices as two parts of 64 as well, so that every contour keeps two CUs busy.
    // DANCE_FUSE16_SMALL_N = largest such n (default 128, 0 = never).
    static const int small_n = env_int("DANCE_FUSE16_SMALL_N", 128);
    return v > 128 || (v > 96 && n <= small_n) ? 2 : 1;
}

extern "C" int dance_fuse16_raw(const void *x, int64_t x_stride, int64_t x_slot_stride, const void *w16,
                                const float *bias_scaled, float w_inv_scale, void *y16, int64_t y_stride,
                                int64_t y_slot_stride, float *gpart, int n, int v, int mfma_terms,
                                void *stream)
{
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(x && w16 && bias_scaled && y16 && gpart);
    Fuse16Args a;
    a.x = (const _Float16 *)x; a.x_stride = x_stride; a.x_slot_stride = x_slot_stride;
    a.w = (const h8 *)w16; a.bias_s = bias_scaled; a.wg = nullptr; a.pb = nullptr; a.winv = w_inv_scale;
    a.y = (_Float16 *)y16; a.y_stride = y_stride; a.y_slot_stride = y_slot_stride; a.v = v;
    a.gpart = gpart;
    // 32-wide vertex tiles; rings above 128 vertices as two parts of rta and rtb tiles
    const int tiles = dance::ceil_div(v, 32), parts = dance_fuse16_parts(n, v);
    const int rta = parts == 1 ? (tiles < 2 ? 2 : tiles) : (tiles + 1) / 2, rtb = parts == 1 ? rta : tiles - rta;
    a.part_v = parts == 1 ? v : rta * 32;
    // DANCE_FUSE16_WAVES=4 (read per call: the tests run both): two 4-wave workgroups per contour part
    // (64-channel chunks, LDS-direct staging) instead of one 8-wave workgroup.  Same box, twice: fusion
    // 0.595 / 0.600 vs 0.613 / 0.616 ms at 128 vertices, 1.055 vs 1.040 ms at 196 -- a wash, opt-in.
    const int waves = env_int("DANCE_FUSE16_WAVES", 8);
    const bool w4 = waves == 4;
    const size_t lds = (size_t)2 * 2 * (w4 ? 8 : 16) * (rta * 32 + 1) * 16;   // two chunk images
    static const int slots = env_int("DANCE_FUSE16_SLOTS", 0);
    dim3 grid(slots > 0 && slots < n ? slots : n, parts, w4 ? 2 : 1);
    int rc_ = 0;
#define FUSE_RAW_K(K, A, B, T, BLOCK)                                                              \
    do {                                                                                           \
        rc_ = dance::allow_lds((const void *)K<A, B, T>, lds, "dance_fuse16_raw");                 \
        if (!rc_)                                                                                  \
            hipLaunchKernelGGL((K<A, B, T>), grid, dim3(BLOCK), lds, (hipStream_t)stream, a, n);   \
    } while (0)
#define FUSE_RAW(A, B)                                                                             \
    do {                                                                                           \
        if (w4 && mfma_terms == 1)                                                                 \
            FUSE_RAW_K(k_fuse16r4, A, B, 1, 256);                                                  \
        else if (w4)                                                                               \
            FUSE_RAW_K(k_fuse16r4, A, B, 3, 256);                                                  \
        else if (mfma_terms == 1)                                                                  \
            FUSE_RAW_K(k_fuse16r, A, B, 1, 512);                                                   \
        else                                                                                       \
            FUSE_RAW_K(k_fuse16r, A, B, 3, 512);                                                   \
    } while (0)
    if (rta == 2 && rtb == 2)
        FUSE_RAW(2, 2);
    else if (rta == 3 && rtb == 3)
        FUSE_RAW(3, 3);
    else if (rta == 4 && rtb == 4)
        FUSE_RAW(4, 4);
    else if (rta == 3 && rtb == 2)
        FUSE_RAW(3, 2);
    else if (rta == 4 && rtb == 3)
        FUSE_RAW(4, 3);
    else
        return dance::fail((int)hipErrorInvalidValue, __func__, "unsupported split of the ring");
#undef FUSE_RAW
#undef FUSE_RAW_K
    if (rc_)
        return rc_;
    return launch_check(__func__);
}

extern "C" int dance_xin_head16(const float *x_in, int64_t x_stride, const void *w16, const float *bias_scaled,
                                float w_inv_scale, const float *scale, const float *shift, void *y16,
                                int64_t y_stride, int n, int v, int c, int mfma_terms, void *stream)
{
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256);
    DANCE_REQUIRE(c >= 64 && c % 64 == 0 && x_stride >= (int64_t)(c + 2) * v);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(x_in && w16 && bias_scaled && scale && shift && y16);
    Conv16Args a;
    a.x = nullptr; a.x_stride = 0; a.x_slot_stride = 0; a.w = (const h8 *)w16; a.bias_s = bias_scaled;
    a.scale = scale; a.shift = shift; a.winv = w_inv_scale; a.y = (_Float16 *)y16; a.y_stride = y_stride;
    a.y_slot_stride = 0; a.yf = nullptr; a.yf_stride = 0; a.v = v;
    a.cin = (c + 2 + 31) & ~31; a.cout = 128; a.taps = 9;
    a.dilation = 1; a.relu = 1; a.residual = 0; a.rowinit = nullptr; a.maxv = 0;
    Gather16Args g;
    g.feat = nullptr; g.att = nullptr; g.loc = x_in; g.image_idx = nullptr; g.att_out = nullptr; g.b = 1;
    g.hf = 1; g.wf = 1; g.c = c; g.reflection = 0; g.deloc = DANCE_DELOC_DEMIN;
    g.x_in = x_in; g.x_stride = x_stride; g.wd = nullptr;
    const int rt = pick_rt16(v);
    const size_t lds = (size_t)2 * 8 * (rt * 32 + 1) * 16 + ((size_t)10 * v + 16) * sizeof(float);
    LAUNCH_RT(k_xin_head16, rt, mfma_terms, dim3(n), dim3(256), lds, (hipStream_t)stream, "dance_xin_head16", a, g);
    return launch_check(__func__);
}

extern "C" int dance_gather_head16(const float *feat, const float *att, const float *loc,
                                   const int32_t *image_idx, const void *w16, const void *wd16,
                                   const float *bias_scaled,
                                   float w_inv_scale, const float *scale, const float *shift, void *y16,
                                   int64_t y_stride, float *att_out, int n, int v, int b, int hf, int wf,
                                   int c, int padding_mode, int deloc_type, int mfma_terms, void *stream)
{
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    DANCE_REQUIRE(n >= 0 && v >= 16 && v <= 256);
    DANCE_REQUIRE(b >= 1 && hf >= 1 && wf >= 1 && (long long)hf * wf < (1ll << 30));
    DANCE_REQUIRE(c >= 64 && c % 64 == 0);
    DANCE_REQUIRE(padding_mode == DANCE_PAD_ZEROS || padding_mode == DANCE_PAD_REFLECTION);
    DANCE_REQUIRE(deloc_type >= DANCE_DELOC_DERANGE && deloc_type <= DANCE_DELOC_NONE);
    if (n == 0)
        return 0;
    DANCE_REQUIRE(feat && loc && w16 && bias_scaled && scale && shift && y16);
    Conv16Args a;
    a.x = nullptr; a.x_stride = 0; a.x_slot_stride = 0; a.w = (const h8 *)w16; a.bias_s = bias_scaled;
    a.scale = scale; a.shift = shift; a.winv = w_inv_scale; a.y = (_Float16 *)y16; a.y_stride = y_stride;
    a.y_slot_stride = 0; a.yf = nullptr; a.yf_stride = 0; a.v = v;
    a.cin = (c + (deloc_type != DANCE_DELOC_NONE ? 2 : 0) + 31) & ~31; a.cout = 128; a.taps = 9;
    a.dilation = 1; a.relu = 1; a.residual = 0; a.rowinit = nullptr; a.maxv = 0;
    Gather16Args g;
    g.feat = feat; g.att = att; g.loc = loc; g.image_idx = image_idx; g.att_out = att_out; g.b = b;
    g.hf = hf; g.wf = wf; g.c = c; g.reflection = padding_mode == DANCE_PAD_REFLECTION;
    g.deloc = deloc_type;
    g.x_in = nullptr; g.x_stride = 0; g.wd = (const h8 *)wd16;
    const int rt = pick_rt16(v);
    const size_t lds = (size_t)2 * 8 * (rt * 32 + 1) * 16 + ((size_t)10 * v + 16) * sizeof(float);
    LAUNCH_RT(k_gather_head16, rt, mfma_terms, dim3(n), dim3(256), lds, (hipStream_t)stream,
              "dance_gather_head16", a, g);
    return launch_check(__func__);
}
