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
// Kernels: k_conv16 (one residual SnakeBlock / k=1 Conv1d: the single-block entry point),
// k_res_stack16s / k_res_stack16w (the 7 residual blocks of a _SnakeNet in one launch).  The head
// block is head16.hip, fusion + prediction[0] fuse16w.hip, the rest of the net tail16.hip.
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
    float rmul;                 // the residual is added as x * rmul (a power of two: input and output
                                // of a block may be stored at different activation scales, p16.h)
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
                                x = fmaf((float)rh[i] + (float)rl[i], a.rmul, x);
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
    float rmul[7];                  // residual multiplier of block l: 2^(scale of slot l + 1 - scale of slot l)
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
        const float winv = a.winv[l], rmul = a.rmul[l];
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
                    x = fmaf((float)rh[i] + (float)rl[i], rmul, x);
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
        const float winv = a.winv[l], rmul = a.rmul[l];
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
                    x = fmaf((float)rh[i] + (float)rl[i], rmul, x);
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

// The residual stack for FEW contours (the reference's published mode: one image, <= 100 instances on
// 256 CUs, dance.py:52): a contour's blocks on TWO workgroups, i.e. two CUs.  The workgroups of a pair
// split the 128 output channels (64 each = one 64-channel chunk image); block l + 1 needs all 128, so
// after every block but the last the halves are exchanged: a workgroup writes its half of state slot
// l + 1 (it goes to HBM for the fusion kernel anyway) with device-scope stores, waits until they
// have left the CU, raises its flag to l + 1, waits for the partner's flag, and reloads the
// partner's chunk image from the slot with device-scope loads.  (Agent-scope FENCES instead --
// __threadfence -- write back and invalidate whole caches: 0.42 ms for the stack at 100 contours
// against 0.13 ms on one workgroup each.)  Same products and summation order per output element as k_res_stack16w / 16s:
// bit-identical results.
//
// Hang safety.  (1) Pairing: workgroup ids p and p + 8 inside a group of 16 (same XCD under the
// round-robin dispatch, so the exchange stays in one L2; correctness does not depend on it), and the
// dispatcher starts a kernel's workgroups in order -- a resident workgroup's partner is at most 8
// places behind it in the queue, so whenever 16 slots are free for this kernel a complete pair runs.
// (2) The wait is bounded by the constant 100 MHz clock (COOP_WAIT_TICKS = 2 s): on expiry the
// workgroup raises the status word, stops waiting for the rest of the contour, fills its half of the
// last slot with NaN (the contour comes out NaN: loud) and finishes -- the grid always drains.
// coop: [0] status (0 = ok), [16 + 2 * contour + half] flags; zeroed by the host entry before the launch.
constexpr long long COOP_WAIT_TICKS = 200000000ll;

// The contraction of one 64-channel chunk image for a wave that owns 2 row tiles x 2 vertex tiles, with
// the 9 taps x 2 steps fully unrolled so that the weights can be prefetched AD - 1 steps ahead across
// tap boundaries (mfma16_core_sw's rotation closes over a tap: one step ahead at 2 steps per tap).
// With a contour per workgroup pair and nothing else on the CU, the latency of the weight fetches
// from L2 -- not the matrix pipe -- is what the loop waits for.  Same products in the same order per
// output element as mfma16_core_sw: bit-identical.
#ifndef DANCE_COOP_AD
#define DANCE_COOP_AD 6
#endif
template <int TERMS>
__device__ __forceinline__ void coop_core(f32x4 (&acc)[2][2], const char *xs, unsigned lds0, int v, const char *wchunk,
                                          unsigned wlane, long long wps_b, long long wts_b, long long wstep_b, int dilation,
                                          int c16, int q4, int j0)
{
    constexpr int CT = 2, NST = 2, TAPS = 9, NG = TAPS * NST, AD = DANCE_COOP_AD, D = AD - 1;
    using G = Geo<4, 8>;
    constexpr int SSTEP = 4 * G::VP * 16;
    const unsigned rowb = lds0 + q4 * G::VP * 16;
    const unsigned lim = rowb + v * 16, d16 = dilation * 16, v16 = v * 16;
    unsigned p[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j)
        p[j] = rowb + ring16((j0 + j) * 16 + c16, -(TAPS >> 1) * dilation, v) * 16;
    h8 ah[AD][2], al[AD][2], bh[2][CT], bl[2][CT];
#define COOP_LOAD_A(G_)                                                                               \
    do {                                                                                              \
        const char *pa_ = wchunk + ((G_) / NST) * wts_b + ((G_) % NST) * wstep_b;                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                            \
            ah[(G_) % AD][i_] = *reinterpret_cast<const h8 *>(pa_ + wlane + i_ * 256);                \
            if constexpr (TERMS == 3)                                                                 \
                al[(G_) % AD][i_] = *reinterpret_cast<const h8 *>(pa_ + wps_b + wlane + i_ * 256);    \
        }                                                                                             \
    } while (0)
#define COOP_LOAD_B(G_)                                                                               \
    do {                                                                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < CT; ++j_) {                                           \
            bh[(G_) % 2][j_] = *reinterpret_cast<const h8 *>(xs + p[j_] + ((G_) % NST) * SSTEP);      \
            if constexpr (TERMS == 3)                                                                 \
                bl[(G_) % 2][j_] = *reinterpret_cast<const h8 *>(xs + p[j_] + ((G_) % NST) * SSTEP + G::PLANE); \
            if ((G_) % NST == NST - 1) {                                                              \
                const unsigned tpos_ = p[j_] + d16;                                                   \
                p[j_] = tpos_ >= lim ? tpos_ - v16 : tpos_;                                           \
            }                                                                                         \
        }                                                                                             \
    } while (0)
    dance::static_for<0, D>([&](auto g) __attribute__((always_inline)) { COOP_LOAD_A(decltype(g)::value); });
    COOP_LOAD_B(0);
    dance::static_for<0, NG>([&](auto gg) __attribute__((always_inline)) {
        constexpr int g = decltype(gg)::value;
        if constexpr (g + D < NG)
            COOP_LOAD_A(g + D);
        if constexpr (g + 1 < NG)
            COOP_LOAD_B(g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {               // an accumulator tile's three products back to back (mfma16.h)
                if constexpr (TERMS == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[g % AD][i], bh[g % 2][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g % AD][i], bl[g % 2][j], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g % AD][i], bh[g % 2][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    });
#undef COOP_LOAD_A
#undef COOP_LOAD_B
}

template <int TERMS>
__global__ __launch_bounds__(512) void k_res_stack16c(Stack16Args a, int n, int *coop)
{
    constexpr int RT = 4, CBC = 8, NW = 8, CTW = 2;
    using G = Geo<RT, CBC>;
    using S = StageGeo<RT, CBC, NW>;
    extern __shared__ __attribute__((aligned(16))) char xs[];   // chunk images 0 | 1 = channels 0..63 | 64..127
    __shared__ int gave_up;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const int within = blockIdx.x & 15, half = within >> 3;
    const long long nn = (long long)(blockIdx.x >> 4) * 8 + (within & 7);
    if (nn >= n)
        return;                                 // (both workgroups of a pair leave together)
    const int o0 = 64 * half + 32 * (wave & 1), j0 = CTW * (wave >> 1);
    const int v = a.v;
    const long long wbs_b = 128 * 16, wts_b = 16 * wbs_b, wps_b = 9 * wts_b;
    const unsigned wlane = (unsigned)(q4 * 128 + c16) * 16u;
    _Float16 *sn = a.state + nn * a.stride;
    int *my_flag = coop + 16 + 2 * nn + half, *partner_flag = coop + 16 + 2 * nn + (half ^ 1);
    if (t == 0)
        gave_up = 0;
    {
        u32x4 sreg[S::N];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            stage_load<RT, CBC, NW>(sreg, 0, S::N, reinterpret_cast<const char *>(sn) + (long long)c * CBC * v * 16, v,
                                    wave, lane);
            stage_commit<RT, CBC, NW>(sreg, xs, c * G::BYTES, v, wave, lane);
        }
    }
    __syncthreads();
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
            coop_core<TERMS>(acc, xs, c * G::BYTES, v, wl + (long long)c * CBC * wbs_b, wlane, wps_b, wts_b, 4 * wbs_b, dil,
                             c16, q4, j0);
        __syncthreads();                        // every wave has read the whole image
        int ce = c16, qe = q4;
        asm volatile("" : "+v"(ce), "+v"(qe));
        _Float16 *yslot = sn + (l + 1) * a.slot;
        const float winv = a.winv[l], rmul = a.rmul[l];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int ob = o0 + 16 * i2 + 4 * qe;
            const float4 sc4 = *reinterpret_cast<const float4 *>(a.scale[l] + ob);
            const float4 sh4 = *reinterpret_cast<const float4 *>(a.shift[l] + ob);
            const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            const int xrow = ob >> 3, xhalf = (ob >> 2) & 1;
            const int xcbl = xrow - half * CBC;                                 // my rows are chunk image `half`
#pragma unroll
            for (int j = 0; j < CTW; ++j) {
                const int vv = (j0 + j) * 16 + ce;
                if (vv >= v)
                    continue;
                char *q = xs + half * G::BYTES + (xcbl * G::VP + vv) * 16 + 8 * xhalf;
                const h4 rh = *reinterpret_cast<const h4 *>(q);
                const h4 rl = *reinterpret_cast<const h4 *>(q + G::PLANE);
                h4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = relu_nan(acc[i2][j][i] * winv) * sc[i] + sh[i];
                    x = fmaf((float)rh[i] + (float)rl[i], rmul, x);
                    _Float16 xh, xl;
                    dance::split_half(x, xh, xl);
                    hi[i] = xh;
                    lo[i] = xl;
                }
                *reinterpret_cast<h4 *>(q) = hi;
                *reinterpret_cast<h4 *>(q + G::PLANE) = lo;
                // device-scope (write-through) stores: the partner workgroup, possibly behind another L2,
                // reads these words in a moment -- no cache-wide write-back fence is needed for them
                unsigned long long *d = reinterpret_cast<unsigned long long *>(yslot + ((long long)xrow * v + vv) * 8 + 4 * xhalf);
                __hip_atomic_store(d, __builtin_bit_cast(unsigned long long, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(d + 32ll * v, __builtin_bit_cast(unsigned long long, lo), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (l + 1 == a.nlayers)
            break;
        // ---- exchange: my half of slot l + 1 out, the partner's in.  Every store above has left this wave
        // (vmcnt counts stores too) before the barrier; the flag goes out after it.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                        // ... everybody's stores are out; the image rows I own are updated
        if (t == 0) {
            __hip_atomic_store(my_flag, l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!gave_up) {
                const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                while (__hip_atomic_load(partner_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < l + 1) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > COOP_WAIT_TICKS) {
                        gave_up = 1;
                        atomicOr(coop, 1);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        {
            // the partner's chunk image: rows (plane, channel block) of v 16-byte words, device-scope loads
            const int pc = half ^ 1;
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(yslot);
            constexpr int HB = CBC;                                         // channel blocks of a half = a chunk image
            constexpr int PER = (2 * HB * RT * 32 + 511) / 512;             // 16-byte words per thread (4)
            unsigned long long w8[PER][2];
            // all loads first (they are independent round trips to the far side of the chip), then the LDS writes
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = min(t + 512 * k, 2 * HB * v - 1);
                const int row = i / v, col = i - row * v;                   // row = plane * HB + block of the half
                const int plane = row >= HB ? 1 : 0, cbl = row - plane * HB;
                const unsigned long long *g = src + ((plane * 16ll + pc * HB + cbl) * v + col) * 2;
                w8[k][0] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w8[k][1] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = t + 512 * k;
                if (i < 2 * HB * v) {
                    const int row = i / v, col = i - row * v;
                    const int plane = row >= HB ? 1 : 0, cbl = row - plane * HB;
                    unsigned long long *dst = reinterpret_cast<unsigned long long *>(
                        xs + pc * G::BYTES + plane * G::PLANE + (cbl * G::VP + col) * 16);
                    dst[0] = w8[k][0];
                    dst[1] = w8[k][1];
                }
            }
        }
        __syncthreads();                        // image = block l + 1's input
    }
    if (gave_up) {
        // never a finite wrong contour: my half of the last slot as NaN
        _Float16 *ylast = sn + (long long)a.nlayers * a.slot;
        const _Float16 qnan = __builtin_bit_cast(_Float16, (unsigned short)0x7e00);
        for (int i = t; i < 8 * v * 8; i += 512) {
            const int row = 8 * half + i / (v * 8), rest = i % (v * 8);
            ylast[((long long)row * v) * 8 + rest] = qnan;
        }
    }
}

int pick_rt16(int v) { return v <= 64 ? 2 : (v <= 128 ? 4 : (v <= 224 ? 7 : 8)); }

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
                            int64_t yf_stride, float residual_mul, int n, int v, int cin, int cout, int taps,
                            int dilation, int mfma_terms, void *stream)
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
    a.rmul = residual_mul;
    DANCE_REQUIRE(!maxv || (y_f32 && !residual));
    DANCE_REQUIRE(rowinit || bias_scaled);
    DANCE_REQUIRE(!scale || shift);
    DANCE_REQUIRE(!residual || (cin == cout && cin <= 128));
    const int rt = pick_rt16(v);
    const int cbc = rt <= 4 ? 8 : 4;                           // k_conv16's CBC
    DANCE_REQUIRE(cin % (cbc * 8) == 0);
    const size_t lds = (size_t)2 * 2 * cbc * (rt * 32 + 1) * 16;   // two chunk images
    dim3 grid(n, dance::ceil_div(cout, 128));                  // one workgroup per contour and 128 outputs
    LAUNCH_RT(k_conv16, rt, mfma_terms, grid, dim3(256), lds, (hipStream_t)stream, "dance_conv16", a, n);
    return launch_check(__func__);
}

extern "C" int dance_res_stack16(const dance_snake_net16 *net, void *state, int64_t stride, int64_t slot,
                                 int n, int v, int mfma_terms, const dance_policy *policy, int32_t *coop_scratch,
                                 void *stream)
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
        a.rmul[i] = net->res_rmul[i];
        DANCE_REQUIRE(a.rmul[i] > 0.0f);
    }
    int rt = pick_rt16(v);
    // 129..192 vertices: images of 160 / 192 columns (5 + 5, 6 + 6 vertex tiles, or one fewer in the
    // second group of waves) instead of 224
    if (v > 128 && v <= 192)
        rt = v <= 160 ? 5 : 6;
    const size_t lds = (size_t)2 * 2 * 8 * (rt * 32 + 1) * 16;     // the whole 128-channel image
    // few contours (the reference's one-image inference: ~100 instances on 256 CUs): a contour per
    // CU either way, so it gets 8 waves instead of 4 (-9 % on the stack at n = 100; same products
    // and summation order as k_res_stack16s, bit-identical): up to DANCE_FEW_STACK contours, unless
    // the policy switches the few-contour forms off.
    const dance_policy pol = dance::policy_or_default(policy);
    const int small_n = pol.few_contour_forms ? DANCE_FEW_STACK : 0;
    // still fewer (at most half the CUs): two workgroups, two CUs, per contour (k_res_stack16c)
    if (rt == 4 && pol.few_contour_forms && pol.coop_stack && coop_scratch != nullptr && n <= DANCE_FEW_COOP &&
        2 * n <= dance::cu_count()) {
        if (int rc = dance::zero_async(coop_scratch, sizeof(int32_t) * (16 + 2 * (size_t)n), (hipStream_t)stream, __func__))
            return rc;
        const dim3 grid(16 * dance::ceil_div(n, 8));
        int rc_ = 0;
        if (mfma_terms == 1) {
            rc_ = big_lds(k_res_stack16c<1>, lds, "dance_res_stack16");
            if (!rc_)
                hipLaunchKernelGGL((k_res_stack16c<1>), grid, dim3(512), lds, (hipStream_t)stream, a, n, coop_scratch);
        } else {
            rc_ = big_lds(k_res_stack16c<3>, lds, "dance_res_stack16");
            if (!rc_)
                hipLaunchKernelGGL((k_res_stack16c<3>), grid, dim3(512), lds, (hipStream_t)stream, a, n, coop_scratch);
        }
        if (rc_)
            return rc_;
        return launch_check(__func__);
    }
    // an odd number of vertex tiles goes k + (k - 1) over the two groups of waves: 13 (193..208 vertices,
    // the shipped yamls' 196) as 7 + 6 instead of 14
    constexpr int odd_split = 1, mid = 1;
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
    if (v > 64 && v <= 112) {
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

