// fuse16w.hip -- the fusion kernel: fusion 1024 -> 256 + max over the ring part, and prediction[0]'s
// 1024 state columns -> 256 (no row constant, no ReLU: tail16.hip), in one pass over the state
// (_SnakeNet.forward snake_head.py:269-278), on v_mfma_f32_16x16x32_f16 with 16-wide vertex tiles.
//
// One 8-wave workgroup per contour part; wave w owns outputs 64 w .. 64 w + 63 (4 row tiles of 16)
// x the part's CT column tiles of 16: 196 vertices run as 7 + 6 tiles (208 columns) instead of 4 + 3
// tiles of 32 (224).  K = 1024 as 32 steps of 32 channels; 128-channel chunk images double-buffered
// in LDS and staged by LDS-direct loads (no staging registers: the 4 x CT accumulator tiles and
// the two rotating weight buffers of 4 row tiles x hi/lo fill the register file).
#include "common.h"
#include "mfma16.h"
#include "p16.h"
#include "sampling.h"

using dance::launch_check;

#ifndef DANCE_FUSE_DS
#define DANCE_FUSE_DS 2
#endif

namespace {

struct Fuse16wArgs {
    const _Float16 *x;
    long long x_stride, x_slot_stride;
    const h8 *w;                  // [plane][1][128][512] words
    const float *bias_s;          // (512): fusion bias * 2^s, then zeros
    float winv;                   // fusion rows: 2^-s
    float winv_p;                 // prediction rows: 2^-s' * 2^(scale of the stored block), p16.h
    _Float16 *y;                  // P16, 2 slots
    long long y_stride, y_slot_stride;
    int v;                        // ring length = vertices per row of x and y
    float *gpart;                 // (n, parts, 256) max of the fusion output over each part
    int part_v;                   // vertices of every part but the last
};

template <int RT, int CT, int TERMS>
__device__ __forceinline__ void fuse16w_body(const Fuse16wArgs &a, int n, int v0, int v, int part, int parts,
                                             long long nn0, int ostep, bool dup)
{
    constexpr int NW = 8, CBC = 16, NST = CBC / 4, NCH = 1024 / (CBC * 8), NT = NST * CT, DS = DANCE_FUSE_DS;
    static_assert(CT * 16 <= RT * 32, "the column tiles fit the image");
    using G = Geo<RT, CBC>;
    using S = StageGeo<RT, CBC, NW>;
    constexpr int SSTEP = 4 * G::VP * 16;
    extern __shared__ __attribute__((aligned(16))) char xs[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const int o0 = wave * 64;
    const int vs = a.v;
    const long long wbs_b = 512 * 16, wps_b = 128 * wbs_b, wstep_b = 4 * wbs_b;
    const unsigned wlane = (unsigned)(q4 * 512 + c16) * 16u;
    const char *wbase = reinterpret_cast<const char *>(a.w) + (long long)o0 * 16;

    auto chunk_src = [&](long long m, int c) __attribute__((always_inline)) {
        const int cb0 = c * CBC;                // first channel block of the chunk: slot cb0 / 16
        return reinterpret_cast<const char *>(a.x + m * a.x_stride + (cb0 >> 4) * a.x_slot_stride) +
               (long long)(cb0 & 15) * vs * 16 + v0 * 16;
    };

    long long nn = nn0;
    const unsigned lds_base = __builtin_amdgcn_groupstaticsize();   // the dynamic segment follows the static one
    stage_dma<RT, CBC, NW>(0, S::N, chunk_src(nn, 0), lds_base, v, wave, lane, vs);
    stage_dma_wait();
    __syncthreads();

    // this lane's word in vertex tile 0 (no ring shift: k = 1); tile j is 256 bytes further.  Columns
    // past the part's last vertex are read as they lie (never staged, never stored: MFMA columns are
    // independent)
    const unsigned pl0 = (unsigned)(q4 * G::VP + c16) * 16u;

    int k = 0;
    for (; nn < n; nn += ostep) {
        f32x4 acc[4][CT];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 b4 = *reinterpret_cast<const float4 *>(a.bias_s + o0 + 16 * i + 4 * q4);
            const f32x4 bv = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < CT; ++j)
                acc[i][j] = bv;
        }
        h8 ah[2][4], al[2][4];
#define FW_LOAD_A(SLOT, PTR)                                                                      \
    do {                                                                                          \
        const char *pa_ = (PTR);                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                        \
            ah[SLOT][i_] = *reinterpret_cast<const h8 *>(pa_ + wlane + i_ * 256);                 \
            if constexpr (TERMS == 3)                                                             \
                al[SLOT][i_] = *reinterpret_cast<const h8 *>(pa_ + wps_b + wlane + i_ * 256);     \
        }                                                                                         \
    } while (0)
        FW_LOAD_A(0, wbase);
        for (int c = 0; c < NCH; ++c, ++k) {
            const unsigned cur = (k & 1) * G::BYTES, nxt = ((k + 1) & 1) * G::BYTES;
            const bool last = c + 1 == NCH;
            const bool has_next = !last || nn + ostep < n;
            const char *ssrc = has_next ? chunk_src(last ? nn + ostep : nn, last ? 0 : c + 1) : chunk_src(nn, c);
            const char *wchunk = wbase + (long long)c * NST * wstep_b;
            h8 bh[2], bl[2];
#define FW_LOAD_B(Q)                                                                              \
    do {                                                                                          \
        constexpr int s_ = (Q) / CT, j_ = (Q) % CT;                                               \
        bh[(Q) % 2] = *reinterpret_cast<const h8 *>(xs + cur + pl0 + j_ * 256 + s_ * SSTEP);      \
        if constexpr (TERMS == 3)                                                                 \
            bl[(Q) % 2] = *reinterpret_cast<const h8 *>(xs + cur + pl0 + j_ * 256 + s_ * SSTEP + G::PLANE); \
    } while (0)
            FW_LOAD_B(0);
            dance::static_for<0, NT>([&](auto qq) __attribute__((always_inline)) {
                constexpr int q = decltype(qq)::value, s = q / CT, j = q % CT;
                if constexpr (j == 0) {
                    // next step's weights: the sequence runs on linearly into the next chunk
                    if (s + 1 < NST || !last)
                        FW_LOAD_A((s + 1) % 2, wchunk + (s + 1) * wstep_b);
                    // the next chunk image, issued over the first DS of the chunk's NST K steps so that
                    // the last loads have NST - DS steps of MFMAs to land behind
                    if (has_next && s < DS)
                        stage_dma<RT, CBC, NW>(s * S::N / DS, (s + 1) * S::N / DS, ssrc, lds_base + nxt, v, wave,
                                               lane, vs);
                }
                if constexpr (q + 1 < NT)
                    FW_LOAD_B(q + 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (TERMS == 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s % 2][i], bh[q % 2], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s % 2][i], bl[q % 2], acc[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s % 2][i], bh[q % 2], acc[i][j], 0, 0, 0);
            });
#undef FW_LOAD_B
            stage_dma_wait();
            __syncthreads();
        }
#undef FW_LOAD_A
        int ce = c16, qe = q4;
        asm volatile("" : "+v"(ce), "+v"(qe));
        const float winv = o0 < 256 ? a.winv : a.winv_p;
        if (o0 < 256) {
            // max over the part of the ring of the fusion output (snake_head.py:271-272): lane = vertex
            // column 16 j + c16, row tile i = channels 16 i + 4 q4 .. + 3; 16 butterflies side by side
            float m[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float mm = -INFINITY;
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        if (j * 16 + ce < v)
                            mm = dance::nan_max(acc[i][j][e] * winv, mm);
                    m[i][e] = mm;
                }
#pragma unroll
            for (int sft = 8; sft >= 1; sft >>= 1)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        m[i][e] = dance::nan_max(__shfl_xor(m[i][e], sft, 64), m[i][e]);
            if (ce == 0) {
                float *dst = a.gpart + (nn * parts + part) * 256;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 m4 = float4{m[i][0], m[i][1], m[i][2], m[i][3]};
                    *reinterpret_cast<float4 *>(dst + o0 + 16 * i + 4 * qe) = m4;
                    if (dup)                            // a whole ring among split ones: both part slots
                        *reinterpret_cast<float4 *>(dst + 256 + o0 + 16 * i + 4 * qe) = m4;
                }
            }
        } else {
            const int p0 = o0 - 256;                        // prediction output channel of row tile 0
            _Float16 *yn = a.y + nn * a.y_stride;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ob = p0 + 16 * i + 4 * qe;
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const int vv = j * 16 + ce;
                    if (vv >= v)
                        continue;
                    float val[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        val[e] = acc[i][j][e] * winv;
                    dance::store_p16(yn + (ob >> 7) * a.y_slot_stride, vs, (ob & 127) >> 3, v0 + vv, (ob >> 2) & 1, val);
                }
            }
        }
    }
}

// grid (contours, parts); every part but the last has a.part_v = CTA * 16 vertices, the last one what
// remains (at most CTB * 16)
template <int CTA, int CTB, int TERMS>
__global__ __launch_bounds__(512) void k_fuse16w(Fuse16wArgs a, int n)
{
    constexpr int RT = CTA <= 4 ? 2 : 4;
    const int part = blockIdx.y, parts = gridDim.y;
    const int v0 = part * a.part_v, vn = min(a.part_v, a.v - v0);
    if (CTB != CTA && part + 1 == parts)
        fuse16w_body<RT, CTB, TERMS>(a, n, v0, vn, part, parts, blockIdx.x, gridDim.x, false);
    else
        fuse16w_body<RT, CTA, TERMS>(a, n, v0, vn, part, parts, blockIdx.x, gridDim.x, false);
}

// 97..128 vertices, one workgroup per CU: n contours on C CUs run as ceil(n / C) rounds, the last one
// with n mod C workgroups (1600 on 256: 6.25 -> 7 rounds).  Here the first n_whole contours are whole
// rings, the rest two half-ring workgroups each (dispatched last, so the last round is one of half the
// length on twice the CUs); gpart is (n, 2, 256), whole rings write their max to both slots.
template <int TERMS>
__global__ __launch_bounds__(512) void k_fuse16w_split(Fuse16wArgs a, int n, int n_whole)
{
    const int bx = blockIdx.x;
    if (bx < n_whole) {
        fuse16w_body<4, 8, TERMS>(a, n_whole, 0, a.v, 0, 2, bx, n, true);
    } else {
        const int idx = bx - n_whole, part = idx & 1, v0 = part * 64;
        fuse16w_body<2, 4, TERMS>(a, n, v0, min(64, a.v - v0), part, 2, n_whole + (idx >> 1), n, false);
    }
}

template <int CTA, int CTB>
int launch_fuse(const Fuse16wArgs &a, int n, int parts, int terms, hipStream_t s)
{
    constexpr int RT = CTA <= 4 ? 2 : 4;
    const size_t lds = (size_t)2 * Geo<RT, 16>::BYTES;          // two 128-channel chunk images
    // One workgroup per CU fits (two chunk images in LDS), so nothing hides a workgroup's prologue
    // (first chunk staged before any MFMA) and epilogue.  When the contour parts are whole rounds
    // of the CUs, one persistent workgroup per CU loops over its contours instead and stages the
    // next contour's first chunk behind the current one's last K steps: fusion 0.495 -> 0.483 ms at
    // 1536 contours (same box, alternating).
    const int cus = dance::cu_count();
    const long long total = (long long)n * parts;
    const int slots = cus > 0 && cus % parts == 0 && total >= 2ll * cus && total % cus == 0 ? cus / parts : 0;
    dim3 grid(slots > 0 && slots < n ? slots : n, parts);
    int rc;
    if (terms == 1) {
        if ((rc = dance::allow_lds((const void *)k_fuse16w<CTA, CTB, 1>, lds, "dance_fuse16w_raw")))
            return rc;
        hipLaunchKernelGGL((k_fuse16w<CTA, CTB, 1>), grid, dim3(512), lds, s, a, n);
    } else {
        if ((rc = dance::allow_lds((const void *)k_fuse16w<CTA, CTB, 3>, lds, "dance_fuse16w_raw")))
            return rc;
        hipLaunchKernelGGL((k_fuse16w<CTA, CTB, 3>), grid, dim3(512), lds, s, a, n);
    }
    return launch_check("dance_fuse16w_raw");
}

}  // namespace

// 1 when dance_fuse16w_raw has a kernel for (parts, v): one part up to 128 vertices, two parts from 97
extern "C" int dance_fuse16w_supported(int parts, int v)
{
    if (parts == 1)
        return v >= 16 && v <= 128;
    if (parts == 2)
        return v > 96 && v <= 256;
    return 0;
}

// parts a ring is split into (one workgroup each): 2 above 128 vertices, and for few contours (the
// reference's one-image inference: ~100 instances on 256 CUs) rings of 97..128 vertices as two parts
// of 64 as well, so that every contour keeps two CUs busy (policy.few_contour_forms, up to
// DANCE_FEW_FUSE contours)
extern "C" int dance_fuse16_parts(int n, int v, const dance_policy *policy)
{
    const int small_n = dance::policy_or_default(policy).few_contour_forms ? DANCE_FEW_FUSE : 0;
    return v > 128 || (v > 96 && n <= small_n) ? 2 : 1;
}

// trailing contours to run as half rings so that the last round of workgroups is a short one: n mod
// CUs when that is at most half the CUs (and there is more than one round); 97..128 vertices only;
// 0 when policy.fuse_tail_split is off.
extern "C" int dance_fuse16w_tail_split(int n, int v, const dance_policy *policy)
{
    if (!dance::policy_or_default(policy).fuse_tail_split || v <= 96 || v > 128)
        return 0;
    const int cus = dance::cu_count();
    if (cus < 2)
        return 0;
    const int rem = n % cus;
    return n > cus && rem > 0 && 2 * rem <= cus ? rem : 0;
}

extern "C" int dance_fuse16w_raw_split(const void *x, int64_t x_stride, int64_t x_slot_stride, const void *w16,
                                       const float *bias_scaled, float w_inv_scale, float pred_w_inv_scale,
                                       void *y16, int64_t y_stride,
                                       int64_t y_slot_stride, float *gpart2, int n, int v, int n_split,
                                       int mfma_terms, void *stream)
{
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    DANCE_REQUIRE(n >= 1 && v > 96 && v <= 128 && n_split >= 1 && n_split <= n);
    DANCE_REQUIRE(x && w16 && bias_scaled && y16 && gpart2);
    Fuse16wArgs a;
    a.x = (const _Float16 *)x; a.x_stride = x_stride; a.x_slot_stride = x_slot_stride;
    a.w = (const h8 *)w16; a.bias_s = bias_scaled; a.winv = w_inv_scale; a.winv_p = pred_w_inv_scale;
    a.y = (_Float16 *)y16; a.y_stride = y_stride; a.y_slot_stride = y_slot_stride; a.v = v;
    a.gpart = gpart2; a.part_v = 64;
    const size_t lds = (size_t)2 * Geo<4, 16>::BYTES;
    const int n_whole = n - n_split;
    dim3 grid(n_whole + 2 * n_split);
    int rc;
    if (mfma_terms == 1) {
        if ((rc = dance::allow_lds((const void *)k_fuse16w_split<1>, lds, "dance_fuse16w_raw_split")))
            return rc;
        hipLaunchKernelGGL((k_fuse16w_split<1>), grid, dim3(512), lds, (hipStream_t)stream, a, n, n_whole);
    } else {
        if ((rc = dance::allow_lds((const void *)k_fuse16w_split<3>, lds, "dance_fuse16w_raw_split")))
            return rc;
        hipLaunchKernelGGL((k_fuse16w_split<3>), grid, dim3(512), lds, (hipStream_t)stream, a, n, n_whole);
    }
    return launch_check("dance_fuse16w_raw_split");
}

extern "C" int dance_fuse16w_raw(const void *x, int64_t x_stride, int64_t x_slot_stride, const void *w16,
                                 const float *bias_scaled, float w_inv_scale, float pred_w_inv_scale,
                                 void *y16, int64_t y_stride,
                                 int64_t y_slot_stride, float *gpart, int n, int v, int parts, int mfma_terms,
                                 void *stream)
{
    DANCE_REQUIRE(mfma_terms == 3 || mfma_terms == 1);
    DANCE_REQUIRE(n >= 0 && dance_fuse16w_supported(parts, v));
    if (n == 0)
        return 0;
    DANCE_REQUIRE(x && w16 && bias_scaled && y16 && gpart);
    Fuse16wArgs a;
    a.x = (const _Float16 *)x; a.x_stride = x_stride; a.x_slot_stride = x_slot_stride;
    a.w = (const h8 *)w16; a.bias_s = bias_scaled; a.winv = w_inv_scale; a.winv_p = pred_w_inv_scale;
    a.y = (_Float16 *)y16; a.y_stride = y_stride; a.y_slot_stride = y_slot_stride; a.v = v;
    a.gpart = gpart;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = dance::ceil_div(v, 16);
    if (parts == 1) {                           // instantiated tile counts: 3 .. 8
        a.part_v = v;
        if (tiles <= 3)
            return launch_fuse<3, 3>(a, n, 1, mfma_terms, s);
        if (tiles <= 4)
            return launch_fuse<4, 4>(a, n, 1, mfma_terms, s);
        if (tiles == 5)
            return launch_fuse<5, 5>(a, n, 1, mfma_terms, s);
        if (tiles == 6)
            return launch_fuse<6, 6>(a, n, 1, mfma_terms, s);
        if (tiles == 7)
            return launch_fuse<7, 7>(a, n, 1, mfma_terms, s);
        return launch_fuse<8, 8>(a, n, 1, mfma_terms, s);
    }
    // two parts: the first of CTA full tiles, the second what remains (at most CTB tiles)
    if (tiles <= 8) {                           // 97..128 vertices, few contours (dance_fuse16_parts)
        a.part_v = 64;
        return launch_fuse<4, 4>(a, n, 2, mfma_terms, s);
    }
    if (tiles <= 10) {
        a.part_v = 80;
        return tiles == 9 ? launch_fuse<5, 4>(a, n, 2, mfma_terms, s) : launch_fuse<5, 5>(a, n, 2, mfma_terms, s);
    }
    if (tiles <= 12) {
        a.part_v = 96;
        return tiles == 11 ? launch_fuse<6, 5>(a, n, 2, mfma_terms, s) : launch_fuse<6, 6>(a, n, 2, mfma_terms, s);
    }
    if (tiles == 13) {
        a.part_v = 112;
        return launch_fuse<7, 6>(a, n, 2, mfma_terms, s);
    }
    if (tiles == 14) {
        a.part_v = 112;
        return launch_fuse<7, 7>(a, n, 2, mfma_terms, s);
    }
    a.part_v = 128;
    return tiles == 15 ? launch_fuse<8, 7>(a, n, 2, mfma_terms, s) : launch_fuse<8, 8>(a, n, 2, mfma_terms, s);
}
