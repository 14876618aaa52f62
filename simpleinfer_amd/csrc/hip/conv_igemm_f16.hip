// conv_igemm_f16.hip -- NHWC fp16 convolution as an implicit GEMM on v_mfma_f32_32x32x16_f16 (fp32 accumulate).
//
// BASELINE.json configs[3] (YOLOv5s batch 32 fp16): the reference is fp32 only, so this path has no reference parity
// target; it keeps the fp32 path's contract (same descriptor, same fused epilogues: bias + activation + residual, split
// destination for sibling convs, YOLOv5 Detect decode) with fp16 storage for activations and weights.  Bias, the
// accumulators and all epilogue arithmetic stay fp32; the Detect epilogue writes fp32.
//
// At 2.5 PFLOP/s the matrix cores are 16x faster than in fp32 while HBM is not, so every YOLOv5s layer is now bound by
// memory or latency.  The kernel is therefore the fp32 fast kernel's structure with the cheapest possible tile
// (conv_igemm.hip): raw buffer loads of 16-byte (8-channel) vectors with out-of-image taps as out-of-range offsets, a
// wave-uniform tap walk over the channel-block-major K order (c/B, kh, kw, c%B) with B = 64 (32 for 32-channel layers),
// ONE LDS stage of [row][B+8] halves (the 16 lanes of a ds_read_b128 phase land on 16 disjoint 4-bank groups) and many
// resident workgroups.
// One ds_read_b128 per operand feeds one 32x32x16 MFMA (lane l holds k = 8*(l>>5) .. +7 of row l&31, A and W alike).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "si_hip.h"
#include "si_hip_internal.h"

// No floating-point contraction in this file: the epilogue exists in several template instantiations (interior / edge
// tile, with / without residual, ...) and an image's result must not depend on which one a pixel happens to go through
// (bit-exact batch sharding); with contraction the compiler fuses a*b+c differently per instantiation.  The MFMA
// builtins are unaffected.
#pragma clang fp contract(off)

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// conv_stem_f16.hip: stem kernel of this path (fp32 image in, fp16 out)
bool si_conv_stem_f16_ok(const SiConv2dDesc* d);
// conv_depthwise_f16.hip: groups == ic == oc
bool si_conv_depthwise_f16_ok(const SiConv2dDesc* d);
// conv_slab_f16.hip: 3x3 stride-1 layers over 128 / 256 channels as one-shot row slabs (round 5)
bool si_conv_slab_f16_ok(const SiConv2dDesc* d);
const char* si_conv_slab_f16_name(const SiConv2dDesc* d);
int si_conv_slab_f16_launch(const SiConv2dDesc* d, const void* in, const void* wl, int wl_nb, int wl_ks, const float* bias,
                            const void* residual, void* out, hipStream_t s);

namespace {

struct ConvArgsH {
    const half_t* in;
    const half_t* w;
    const half_t* wl;           // the same weights in MFMA B-operand lane order (f16_lane_*; conv_igemm_f16_bd_kernel)
    int wl_nb, wl_ks;           // its extents per group: 32-channel column blocks, 16-deep k-steps
    const float* bias;
    const half_t* res;
    void* out;                  // half, or float when ymode / out_f32
    int ih, iw, in_ld;
    int oh, ow, out_ld, res_ld;
    int kh, kw, sh, sw, dh, dw, pt, pl;
    int icg, ocg, oc;
    int Kp;                     // kh*kw*icg
    int M, ohow;
    unsigned mg_ohow, mg_ow;    // floor(2^32 / d) of the two per-row divisions of the prologue (fast_div_h)
    int m_tiles, n_tiles;
    int act1, act2;
    float act_param;
    unsigned in_bytes;
    unsigned out_bytes;         // extent of `out` when it fits a buffer resource (the patch kernels' countable stores), else 0
    int pointwise;              // 1x1 stride 1 pad 0: the A matrix is the input tensor itself
    int out_f32;                // plain epilogue writing fp32 (graph outputs)
    half_t* out2;               // split output (sibling convs): channels >= split go to out2
    int out2_ld, split;
    int ymode, yna, yne, yrows_total, yrow_off;
    float ystride;
    const float* ygrid;
    const float* yanchor;
    // dual-source input of a 1x1 conv that consumes cat(..., nn.Upsample(x), ...) (conv_igemm.hip ConvArgs::up): channel blocks
    // [up_cb0, up_cb1) of the K axis are read from the LOW-RESOLUTION tensor `up` at the nearest-neighbour source pixel
    const half_t* up;
    int up_ih, up_iw, up_ld, up_cb0, up_cb1;
    float up_inv_h, up_inv_w;
    unsigned up_bytes;
};

// n / d for 0 <= n < 2^32 with mg = floor(2^32 / d) (0xFFFFFFFF for d = 1): the high product is the quotient or one less
__device__ __forceinline__ int fast_div_h(int n, int d, unsigned mg) {
    unsigned q = __umulhi((unsigned)n, mg);
    if ((unsigned)n - q * (unsigned)d >= (unsigned)d) ++q;
    return (int)q;
}

__device__ __forceinline__ float act_h(int act, float v, float p) {
    switch (act) {
        case SI_ACT_RELU: return fmaxf(v, 0.0f);
        case SI_ACT_SILU: return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        case SI_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        case SI_ACT_HARDSIGMOID: return fminf(fmaxf(v * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f);
        case SI_ACT_HARDSWISH: return v * fminf(fmaxf(v * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f);
        case SI_ACT_LEAKYRELU: return v > 0.0f ? v : v * p;
        default: return v;
    }
}

// 32x32 C/D map: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
template <int TM, int TN, typename OutT>
__device__ __forceinline__ void epilogue_plain(const ConvArgsH& a, f32x16 (&acc)[TM][TN], int g, int mrow0, int ocol0) {
    const bool has_res = a.res != nullptr;
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int o = ocol0 + u * 32;
        if (o >= a.ocg) continue;
        const int oc_abs = g * a.ocg + o;
        const float bv = a.bias ? a.bias[oc_abs] : 0.0f;
        const bool second = a.out2 != nullptr && oc_abs >= a.split;
        OutT* const obase = second ? reinterpret_cast<OutT*>(a.out2) + (oc_abs - a.split) : static_cast<OutT*>(a.out) + oc_abs;
        const int old = second ? a.out2_ld : a.out_ld;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int mb = mrow0 + t * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m < a.M) {
                    float v = acc[t][u][e] + bv;
                    v = act_h(a.act1, v, a.act_param);
                    if (has_res) v += (float)a.res[(size_t)m * a.res_ld + oc_abs];
                    v = act_h(a.act2, v, a.act_param);
                    obase[(size_t)m * old] = si_store_cast<OutT>(v);
                }
            }
        }
    }
}

template <int ACT>
__device__ __forceinline__ float act_c(float v, float p) {
    if (ACT == SI_ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == SI_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v;
}

// straight-line fp16 epilogue for the combinations the graphs produce (see conv_igemm.hip epilogue_lean)
template <int TM, int TN, int ACT1, int ACT2, bool HAS_RES, bool INTERIOR>
__device__ __forceinline__ void epilogue_lean_h(const ConvArgsH& a, f32x16 (&acc)[TM][TN], int g, int mrow0, int ocol0,
                                                const float* bias_pre) {
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int o = ocol0 + u * 32;
        if (o >= a.ocg) continue;
        const int oc_abs = g * a.ocg + o;
        const float bv = bias_pre[u];  // loaded before the K loop
        const bool second = a.out2 != nullptr && oc_abs >= a.split;
        half_t* const obase = second ? a.out2 + (oc_abs - a.split) : static_cast<half_t*>(a.out) + oc_abs;
        const int old = second ? a.out2_ld : a.out_ld;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int mb = mrow0 + t * 32;
            half_t* const op = obase + (size_t)mb * old;
            const half_t* const rp = HAS_RES ? a.res + (size_t)mb * a.res_ld + oc_abs : nullptr;
            // the block's residual values, ALL requested before its first store: a load behind a store that may alias it is not
            // moved up by the compiler, and on gfx9 vmcnt counts the stores too, so "load, wait, add, store" per element was a
            // full memory round trip per element (64 in series per wave on the 64x32 wave tiles of the C3 bottleneck convs).
            // (A row behind M re-reads row M - 1: always a valid address.)
            half_t rv[16];
            if (HAS_RES) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = (e & 3) + 8 * (e >> 2);
                    rv[e] = rp[((INTERIOR || mb + c < a.M) ? c : a.M - 1 - mb) * a.res_ld];
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = (e & 3) + 8 * (e >> 2);
                if (INTERIOR || mb + c < a.M) {
                    float v = act_c<ACT1>(acc[t][u][e] + bv, a.act_param);
                    if (HAS_RES) v += (float)rv[e];
                    op[c * old] = si_store_cast<half_t>(act_c<ACT2>(v, a.act_param));
                }
            }
        }
    }
}

template <int TM, int TN, int ACT1, int ACT2>
__device__ __forceinline__ void epilogue_pick_h(const ConvArgsH& a, f32x16 (&acc)[TM][TN], int g, int mrow0, int ocol0, bool interior,
                                                const float* bias_pre) {
    if (a.res) {
        if (interior) epilogue_lean_h<TM, TN, ACT1, ACT2, true, true>(a, acc, g, mrow0, ocol0, bias_pre);
        else epilogue_lean_h<TM, TN, ACT1, ACT2, true, false>(a, acc, g, mrow0, ocol0, bias_pre);
    } else {
        if (interior) epilogue_lean_h<TM, TN, ACT1, ACT2, false, true>(a, acc, g, mrow0, ocol0, bias_pre);
        else epilogue_lean_h<TM, TN, ACT1, ACT2, false, false>(a, acc, g, mrow0, ocol0, bias_pre);
    }
}

// Detect decode in the epilogue, fp32 out (same as conv_igemm.hip epilogue_yolo; src/layer/yolo_detect.cpp:223-266)
template <int TM, int TN>
__device__ __forceinline__ void epilogue_yolo_h(const ConvArgsH& a, f32x16 (&acc)[TM][TN], int mrow0, int ocol0) {
    const int per_pix = a.yna * a.yne;
    float* out = static_cast<float*>(a.out);
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int o = ocol0 + u * 32;
        if (o >= a.ocg) continue;
        const float bv = a.bias ? a.bias[o] : 0.0f;
        const int anc = o / a.yne;
        const int e_ = o - anc * a.yne;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int mb = mrow0 + t * 32;
            const int img0 = mb / a.ohow;
            const int pix0 = mb - img0 * a.ohow;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int dm = (e & 3) + 8 * (e >> 2);
                if (mb + dm < a.M) {
                    int pix = pix0 + dm, img = img0;
                    if (pix >= a.ohow) {
                        const int adv = pix / a.ohow;
                        pix -= adv * a.ohow;
                        img += adv;
                    }
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-(acc[t][u][e] + bv)));
                    const size_t row = (size_t)pix * a.yna + anc;
                    float v = sg;
                    if (e_ < 2) {
                        v = (sg * 2.0f + a.ygrid[row * 2 + e_]) * a.ystride;
                    } else if (e_ < 4) {
                        const float t2 = sg * 2.0f;
                        v = t2 * t2 * a.yanchor[row * 2 + (e_ - 2)];
                    }
                    out[((size_t)img * a.yrows_total + a.yrow_off) * a.yne + (size_t)pix * per_pix + o] = v;
                }
            }
        }
    }
}

constexpr unsigned OOB_A = 0xFFFFFF00u;
constexpr unsigned OOB_B = 0x80000000u;

// BKH: channels per K-tile = one tap of one BKH-channel block.  64 whenever the channel count allows: a row of the
// tile is then a whole 128-byte line per load instruction (32: half lines) and a 1x1 conv over 64 channels needs a
// single tile with all of its loads in flight at once.
// UPS: the dual-source pointwise form (ConvArgsH::up), its own instantiation so that the ordinary layers carry neither the extra
// row offsets nor the per-K-tile source select
template <int BM, int BN, int WM, int WN, int BKH, bool UPS = false>
__global__ __launch_bounds__(256) void conv_igemm_f16_kernel(const ConvArgsH a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int LDH = BKH + 8;       // halves per LDS row (80 / 144 bytes: conflict-free ds_read_b128 phases)
    constexpr int VPR = BKH / 8;       // 16-byte vectors per row
    constexpr int RPP = 256 / VPR;     // rows per pass of the 256 threads
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int A_IT = BM / RPP;     // 16-byte vectors per thread per K-tile
    constexpr int B_IT = BN / RPP;
    static_assert(TM >= 1 && TN >= 1 && A_IT >= 1 && B_IT >= 1, "tile too small");

    __shared__ __attribute__((aligned(16))) half_t lds[(BM + BN) * LDH];

    const int g = blockIdx.y;
    const int per_chunk = 8 * a.n_tiles;
    const int chunk = blockIdx.x / per_chunk;
    const int r = blockIdx.x - chunk * per_chunk;
    const int m_tile = chunk * 8 + (r & 7);
    const int n_tile = r >> 3;
    if (m_tile >= a.m_tiles) return;
    const int m0 = m_tile * BM;
    const int n0 = n_tile * BN;

    const int tid = threadIdx.x;
    const int kv = tid % VPR;
    const int r0 = tid / VPR;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(a.in + (size_t)g * a.icg), 0, a.in_bytes - (unsigned)g * a.icg * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(a.w + (size_t)g * a.ocg * a.Kp), 0, (unsigned)a.ocg * a.Kp * 2u, 0x00020000);

    const __amdgpu_buffer_rsrc_t rs_up = UPS ? __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.up), 0, a.up_bytes, 0x00020000) : rs_in;
    unsigned a_off[A_IT];
    unsigned u_off[UPS ? A_IT : 1];   // UPS: byte offset of the row's source pixel in the low-resolution tensor
    unsigned long long a_mask[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + RPP * i;
        a_off[i] = 0;
        a_mask[i] = 0ull;
        if (UPS) {
            u_off[i] = OOB_A;
            if (m < a.M) {
                const int img = fast_div_h(m, a.ohow, a.mg_ohow);
                const int rem = m - img * a.ohow;
                const int oy = fast_div_h(rem, a.ow, a.mg_ow);
                const int ox = rem - oy * a.ow;
                // src/layer/upsample.cpp:85-92: src = clamp(int(float(dst) * (1 / scale)), 0, in - 1)
                int sy = (int)((float)oy * a.up_inv_h), sx = (int)((float)ox * a.up_inv_w);
                sy = max(0, min(a.up_ih - 1, sy));
                sx = max(0, min(a.up_iw - 1, sx));
                u_off[i] = (unsigned)((img * a.up_ih + sy) * a.up_iw + sx) * (unsigned)(a.up_ld * 2) + (unsigned)(kv * 16);
            }
        }
        if (m < a.M && a.pointwise) {
            a_off[i] = (unsigned)m * (unsigned)(a.in_ld * 2) + (unsigned)(kv * 16);
            a_mask[i] = 1ull;
        } else if (m < a.M) {
            const int img = fast_div_h(m, a.ohow, a.mg_ohow);
            const int rem = m - img * a.ohow;
            const int oy = fast_div_h(rem, a.ow, a.mg_ow);
            const int ox = rem - oy * a.ow;
            const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
            a_off[i] = (unsigned)((img * a.ih + y0) * a.iw + x0) * (unsigned)(a.in_ld * 2) + (unsigned)(kv * 16);
            unsigned long long mk = 0ull;
            for (int ky = 0; ky < a.kh; ++ky)
                for (int kx = 0; kx < a.kw; ++kx) {
                    const int y = y0 + ky * a.dh, x = x0 + kx * a.dw;
                    if ((unsigned)y < (unsigned)a.ih && (unsigned)x < (unsigned)a.iw) mk |= 1ull << (ky * a.kw + kx);
                }
            a_mask[i] = mk;
        }
    }
    unsigned b_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int o = n0 + r0 + RPP * i;
        b_off[i] = o < a.ocg ? (unsigned)(o * a.Kp * 2 + kv * 16) : OOB_B;
    }

    // Register prefetch ring, DEPTH K-tiles deep.  Measured on YOLOv5s batch 32 (MI355X): depth 1 12.2 k img/s, depth 2
    // 12.1 k, depth 4 10.4 k -- the extra registers cost resident workgroups, and residency (8+ workgroups per CU) is
    // what hides the memory latency here, as in the fp32 kernel.
    constexpr int DEPTH = 1;
    u32x4 pa[DEPTH][A_IT], pb[DEPTH][B_IT];
    int cb = 0, ky = 0, kx = 0;  // wave-uniform K walk (tiles are requested in ascending order)
    auto load_tile = [&](u32x4 (&qa)[A_IT], u32x4 (&qb)[B_IT], int kt) {
        const unsigned delta = (unsigned)((ky * a.dh * a.iw + kx * a.dw) * a.in_ld + cb * BKH) * 2u;
        const int tapbit = ky * a.kw + kx;
        // (a conv whose channel count is a multiple of 8 but not of 32 has every tap's channels zero-padded to whole blocks in the weights:
        // a vector that lies behind the last channel must read zeros, not the next pixel)
        const bool cok = cb * BKH + kv * 8 < a.icg;
        const bool from_up = UPS && cb >= a.up_cb0 && cb < a.up_cb1;   // wave-uniform: this K-tile's channels are upsampled ones
        if (from_up) {
            const unsigned du = (unsigned)(cb - a.up_cb0) * (unsigned)(BKH * 2);
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                qa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_up, u_off[i] == OOB_A ? OOB_A : u_off[i] + du, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const bool ok = ((a_mask[i] >> tapbit) & 1ull) && cok;
                qa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? a_off[i] + delta : OOB_A, 0, 0);
            }
        }
        const unsigned kb = (unsigned)kt * (BKH * 2);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) qb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[i] + kb, 0, 0);
        if (++kx == a.kw) {
            kx = 0;
            if (++ky == a.kh) {
                ky = 0;
                ++cb;
            }
        }
    };
    auto store_tile = [&](const u32x4 (&qa)[A_IT], const u32x4 (&qb)[B_IT]) {
        half_t* As = lds;
        half_t* Bs = lds + BM * LDH;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<u32x4*>(As + (r0 + RPP * i) * LDH + kv * 8) = qa[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<u32x4*>(Bs + (r0 + RPP * i) * LDH + kv * 8) = qb[i];
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.0f;

    const int nk = a.Kp / BKH;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nk) load_tile(pa[d], pb[d], d);
    // this lane's bias values ride along with the first tile's loads (in the epilogue their latency would be paid by every
    // workgroup of a round at the same time)
    float bias_pre[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int o = n0 + wn * TN * 32 + l31 + u * 32;
        bias_pre[u] = (a.bias && o < a.ocg) ? a.bias[g * a.ocg + o] : 0.0f;
    }

    const half_t* As = lds + (wm * TM * 32 + l31) * LDH + lh * 8;
    const half_t* Bs = lds + BM * LDH + (wn * TN * 32 + l31) * LDH + lh * 8;
    for (int kt0 = 0; kt0 < nk; kt0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kt = kt0 + d;
            if (kt < nk) {  // wave-uniform
                if (kt > 0) __syncthreads();  // everyone is done reading the previous tile
                store_tile(pa[d], pb[d]);
                __syncthreads();
                if (kt + DEPTH < nk) load_tile(pa[d], pb[d], kt + DEPTH);  // refill the slot just emptied
#pragma unroll
                for (int q = 0; q < BKH / 16; ++q) {
                    f16x8 fa[TM], fb[TN];
#pragma unroll
                    for (int t = 0; t < TM; ++t) fa[t] = *reinterpret_cast<const f16x8*>(As + t * 32 * LDH + q * 16);
#pragma unroll
                    for (int u = 0; u < TN; ++u) fb[u] = *reinterpret_cast<const f16x8*>(Bs + u * 32 * LDH + q * 16);
#pragma unroll
                    for (int t = 0; t < TM; ++t)
#pragma unroll
                        for (int u = 0; u < TN; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t], fb[u], acc[t][u], 0, 0, 0);
                }
            }
        }
    }

    const int mrow0 = m0 + wm * TM * 32 + 4 * lh, ocol0 = n0 + wn * TN * 32 + l31;
    if (a.ymode) {
        const int img = m0 / a.ohow;
        if (m0 + BM <= a.M && m0 - img * a.ohow + BM <= a.ohow)  // the tile lies inside one image: straight-line decode
            si_yolo_tile_one_image<TM, TN>(a, static_cast<float*>(a.out), acc, mrow0, ocol0, img);
        else
            epilogue_yolo_h<TM, TN>(a, acc, mrow0, ocol0);
    } else if (a.out_f32) epilogue_plain<TM, TN, float>(a, acc, g, mrow0, ocol0);
    else {
        const bool interior = m0 + BM <= a.M;
        if (a.act1 == SI_ACT_SILU && a.act2 == SI_ACT_NONE) epilogue_pick_h<TM, TN, SI_ACT_SILU, SI_ACT_NONE>(a, acc, g, mrow0, ocol0, interior, bias_pre);
        else if (a.act1 == SI_ACT_NONE && a.act2 == SI_ACT_NONE) epilogue_pick_h<TM, TN, SI_ACT_NONE, SI_ACT_NONE>(a, acc, g, mrow0, ocol0, interior, bias_pre);
        else if (a.act1 == SI_ACT_RELU && a.act2 == SI_ACT_NONE) epilogue_pick_h<TM, TN, SI_ACT_RELU, SI_ACT_NONE>(a, acc, g, mrow0, ocol0, interior, bias_pre);
        else if (a.act1 == SI_ACT_NONE && a.act2 == SI_ACT_RELU) epilogue_pick_h<TM, TN, SI_ACT_NONE, SI_ACT_RELU>(a, acc, g, mrow0, ocol0, interior, bias_pre);
        else epilogue_plain<TM, TN, half_t>(a, acc, g, mrow0, ocol0);
    }
}

// ---- B operand straight from L2 ("bd" kernels, round 4) ---------------------------------------------------------------------
// The one-stage kernel above moves BOTH operands global -> registers -> ds_write_b128 -> barrier -> ds_read_b128 -> MFMA.  At
// 32 cycles per v_mfma_f32_32x32x16_f16 that path, not the matrix pipe, sets the pace (rocprofv3 --pmc on the K = 2304 layers:
// pipe 21 % busy, 61 % of the issue stalls are LDS issue, profiles/r03_f16_pipe_sweep.txt): a 32x32 wave tile needs two 1 KB
// fragment reads per MFMA and the weight tile is written to and read back from LDS by every workgroup although its layout is
// known at load time.  Here the weights are packed ONCE (si_hip_conv2d_f16_pack_weight_host) in the MFMA's B-operand LANE ORDER:
// per (32-channel column block, 16-deep k-step) one contiguous 1 KB block, lane l holding W[col l & 31][k = 8 (l >> 5) .. + 7],
// so a wave fetches a B fragment with ONE fully coalesced 16-byte-per-lane buffer load -- no ds_write, no ds_read, no barrier
// dependency for B.  LDS carries the im2col A tile only (two stages, one barrier per K-tile), wave tiles are up to 64x64 (four
// MFMAs per A read), and the fragments of K-tile t+1 are in flight in a second register slot while tile t is consumed.
// Same k order and the same 16-wide MFMA steps as the one-stage kernel: an output element sees the same MFMA sequence, so the
// two kernels agree bit for bit and the tile policy may follow the launch size (tests/test_gpu_f16.py).
// SI_F16_ABL (diagnostic builds only, tools/f16_ablate.sh): bit 0 no A loads in the loop, 1 no B loads, 2 no LDS stores, 3 no
// MFMAs (operands kept alive), 4 no barriers -- wrong results, timing only.  0 in the product build.
#ifndef SI_F16_ABL
#define SI_F16_ABL 0
#endif
// NA / NB: register slots of the A / B prefetch rings.  A tile kt + 1 + NA and B tile kt + NB - 1 are requested while tile kt
// multiplies (NA = 1, NB = 2: one K-tile ahead each, the form of the sweeps); deeper rings keep more bytes in flight per wave, which
// is what a launch of one or two workgroups per CU is short of (bytes in flight per CU / memory latency = its load rate).
template <int BM, int BN, int WM, int WN, int BKH, int MINW = 1, int NA = 1, int NB = 2>
__global__ __launch_bounds__(256, MINW) void conv_igemm_f16_bd_kernel(const ConvArgsH a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int LDH = BKH + 8;
    constexpr int VPR = BKH / 8;
    constexpr int RPP = 256 / VPR;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int A_IT = (BM + RPP - 1) / RPP;   // (a 32-row tile has fewer rows than one pass of the 256 threads: the upper threads stage nothing)
    constexpr int QS = BKH / 16;       // MFMA k-steps per K-tile
    static_assert(TM >= 1 && TN >= 1 && A_IT >= 1, "tile too small");

    __shared__ __attribute__((aligned(16))) half_t lds[2][BM * LDH];

    const int g = blockIdx.y;
    const int per_chunk = 8 * a.n_tiles;
    const int chunk = blockIdx.x / per_chunk;
    const int r = blockIdx.x - chunk * per_chunk;
    const int m_tile = chunk * 8 + (r & 7);
    const int n_tile = r >> 3;
    if (m_tile >= a.m_tiles) return;
    const int m0 = m_tile * BM;
    const int n0 = n_tile * BN;

    const int tid = threadIdx.x;
    const int kv = tid % VPR;
    const int r0 = tid / VPR;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(a.in + (size_t)g * a.icg), 0, a.in_bytes - (unsigned)g * a.icg * 2u, 0x00020000);
    const unsigned wl_group_bytes = (unsigned)a.wl_nb * (unsigned)a.wl_ks * 1024u;
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(a.wl) + (size_t)g * (wl_group_bytes / 2u), 0, wl_group_bytes, 0x00020000);

    unsigned a_off[A_IT];
    unsigned long long a_mask[A_IT];
    const bool stager = BM >= RPP || r0 < BM;   // this thread stages A rows at all
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + RPP * i;
        a_off[i] = 0;
        a_mask[i] = 0ull;
        if (!stager) continue;
        if (m < a.M && a.pointwise) {
            a_off[i] = (unsigned)m * (unsigned)(a.in_ld * 2) + (unsigned)(kv * 16);
            a_mask[i] = 1ull;
        } else if (m < a.M) {
            const int img = fast_div_h(m, a.ohow, a.mg_ohow);
            const int rem = m - img * a.ohow;
            const int oy = fast_div_h(rem, a.ow, a.mg_ow);
            const int ox = rem - oy * a.ow;
            const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
            a_off[i] = (unsigned)((img * a.ih + y0) * a.iw + x0) * (unsigned)(a.in_ld * 2) + (unsigned)(kv * 16);
            unsigned long long mk = 0ull;
            for (int ky = 0; ky < a.kh; ++ky)
                for (int kx = 0; kx < a.kw; ++kx) {
                    const int y = y0 + ky * a.dh, x = x0 + kx * a.dw;
                    if ((unsigned)y < (unsigned)a.ih && (unsigned)x < (unsigned)a.iw) mk |= 1ull << (ky * a.kw + kx);
                }
            a_mask[i] = mk;
        }
    }
    // this wave's column blocks in the lane-order image (a block behind the last one reads zeros)
    unsigned b_off[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int nb = (n0 >> 5) + wn * TN + u;
        b_off[u] = nb < a.wl_nb ? (unsigned)nb * (unsigned)a.wl_ks * 1024u + (unsigned)lane * 16u : OOB_B;
    }

    // The K loop is ONE basic block: every load is issued unconditionally and a K-tile behind the last one is an out-of-range
    // offset (zeros, never consumed).  With branches around the prefetches hipcc parks the 64 accumulator registers in VGPRs
    // across the block boundaries (128 v_accvgpr moves per K-tile) and waits for ALL outstanding loads before the first MFMA.
    u32x4 ra[NA][A_IT];
    f16x8 rb[NB][TN][QS];
    const int nk = a.Kp / BKH;
    int cb = 0, ky = 0, kx = 0;  // wave-uniform K walk of the A loads (ascending K-tiles)
    auto load_a = [&](u32x4 (&ra)[A_IT], int kt) {
        const unsigned delta = (unsigned)((ky * a.dh * a.iw + kx * a.dw) * a.in_ld + cb * BKH) * 2u;
        const int tapbit = ky * a.kw + kx;
        // (zero-padded K axis of a 1x1 conv: a vector behind the last channel reads zeros)
        const bool cok = cb * BKH + kv * 8 < a.icg && kt < nk;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const bool ok = ((a_mask[i] >> tapbit) & 1ull) && cok;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? a_off[i] + delta : OOB_A, 0, 0);
        }
        // branch-free walk: kx, then ky, then the channel block
        ++kx;
        const int wx = kx == a.kw ? 1 : 0;
        kx = wx ? 0 : kx;
        ky += wx;
        const int wy = ky == a.kh ? 1 : 0;
        ky = wy ? 0 : ky;
        cb += wy;
    };
    auto store_a = [&](const u32x4 (&ra)[A_IT], int stage) {
        half_t* As = lds[stage];
        if (BM < RPP && r0 >= BM) return;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<u32x4*>(As + (r0 + RPP * i) * LDH + kv * 8) = ra[i];
    };
    auto load_b = [&](f16x8 (&q)[TN][QS], int kt) {
        // the K-tile's k-steps are consecutive 1 KB blocks: their offset rides in the load's scalar offset; a tile behind the
        // last one is an out-of-range VECTOR offset (the scalar offset takes no part in the range check)
        const bool live = kt < nk;
        const unsigned kb = (unsigned)(live ? kt : 0) * (unsigned)(QS * 1024);
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const unsigned vo = live ? b_off[u] : OOB_B;
#pragma unroll
            for (int s = 0; s < QS; ++s)
                q[u][s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wl, vo + (unsigned)(s * 1024), kb, 0));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.0f;

    load_a(ra[0], 0);
#pragma unroll
    for (int j = 0; j + 1 < NB; ++j) load_b(rb[j], j);
    float bias_pre[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int o = n0 + wn * TN * 32 + l31 + u * 32;
        bias_pre[u] = (a.bias && o < a.ocg) ? a.bias[g * a.ocg + o] : 0.0f;
    }
    store_a(ra[0], 0);
#pragma unroll
    for (int j = 1; j <= NA; ++j) load_a(ra[j % NA], j);   // (ascending: the K walk inside load_a is sequential)
    __syncthreads();

    // one K-tile: B(kt+1) requested, A(kt+1) committed to the other stage and A(kt+2) requested, then the MFMAs of tile kt with
    // the A fragments of step s+1 read while step s multiplies; one barrier
    // PH: kt modulo the unroll factor (a compile-time constant, so every ring slot is a fixed register set)
    auto k_tile = [&](int kt, auto ph) {
        constexpr int PH = decltype(ph)::value;
        constexpr int cur = PH & 1;
        constexpr int sa = (PH + 1) % NA;            // slot of A tile kt + 1: committed to LDS now, then refilled with tile kt + 1 + NA
        const f16x8 (&bcur)[TN][QS] = rb[PH % NB];
        if (!(SI_F16_ABL & 4)) store_a(ra[sa], cur ^ 1);
        if (!(SI_F16_ABL & 2)) load_b(rb[(PH + NB - 1) % NB], kt + NB - 1);
        if (!(SI_F16_ABL & 1)) load_a(ra[sa], kt + 1 + NA);
        // every request of this K-tile is issued before its first MFMA: left to itself the scheduler sinks the A loads (and their
        // address arithmetic) two thirds into the MFMA sequence, a few hundred cycles before the wait that needs them
        __builtin_amdgcn_sched_barrier(0);
        const half_t* As = lds[cur] + (wm * TM * 32 + l31) * LDH + lh * 8;
        f16x8 fa[2][TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[0][t] = *reinterpret_cast<const f16x8*>(As + t * 32 * LDH);
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            if (s + 1 < QS) {
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[(s + 1) & 1][t] = *reinterpret_cast<const f16x8*>(As + t * 32 * LDH + (s + 1) * 16);
            }
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u) {
                    if (SI_F16_ABL & 8) asm volatile("" ::"v"(fa[s & 1][t]), "v"(bcur[u][s]));
                    else acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][t], bcur[u][s], acc[t][u], 0, 0, 0);
                }
        }
        if (!(SI_F16_ABL & 16)) __syncthreads();
    };
    constexpr int UN = (NA == 3 || NB == 3) ? 6 : ((NA == 4 || NB == 4) ? 4 : 2);   // lcm(2, NA, NB) for the depths instantiated
    static_assert(UN % 2 == 0 && UN % NA == 0 && UN % NB == 0, "unroll factor");
    const int nfull = nk / UN;
    for (int kq = 0; kq < nfull; ++kq) {
        const int k0 = kq * UN;
        k_tile(k0, std::integral_constant<int, 0>{});
        k_tile(k0 + 1, std::integral_constant<int, 1>{});
        if constexpr (UN > 2) {
            k_tile(k0 + 2, std::integral_constant<int, 2>{});
            k_tile(k0 + 3, std::integral_constant<int, 3>{});
        }
        if constexpr (UN > 4) {
            k_tile(k0 + 4, std::integral_constant<int, 4>{});
            k_tile(k0 + 5, std::integral_constant<int, 5>{});
        }
    }
    {
        const int k0 = nfull * UN, rem = nk - k0;
        if (rem > 0) k_tile(k0, std::integral_constant<int, 0>{});
        if constexpr (UN > 2) {
            if (rem > 1) k_tile(k0 + 1, std::integral_constant<int, 1>{});
            if (rem > 2) k_tile(k0 + 2, std::integral_constant<int, 2>{});
        }
        if constexpr (UN > 4) {
            if (rem > 3) k_tile(k0 + 3, std::integral_constant<int, 3>{});
            if (rem > 4) k_tile(k0 + 4, std::integral_constant<int, 4>{});
        }
    }

    const int mrow0 = m0 + wm * TM * 32 + 4 * lh, ocol0 = n0 + wn * TN * 32 + l31;
    if (a.ymode) {
        const int img = m0 / a.ohow;
        if (m0 + BM <= a.M && m0 - img * a.ohow + BM <= a.ohow)
            si_yolo_tile_one_image<TM, TN>(a, static_cast<float*>(a.out), acc, mrow0, ocol0, img);
        else
            epilogue_yolo_h<TM, TN>(a, acc, mrow0, ocol0);
    } else if (a.out_f32) epilogue_plain<TM, TN, float>(a, acc, g, mrow0, ocol0);
    else {
        const bool interior = m0 + BM <= a.M;
        if (a.act1 == SI_ACT_SILU && a.act2 == SI_ACT_NONE) epilogue_pick_h<TM, TN, SI_ACT_SILU, SI_ACT_NONE>(a, acc, g, mrow0, ocol0, interior, bias_pre);
        else if (a.act1 == SI_ACT_NONE && a.act2 == SI_ACT_NONE) epilogue_pick_h<TM, TN, SI_ACT_NONE, SI_ACT_NONE>(a, acc, g, mrow0, ocol0, interior, bias_pre);
        else if (a.act1 == SI_ACT_RELU && a.act2 == SI_ACT_NONE) epilogue_pick_h<TM, TN, SI_ACT_RELU, SI_ACT_NONE>(a, acc, g, mrow0, ocol0, interior, bias_pre);
        else if (a.act1 == SI_ACT_NONE && a.act2 == SI_ACT_RELU) epilogue_pick_h<TM, TN, SI_ACT_NONE, SI_ACT_RELU>(a, acc, g, mrow0, ocol0, interior, bias_pre);
        else epilogue_plain<TM, TN, half_t>(a, acc, g, mrow0, ocol0);
    }
}

template <int BM, int BN, int WM, int WN, int BKH, int MINW = 1, int NA = 1, int NB = 2>
int launch_bd(const ConvArgsH& a, int groups, hipStream_t s) {
    ConvArgsH b = a;
    b.m_tiles = (a.M + BM - 1) / BM;
    b.n_tiles = (a.ocg + BN - 1) / BN;
    const int chunks = (b.m_tiles + 7) / 8;
    dim3 grid(chunks * 8 * b.n_tiles, groups, 1);
    hipLaunchKernelGGL((conv_igemm_f16_bd_kernel<BM, BN, WM, WN, BKH, MINW, NA, NB>), grid, dim3(256), 0, s, b);
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int BKH>
int launch_h(const ConvArgsH& a, int groups, hipStream_t s) {
    ConvArgsH b = a;
    b.m_tiles = (a.M + BM - 1) / BM;
    b.n_tiles = (a.ocg + BN - 1) / BN;
    const int chunks = (b.m_tiles + 7) / 8;
    dim3 grid(chunks * 8 * b.n_tiles, groups, 1);
    if (a.up) {
        if constexpr (BM == 64 && BN == 64) hipLaunchKernelGGL((conv_igemm_f16_kernel<BM, BN, WM, WN, BKH, true>), grid, dim3(256), 0, s, b);
        else return SI_E_UNSUPPORTED;
    } else {
        hipLaunchKernelGGL((conv_igemm_f16_kernel<BM, BN, WM, WN, BKH>), grid, dim3(256), 0, s, b);
    }
    return (int)hipGetLastError();
}

// ---- YOLOv5 Detect level as ONE tile shape of its own (round 4) ------------------------------------------------------------------
// A Detect level is a 1x1 conv to na*ne = 255 columns whose decoded fp32 result is 17 x its fp16 input: the launch is a WRITE stream
// (batch 32, 80x80x128: 26 MB in, 209 MB out).  Through the 64x64 tiles above it ran as 12 800 workgroups of two K-tiles each, every
// one a load -> LDS -> barrier -> MFMA -> decode chain ending in 256-byte dword stores at a 1020-byte row pitch: 113 us = 1.85 TB/s.
// Here a workgroup owns 64 CONSECUTIVE PIXELS of one image and ALL columns: in the [H][W][A] row order of the output
// (src/layer/yolo_detect.cpp:223-266 writes the same bytes) those are one contiguous run of 64 * 255 floats.  The whole A tile
// (64 pixels x K) is requested at once, the weights come from the lane-order image (L2 -> registers, four waves x 64 columns), and
// the decoded values are staged through LDS in two 32-pixel halves and leave as 16-byte-per-lane stores down the run.
// Same k order and the same 16-deep MFMA steps as the other fp16 kernels, the same decode expressions: the same bits
// (tests/test_gpu_f16.py).  NCH = K / 128.
// SI_DET_ABL (diagnostic builds only, tools/detect_ablate.sh): bit 0 no sigmoid (bias add only), 1 no global stores, 2 no A loads,
// 3 no MFMA loop, 4 no staging writes -- wrong results, timing only.  0 in the product build.
#ifndef SI_DET_ABL
#define SI_DET_ABL 0
#endif
#ifndef SI_DET_RING
#define SI_DET_RING 4
#endif
#ifndef SI_DET_MINW
#define SI_DET_MINW 3
#endif
template <int NCH>
__global__ __launch_bounds__(256, NCH <= 2 ? SI_DET_MINW : 2) void detect_f16_tile_kernel(const ConvArgsH a) {
    constexpr int K = NCH * 128;
    constexpr int LDH = K + 8;            // halves per LDS row: (2K + 16) mod 128 = 16, the conflict-free pitch of the kernels above
    constexpr int VPR = K / 8;            // 16-byte vectors per row
    constexpr int A_IT = 64 * VPR / 256;  // vectors per thread
    constexpr int KS = K / 16;            // MFMA k-steps
    constexpr int RING = SI_DET_RING;     // weight fragments in flight, in k-steps (K = 128: all of them, requested with the A tile)
    extern __shared__ __attribute__((aligned(16))) unsigned char det_smem[];
    half_t* const As = reinterpret_cast<half_t*>(det_smem);
    float* const stage = reinterpret_cast<float*>(det_smem);   // 32 pixels x 255 floats, over the A tile once the MFMAs are done

    const int tiles_per_img = (a.ohow + 63) >> 6;
    const int img = blockIdx.x / tiles_per_img;
    const int pix0 = (blockIdx.x - img * tiles_per_img) * 64;
    const int valid = min(64, a.ohow - pix0);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(a.wl), 0, (unsigned)a.wl_nb * (unsigned)a.wl_ks * 1024u, 0x00020000);

    u32x4 ra[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / VPR, kv = idx - row * VPR;
        const unsigned off = (unsigned)(img * a.ohow + pix0 + row) * (unsigned)(a.in_ld * 2) + (unsigned)(kv * 16);
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (row < valid && !(SI_DET_ABL & 4)) ? off : OOB_A, 0, 0);
    }
    unsigned b_off[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int nb = wave * 2 + u;
        b_off[u] = nb < a.wl_nb ? (unsigned)nb * (unsigned)a.wl_ks * 1024u + (unsigned)lane * 16u : OOB_B;
    }
    f16x8 rb[RING][2];
#pragma unroll
    for (int s = 0; s < RING; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            rb[s][u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wl, b_off[u], (unsigned)(s * 1024), 0));
    float bv[2];
    int col[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        col[u] = wave * 64 + u * 32 + l31;
        bv[u] = (a.bias && col[u] < a.ocg) ? a.bias[col[u]] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / VPR, kv = idx - row * VPR;
        *reinterpret_cast<u32x4*>(As + row * LDH + kv * 8) = ra[i];
    }
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.0f;
    const half_t* const Ar = As + l31 * LDH + lh * 8;
    f16x8 fa[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) fa[0][t] = *reinterpret_cast<const f16x8*>(Ar + t * 32 * LDH);
#pragma unroll
    for (int s = 0; s < ((SI_DET_ABL & 8) ? 0 : KS); ++s) {
        if (s + 1 < KS) {
#pragma unroll
            for (int t = 0; t < 2; ++t) fa[(s + 1) & 1][t] = *reinterpret_cast<const f16x8*>(Ar + t * 32 * LDH + (s + 1) * 16);
        }
        f16x8 bc[2] = {rb[s % RING][0], rb[s % RING][1]};
        if (s + RING < KS) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                rb[s % RING][u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wl, b_off[u], (unsigned)((s + RING) * 1024), 0));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][t], bc[u], acc[t][u], 0, 0, 0);
    }

    // decode (si_yolo_tile_one_image's expressions) -> LDS -> the contiguous run, 32 pixels at a time
    const int per_pix = a.yna * a.yne;
    const bool coco = a.yna == 3 && a.yne == 85 && valid == 64 && !(SI_DET_ABL & 17);   // workgroup-uniform
    float* const orun = static_cast<float*>(a.out) + ((size_t)img * a.yrows_total + a.yrow_off) * a.yne + (size_t)pix0 * per_pix;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        __syncthreads();   // t = 0: every wave is done reading A; t = 1: the first half has left the stage
        const int rows = min(32, valid - 32 * t);
        if (coco) {
            // the common head (3 anchors x 85 entries, a full tile) with nothing per element but the sigmoid and one LDS write at a
            // compile-time offset: the general form below spent ~25 instructions per element (a select chain the compiler turned
            // into two exec-masked branches, a 32-bit multiply for the row offset, the dead-column test) -- ~35 us of vector issue
            // per CU on level 0, as much as its 209 MB output stream takes.  Box columns (x, y, w, h of an anchor: 0-3, 85-88,
            // 170-173) lie in three of the eight 32-column blocks; the other five never see the box arithmetic.  Same expressions,
            // same bits (tests/test_gpu_f16.py).
#pragma clang fp contract(off)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int blk = wave * 2 + u;   // wave-uniform
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = __builtin_amdgcn_rcpf(1.0f + __expf(-(acc[t][u][e] + bv[u])));
                if (blk == 0 || blk == 2 || blk == 5) {
                    const int oo = col[u] < 255 ? col[u] : 0;
                    const int anc = oo / 85;
                    const int e_ = oo - anc * 85;
                    const bool is_xy = e_ < 2, is_box = e_ < 4;
                    const float* const ap = (is_xy ? a.ygrid + e_ : a.yanchor + (is_box ? e_ - 2 : 0)) + anc * 2 + (size_t)(pix0 + 32 * t + 4 * lh) * 6;
                    float auxv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) auxv[e] = 0.0f;
                    if (is_box) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) auxv[e] = ap[((e & 3) + 8 * (e >> 2)) * 6];
                    }
                    const unsigned mxy = is_xy ? ~0u : 0u, mwh = (is_box && !is_xy) ? ~0u : 0u, msg = is_box ? 0u : ~0u;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float t2 = v[e] * 2.0f;
                        const float xy = (t2 + auxv[e]) * a.ystride;
                        const float wh = t2 * t2 * auxv[e];
                        v[e] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, xy) & mxy) | (__builtin_bit_cast(unsigned, wh) & mwh) |
                                                             (__builtin_bit_cast(unsigned, v[e]) & msg));
                    }
                }
                if (col[u] < 255) {
                    float* const sp = stage + 4 * lh * 255 + col[u];
#pragma unroll
                    for (int e = 0; e < 16; ++e) sp[((e & 3) + 8 * (e >> 2)) * 255] = v[e];
                }
            }
        } else if (rows > 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool live = col[u] < a.ocg;
                const int oo = live ? col[u] : 0;
                const int anc = oo / a.yne;
                const int e_ = oo - anc * a.yne;
                const bool is_xy = e_ < 2, is_box = e_ < 4;
                const float* const ap = (is_xy ? a.ygrid + e_ : a.yanchor + (is_box ? e_ - 2 : 0)) + anc * 2 + (size_t)(pix0 + 32 * t) * a.yna * 2;
                float auxv[16];   // (all sixteen requested before the first is used: si_yolo_tile_one_image)
#pragma unroll
                for (int e = 0; e < 16; ++e) auxv[e] = 0.0f;
                if (is_box) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int dm = (e & 3) + 8 * (e >> 2) + 4 * lh;
                        auxv[e] = ap[(dm < rows ? dm : 0) * a.yna * 2];
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int dm = (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float sg = (SI_DET_ABL & 1) ? acc[t][u][e] + bv[u] : __builtin_amdgcn_rcpf(1.0f + __expf(-(acc[t][u][e] + bv[u])));
                    const float aux = auxv[e];
                    const float t2 = sg * 2.0f;
                    const float xy = (t2 + aux) * a.ystride;
                    const float wh = t2 * t2 * aux;
                    const float v = is_xy ? xy : (is_box ? wh : sg);
                    if (SI_DET_ABL & 16) asm volatile("" ::"v"(v));
                    else if (live) stage[dm * per_pix + oo] = v;
                }
            }
        }
        __syncthreads();
        if (rows > 0 && !(SI_DET_ABL & 2)) {
            float* const dst = orun + (size_t)(32 * t) * per_pix;
            const int nfl = rows * per_pix;
            if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                const int n4 = nfl >> 2;
                for (int i = tid; i < n4; i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(stage)[i];
                const int done = n4 << 2;
                if (tid < nfl - done) dst[done + tid] = stage[done + tid];
            } else {
                for (int i = tid; i < nfl; i += 256) dst[i] = stage[i];
            }
        }
    }
}

// on when the shape allows unless the call's plan says 0 (SiConvPlan::f16_detect_tile: the generic tiles above)
bool detect_tile_on(const SiConv2dDesc* d) {
    const int v = (d && d->plan && d->plan->f16_detect_tile >= 0) ? d->plan->f16_detect_tile : SI_ENV_INT("SI_DETECT_F16_TILE", 1);
    return v != 0;
}
// shape half of the eligibility: a plain pointwise conv over 128 / 256 / 512 channels to at most 256 columns, rows of at most
// 32 x 255 floats per stage half
bool detect_tile_shape_ok(const SiConv2dDesc* d, const SiYoloLevel* y) {
    const bool pointwise = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->pl == 0 && d->ih == d->oh && d->iw == d->ow;
    return pointwise && d->groups == 1 && (d->ic == 128 || d->ic == 256 || d->ic == 512) && d->oc <= 256 && y->na * y->ne == d->oc && !d->has_residual;
}
template <int NCH>
int launch_detect_tile(const ConvArgsH& a, int n, hipStream_t s) {
    constexpr int K = NCH * 128;
    const size_t a_bytes = (size_t)64 * (K + 8) * 2, st_bytes = (size_t)32 * a.ocg * 4;
    const size_t smem = a_bytes > st_bytes ? a_bytes : st_bytes;
    static const int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&detect_f16_tile_kernel<NCH>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (K + 8) * 2 > 32 * 256 * 4 ? 64 * (K + 8) * 2 : 32 * 256 * 4);
    if (attr_rc != 0) return attr_rc;
    const int tiles = ((a.ohow + 63) >> 6) * n;
    hipLaunchKernelGGL((detect_f16_tile_kernel<NCH>), dim3(tiles), dim3(256), smem, s, a);
    return (int)hipGetLastError();
}

// ---- 3x3 conv over ONE 32-channel block as a persistent spatial-tile kernel (round 4, late) ---------------------------------------
// YOLOv5s conv_1 (320x320x32 -> 160x160x64, stride 2, at batch 32: 210 MB in, 105 MB out) ran 102-147 us through EVERY tile shape
// above (profiles/r04_f16_bd_sweep.txt) -- not a pipe: 12 800 workgroups of nine K-tiles, each K-tile a load -> LDS -> barrier round
// trip.  Here a workgroup is persistent and owns TR x 16 output pixels x 32 NBW channels per item: the input patch of item i + 1 is
// requested (global -> registers) before item i is computed and committed to the OTHER LDS buffer after it, so a memory round trip
// is paid once per workgroup, not nine times per tile; all nine taps read their A fragments from the patch at shifted addresses;
// the wave's 18 weight fragments (lane-order image) stay in registers for the workgroup's lifetime.  Waves: 4 / NBW pixel blocks
// (two output rows of 16 each) x NBW 32-channel column blocks.  LDS image, 80-byte pixels: stride 2 -- per patch row the ODD input
// columns (x = 2 ox0 - 1 + 2j, 17 pixels) then the EVEN ones (16), so that the 16 output columns a fragment read spans are
// CONSECUTIVE pixels of a plane, row pitch 2688 B; stride 1 -- 18 pixels, row pitch 1536 B: every ds_read_b128 lane group lands on
// 16 distinct 16-byte slots (searched over the lane groups of MI355X_MICROARCH.md section LDS).  Optional residual (the C3
// bottleneck's shortcut), all of a block's values requested before the matrix loop.
// Same k order (tap-major, 16-deep steps) and the same epilogue expressions as the tiles above: the same bits.
// CB = 64 (stride 1, 64 output channels: the 80x80 C3 bottleneck convs): 144-byte pixels, 2816-byte rows, 36 weight fragments per wave
// (144 registers: two workgroups per CU).
template <int STRIDE, int NBW, int ACT1, bool HAS_RES, int CB = 32>
__global__ __launch_bounds__(256, CB == 64 ? 2 : 3) void conv_c32_patch_f16_kernel(const ConvArgsH a, int tiles_x, int tiles_y, int items) {
    static_assert((STRIDE == 1 || STRIDE == 2) && (NBW == 1 || NBW == 2 || NBW == 4) && (CB == 32 || CB == 64), "instantiated forms");
    constexpr int PITCH = CB * 2 + 16, ROWP = CB == 64 ? (STRIDE == 2 ? 4864 : 2816) : (STRIDE == 2 ? 2688 : 1536), EOFF = 17 * PITCH;   // bytes
    constexpr int TR = 2 * (4 / NBW);                                                 // output rows per item
    constexpr int CH8 = CB / 8, QS = CB / 16, KS = 9 * QS;                            // 16-byte chunks per pixel, k-steps per tap, in all
    constexpr int PR = STRIDE * (TR - 1) + 3, PW = STRIDE * 15 + 3, NCH = PR * PW * CH8, N_IT = (NCH + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char patch[2][PR * ROWP];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / NBW, wn = wave - wm * NBW;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(a.wl), 0, (unsigned)a.wl_nb * (unsigned)a.wl_ks * 1024u, 0x00020000);

    // staging chunks of this thread: chunk c -> (patch row, patch column, 16-byte channel chunk); LDS offsets are item-invariant
    int l_off[N_IT], c_pr[N_IT], c_px[N_IT], c_ch[N_IT];
#pragma unroll
    for (int i = 0; i < N_IT; ++i) {
        const int c = tid + 256 * i;
        const int ch = c % CH8, px = (c / CH8) % PW, pr = (c / CH8) / PW;
        c_pr[i] = pr; c_px[i] = px; c_ch[i] = ch;
        const int pxo = STRIDE == 2 ? ((px & 1) ? EOFF + (px >> 1) * PITCH : (px >> 1) * PITCH) : px * PITCH;
        l_off[i] = c < NCH ? pr * ROWP + pxo + ch * 16 : -1;
    }
    u32x4 rp[N_IT];
    auto prefetch = [&](int item) {
        const int tx = item % tiles_x, t2 = item / tiles_x;
        const int ty = t2 % tiles_y, img = t2 / tiles_y;
        const int y0 = ty * (TR * STRIDE) - 1, x0 = tx * (16 * STRIDE) - 1;
#pragma unroll
        for (int i = 0; i < N_IT; ++i) {
            const int gy = y0 + c_pr[i], gx = x0 + c_px[i];
            const bool ok = l_off[i] >= 0 && (unsigned)gy < (unsigned)a.ih && (unsigned)gx < (unsigned)a.iw;
            const unsigned off = (unsigned)((img * a.ih + gy) * a.iw + gx) * (unsigned)(a.in_ld * 2) + (unsigned)(c_ch[i] * 16);
            rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off : OOB_A, 0, 0);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < N_IT; ++i)
            if (l_off[i] >= 0) *reinterpret_cast<u32x4*>(patch[buf] + l_off[i]) = rp[i];
    };

    int item = blockIdx.x;
    if (item >= items) return;
    prefetch(item);
    // this wave's column block of the weights: 18 k-steps (tap-major, two 16-channel halves per tap), resident in registers
    f16x8 wf[KS];
    {
        const unsigned vo = wn < a.wl_nb ? (unsigned)wn * (unsigned)a.wl_ks * 1024u + (unsigned)lane * 16u : OOB_B;
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wl, vo, (unsigned)(s * 1024), 0));
    }
    const int o = wn * 32 + l31;
    const float bv = (a.bias && o < a.ocg) ? a.bias[o] : 0.0f;
    commit(0);
    // the weights (and the bias) have LANDED before the loop: otherwise every MFMA of the item loop carries a vmcnt(N) wait for "its"
    // weight register -- N counting down to 0 over the 18 / 36 steps -- and, the counter being in order, the last ones wait for the
    // next item's patch loads and this item's output stores as well: a memory round trip per item (seen in the ISA)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();

    // A fragment base: block row = output pixel (2 wm + (l31 >> 4), l31 & 15) of the tile -> patch row STRIDE * that; patch pixel
    // STRIDE * (l31 & 15) (+ kx), which for stride 2 is pixel l31 & 15 of a plane
    const int a_base = (STRIDE * (2 * wm + (l31 >> 4))) * ROWP + (l31 & 15) * PITCH + lh * 16;
    int cur = 0;
    for (; item < items; item += gridDim.x) {
        const int next = item + gridDim.x;
        if (next < items) prefetch(next);
        const int tx = item % tiles_x, t2 = item / tiles_x;
        const int ty = t2 % tiles_y, img = t2 / tiles_y;
        const int oy0 = ty * TR + 2 * wm, ox0 = tx * 16;
        // residual values of this lane's 16 pixels, requested before the matrix loop (a pixel outside the tensor re-reads element 0)
        half_t rv[HAS_RES ? 16 : 1];
        if (HAS_RES) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
                const bool in = oy < a.oh && ox < a.ow && o < a.ocg;
                rv[e] = a.res[in ? (size_t)((img * a.oh + oy) * a.ow + ox) * a.res_ld + o : (size_t)0];
            }
        }
        const unsigned char* const P = patch[cur] + a_base;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        // A fragments in groups of three, the next group requested before this one multiplies: left one by one hipcc keeps a single
        // read ahead of a chain of dependent MFMAs -- an LDS round trip per 32-cycle MFMA
        auto frag = [&](int s) {
            const int tap = s / QS, ky = tap / 3, kx = tap - 3 * ky;
            const int off = ky * ROWP + (STRIDE == 2 ? (kx == 1 ? EOFF : (kx == 2 ? PITCH : 0)) : kx * PITCH) + (s % QS) * 32;
            return *reinterpret_cast<const f16x8*>(P + off);
        };
        constexpr int NG = KS / 3;
        f16x8 fg[2][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) fg[0][i] = frag(i);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int i = 0; i < 3; ++i) fg[(g + 1) & 1][i] = frag(3 * (g + 1) + i);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fg[g & 1][i], wf[3 * g + i], acc, 0, 0, 0);
        }
        // epilogue (epilogue_lean_h's expressions): C/D map col = lane & 31 (channel), row = (e & 3) + 8 (e >> 2) + 4 lh (tile pixel)
        {
            // buffer stores, everything that keeps a value from being written (its pixel outside the tensor, its channel past the last one) as an
            // out-of-range OFFSET: no branch around a store, so the wait in front of the next patch's commit() counts the stores instead of
            // draining them (conv_stem_s2c32_f16.hip BST; LAB_NOTEBOOK R6.10)
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
            const int ow_live = o < a.ocg ? a.ow : 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
                // ACT1 == SiLU: SiLU then nothing (the YOLOv5 forms); otherwise whatever act1 / act2 the layer carries (ResNet's
                // ReLU before or behind the shortcut), as epilogue_plain evaluates them
                float v = ACT1 == SI_ACT_SILU ? act_c<SI_ACT_SILU>(acc[e] + bv, a.act_param) : act_h(a.act1, acc[e] + bv, a.act_param);
                if (HAS_RES) v += (float)rv[e];
                if (ACT1 != SI_ACT_SILU) v = act_h(a.act2, v, a.act_param);
                const unsigned off = ((unsigned)((img * a.oh + oy) * a.ow + ox) * (unsigned)a.out_ld + (unsigned)o) * 2u;
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, si_store_cast<half_t>(v)), rs_out,
                                                      (oy < a.oh && ox < ow_live) ? off : OOB_A, 0, 0);
            }
        }
        if (next < items) {
            commit(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
}

int f16_cu_count();
int f16_forced_variant(const SiConv2dDesc* d);
// on when the shape allows unless the call's plan says 0 (SiConvPlan::f16_s2c32)
bool s2c32_on(const SiConv2dDesc* d) {
    const int v = (d && d->plan && d->plan->f16_s2c32 >= 0) ? d->plan->f16_s2c32 : SI_ENV_INT("SI_CONV_F16_S2C32", 1);
    return v != 0;
}
bool s2c32_shape_ok(const SiConv2dDesc* d) {
    // (64 input channels: only whole 4 x 16 tiles -- ResNet18's 56 x 56 maps, 3.5 tiles wide, measured 2.7 % slower than the generic
    // tiles over the network; the 80 x 80 maps of YOLOv5s 1.4x faster per layer)
    const bool c64 = d->ic == 64 && d->oc == 64 && d->sh == 1 && d->ow % 16 == 0 && d->oh % 4 == 0;
    // 64 -> 128 channels at stride 2 (YOLOv5s conv_7): four column-block waves over one 2 x 16-pixel block
    const bool c64s2 = d->ic == 64 && d->oc == 128 && d->sh == 2 && d->ow % 16 == 0 && d->oh % 2 == 0;
    return d->groups == 1 && ((d->ic == 32 && (d->oc == 32 || d->oc == 64)) || c64 || c64s2) && d->kh == 3 && d->kw == 3 && d->sh == d->sw &&
           (d->sh == 1 || d->sh == 2) && d->dh == 1 && d->dw == 1 && d->pt == 1 && d->pl == 1 &&
           (!d->has_residual || d->res_ld % 2 == 0) &&
           d->oh == (d->ih + 2 - 3) / d->sh + 1 && d->ow == (d->iw + 2 - 3) / d->sw + 1;
}
template <int STRIDE, int NBW, int CB = 32>
int launch_c32_patch(const ConvArgsH& a, const SiConv2dDesc* d, hipStream_t s) {
    constexpr int TR = 2 * (4 / NBW);
    const int tiles_x = (d->ow + 15) / 16, tiles_y = (d->oh + TR - 1) / TR;
    const long long items = (long long)d->n * tiles_x * tiles_y;
    if (items > 0x7fffffffLL || a.out_bytes == 0) return SI_E_UNSUPPORTED;
    auto go = [&](auto kern) {
        const int per_cu = si_resident_blocks(kern, 256, 0);
        long long grid = (long long)f16_cu_count() * per_cu;
        if (grid > items) grid = items;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, s, a, tiles_x, tiles_y, (int)items);
        return (int)hipGetLastError();
    };
    const bool silu = d->act1 == SI_ACT_SILU && d->act2 == SI_ACT_NONE;
    if (d->has_residual) return silu ? go(conv_c32_patch_f16_kernel<STRIDE, NBW, SI_ACT_SILU, true, CB>) : go(conv_c32_patch_f16_kernel<STRIDE, NBW, SI_ACT_NONE, true, CB>);
    return silu ? go(conv_c32_patch_f16_kernel<STRIDE, NBW, SI_ACT_SILU, false, CB>) : go(conv_c32_patch_f16_kernel<STRIDE, NBW, SI_ACT_NONE, false, CB>);
}
int launch_s2c32(const ConvArgsH& a, const SiConv2dDesc* d, hipStream_t s) {
    if (d->ic == 64) return d->sh == 2 ? launch_c32_patch<2, 4, 64>(a, d, s) : launch_c32_patch<1, 2, 64>(a, d, s);
    if (d->sh == 2) return d->oc == 64 ? launch_c32_patch<2, 2>(a, d, s) : launch_c32_patch<2, 1>(a, d, s);
    return d->oc == 64 ? launch_c32_patch<1, 2>(a, d, s) : launch_c32_patch<1, 1>(a, d, s);
}

struct SplitOutH {
    half_t* out2;
    int out2_ld, split;
};

// channel-block size of the K order (a property of the packed weights, so of the layer's static shape only)
// channels per tap in the packed weights: ic / groups, or -- ungrouped convs with a channel count that is a multiple of 8 but
// not of 32 (MobileNet's pointwise and squeeze-excite convs: 16 / 24 / 40 / 72 / 96 ...) -- that rounded up to x32 with zero weights
int f16_icg_pad(const SiConv2dDesc* d) {
    const int icg = d->ic / d->groups;
    return icg % 32 == 0 ? icg : (icg + 31) / 32 * 32;
}
int f16_block(const SiConv2dDesc* d) { return (f16_icg_pad(d) % 64 == 0) ? 64 : 32; }

bool f16_shape_ok(const SiConv2dDesc* d) {
    if (d->groups <= 0 || d->ic % d->groups != 0 || d->oc % d->groups != 0) return false;
    const int icg = d->ic / d->groups;
    if (icg % 32 == 0) return d->kh * d->kw <= 64;
    // (round 5: any ungrouped conv over a multiple of 8 channels -- until then only the 1x1 ones: every tap's channels are zero-padded to
    // whole 32-channel blocks in the weights, and a 16-byte A vector behind the last channel reads zeros, not the next pixel)
    return d->kh * d->kw <= 64 && d->groups == 1 && icg % 8 == 0 && icg >= 8;
}

// extents of the lane-order weight image (per group): 32-channel column blocks x 16-deep k-steps x 1 KB
int f16_lane_nb(const SiConv2dDesc* d) { return (d->oc / d->groups + 31) / 32; }
int f16_lane_ks(const SiConv2dDesc* d) { return d->kh * d->kw * f16_icg_pad(d) / 16; }
size_t f16_row_elems(const SiConv2dDesc* d) { return (size_t)d->oc * d->kh * d->kw * f16_icg_pad(d); }
size_t f16_lane_elems(const SiConv2dDesc* d) { return (size_t)d->groups * f16_lane_nb(d) * f16_lane_ks(d) * 512; }

// Tile variants.  One-stage kernel (both operands through LDS): 0: 64x64, 1: 128x64, 2: 128x128.  "bd" kernels (weights straight
// from L2 in lane order, A through two LDS stages): 3: 128x128 as 2x2 waves of 64x64 (the form the ablations of
// profiles/r04_f16_ablation.txt were taken on), 7: 128x32 as 4x1 waves, 9: 128x128 as 1x4 waves of 128x32 (every wave its own weight
// columns, A shared through LDS), 10: 64x128 as 1x4 waves of 64x32, 11: 9 compiled for three waves per SIMD.  Measured and retired
// (profiles/r04_f16_bd_sweep.txt; the ids are not accepted): 4 128x64 2x2, 5 64x128 2x2, 6 64x64 2x2, 8 256x64 4x1, 12 64x256 1x4,
// 13 32x128 1x4, 14 32x256 1x4.  All variants produce the same bits (same k order, same 16-deep MFMA steps).
// SiConvPlan::f16_tile forces one for a call (tests, sweeps); -1: the policy.
// Also measured and retired (round 4, late; profiles/r04_f16_deep_rings.txt): 10 / 9 / 7 with deeper prefetch rings (NA 2 or 4 register
// slots for A, NB 4 for B -- the kernel template still takes the depths): never faster, 3-60 % slower; what these launches wait for is not
// bytes in flight.
constexpr int kF16Variants = 12;
bool f16_variant_valid(int v) { return v >= 0 && v < kF16Variants && v != 4 && v != 5 && v != 6 && v != 8; }
// a forced tile comes with the call (SiConvPlan::f16_tile); -1: the policy
int f16_forced_variant(const SiConv2dDesc* d) {
    int v = (d && d->plan) ? d->plan->f16_tile : -1;
    if (v < 0) v = SI_ENV_INT("SI_CONV_F16_VARIANT", -1);
    return f16_variant_valid(v) ? v : -1;
}
int f16_cu_count() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}
// The policy (round 4; tools/f16_sweep.sh on MI355X, sustained, every YOLOv5s shape at batch 32, two boxes:
// profiles/r04_f16_bd_sweep.txt).  Whatever is chosen the bits are the same (tests/test_gpu_f16.py), so it may look at M:
//   * <= 32 output channels: 128x32 as four 32x32 waves with lane-order weights (160x160x32 3x3 61 -> 53 us, 1x1 22 -> 19);
//   * 64 output channels, or a short K with few columns: the one-stage 64x64 kernel (these layers are HBM-bound; every
//     lane-order tile measured slower);
//   * K >= 512 and >= 128 columns (the 3x3 layers from 64 input channels on, the wide 1x1 layers), or K >= 256 with >= 256
//     columns: 64x128 as four 64x32 waves side by side (A shared through LDS, every wave its own weight columns, 4 workgroups
//     per CU) -- 20x20x256 3x3 35 -> 25 us, 80x80x128 -> 40x40x256 s2 60 -> 47; with >= 512 columns and enough 128-row tiles to
//     cover the chip, 128x128 as four 128x32 waves (40x40x256 -> 20x20x512 s2 60 -> 45 us, 20x20x1024 -> 512 1x1 29 -> 23).
int f16_variant(const SiConv2dDesc* d) {
    const int forced = f16_forced_variant(d);
    if (forced >= 0) return forced;
    static const bool policy_on = SI_ENV_INT("SI_CONV_F16_POLICY", 1) != 0;   // (experiment build: 0 restores "64x64 everywhere")
    if (!policy_on) return 0;
    const int ocg = d->oc / d->groups;
    const long long K = (long long)d->kh * d->kw * f16_icg_pad(d);
    const long long M = (long long)d->n * d->oh * d->ow;
    if (ocg <= 32) return 7;
    if (ocg < 128) return 0;
    const bool wide = K >= 512 || (K >= 256 && ocg >= 256);
    if (!wide) return 0;
    const long long tiles128 = ((M + 127) / 128) * ((ocg + 127) / 128) * d->groups;
    if (ocg >= 512 && tiles128 >= f16_cu_count()) return 9;
    return 10;
}

// what si_hip_conv2d_upcat_f16 requires of a problem apart from the pointers' alignment (shape only): a plain pointwise conv,
// whole K blocks on both sides of the seam, both tensors below 4 GiB
bool f16_upcat_shape_ok(const SiConv2dDesc* d, const SiConv2dUpsampledSource* up) {
    if (!d || !up || d->groups != 1 || d->ic <= 0 || d->oc <= 0 || !f16_shape_ok(d)) return false;
    const bool pointwise = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->pl == 0 && d->ih == d->oh && d->iw == d->ow;
    if (!pointwise || d->has_residual || d->ic % 32 != 0) return false;
    const int blk = f16_block(d);
    if (up->c <= 0 || up->c % blk != 0 || up->c0 % blk != 0 || up->c0 + up->c > d->ic || up->ld % 8 != 0 || up->ih <= 0 || up->iw <= 0) return false;
    if (d->in_ld % 8 != 0) return false;
    const unsigned long long ub = (unsigned long long)d->n * up->ih * up->iw * up->ld * 2ull;
    return ub < 0xFFFFFF00ull;
}

int dispatch_h(const SiConv2dDesc* d, const void* in, const void* w_packed, const float* bias, const void* residual,
               void* out, int out_f32, si_stream_t stream, const SiYoloLevel* yolo, const float* ygrid, const float* yanchor,
               const SplitOutH* split, const SiConv2dUpsampledSource* up = nullptr) {
    if (!d || !in || !w_packed || !out) return SI_E_BADARG;
    if (d->n <= 0 || d->oh <= 0 || d->ow <= 0) return SI_E_BADARG;
    if (d->has_bias && !bias) return SI_E_BADARG;
    if (d->has_residual && !residual) return SI_E_BADARG;
    if (!f16_shape_ok(d)) return SI_E_UNSUPPORTED;
    if (d->plan && d->plan->f16_tile >= 0 && !f16_variant_valid(d->plan->f16_tile)) return SI_E_BADARG;   // an unknown or retired tile id
    if (d->in_ld % 8 != 0 || (reinterpret_cast<uintptr_t>(in) & 15) != 0) return SI_E_UNSUPPORTED;
    if ((long long)d->n * d->oh * d->ow > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->ih * d->iw * d->in_ld * 2ull;
    const unsigned long long w_bytes = (unsigned long long)(f16_row_elems(d) + f16_lane_elems(d)) * 2ull;
    if (in_bytes >= 0xFFFFFF00ull || w_bytes >= 0x40000000ull) return SI_E_UNSUPPORTED;

    ConvArgsH a;
    a.in = static_cast<const half_t*>(in);
    a.w = static_cast<const half_t*>(w_packed);
    a.wl = a.w + f16_row_elems(d);   // the lane-order image sits behind the row-major one
    a.wl_nb = f16_lane_nb(d);
    a.wl_ks = f16_lane_ks(d);
    a.bias = d->has_bias ? bias : nullptr;
    a.res = d->has_residual ? static_cast<const half_t*>(residual) : nullptr;
    a.out = out;
    a.ih = d->ih; a.iw = d->iw; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.dh = d->dh; a.dw = d->dw;
    a.pt = d->pt; a.pl = d->pl;
    a.icg = d->ic / d->groups;
    a.ocg = d->oc / d->groups;
    a.oc = d->oc;
    a.Kp = d->kh * d->kw * f16_icg_pad(d);
    a.M = d->n * d->oh * d->ow;
    a.ohow = d->oh * d->ow;
    a.mg_ohow = a.ohow > 1 ? (unsigned)(0x100000000ull / (unsigned)a.ohow) : 0xFFFFFFFFu;
    a.mg_ow = d->ow > 1 ? (unsigned)(0x100000000ull / (unsigned)d->ow) : 0xFFFFFFFFu;
    a.m_tiles = a.n_tiles = 0;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    a.in_bytes = (unsigned)in_bytes;
    {
        const unsigned long long ob = (((unsigned long long)d->n * d->oh * d->ow - 1) * d->out_ld + d->oc) * (out_f32 ? 4ull : 2ull);
        a.out_bytes = ob < 0xFFFFFF00ull ? (unsigned)ob : 0u;
    }
    a.pointwise = (d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->pl == 0 && d->ih == d->oh && d->iw == d->ow) ? 1 : 0;
    a.out_f32 = out_f32;
    a.out2 = nullptr; a.out2_ld = 0; a.split = 0;
    a.ymode = 0; a.yna = a.yne = a.yrows_total = a.yrow_off = 0; a.ystride = 0.f; a.ygrid = a.yanchor = nullptr;
    a.up = nullptr; a.up_ih = a.up_iw = a.up_ld = a.up_cb0 = a.up_cb1 = 0; a.up_inv_h = a.up_inv_w = 0.f; a.up_bytes = 0;
    if (up) {
        // channels [c0, c0 + c) of this 1x1 conv's input are nn.Upsample(nearest) of up->src: read them at the source
        if (!up->src || yolo || !f16_upcat_shape_ok(d, up) || (reinterpret_cast<uintptr_t>(up->src) & 15) != 0) return SI_E_UNSUPPORTED;
        const int blk = f16_block(d);
        a.up = reinterpret_cast<const half_t*>(up->src);
        a.up_ih = up->ih; a.up_iw = up->iw; a.up_ld = up->ld; a.up_cb0 = up->c0 / blk; a.up_cb1 = (up->c0 + up->c) / blk;
        a.up_inv_h = up->inv_scale_h; a.up_inv_w = up->inv_scale_w;
        a.up_bytes = (unsigned)((unsigned long long)d->n * up->ih * up->iw * up->ld * 2ull);
    }
    if (split) {
        if (d->groups != 1 || out_f32 || split->split <= 0 || split->split >= d->oc || split->split % 32 != 0 || !split->out2)
            return SI_E_BADARG;
        a.out2 = split->out2; a.out2_ld = split->out2_ld; a.split = split->split;
    }
    if (yolo) {
        if (d->groups != 1 || d->has_residual || yolo->na * yolo->ne != d->oc) return SI_E_UNSUPPORTED;
        a.ymode = 1; a.yna = yolo->na; a.yne = yolo->ne; a.yrows_total = yolo->rows_total; a.yrow_off = yolo->row_off;
        a.ystride = yolo->stride; a.ygrid = ygrid; a.yanchor = yanchor;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (yolo && detect_tile_on(d) && detect_tile_shape_ok(d, yolo)) {
        switch (d->ic) {
            case 128: return launch_detect_tile<1>(a, d->n, s);
            case 256: return launch_detect_tile<2>(a, d->n, s);
            default: return launch_detect_tile<4>(a, d->n, s);
        }
    }
    if (!up && !yolo && !split && !out_f32 && s2c32_on(d) && f16_forced_variant(d) < 0 && s2c32_shape_ok(d) && a.out_bytes != 0 &&
        (!d->has_residual || (reinterpret_cast<uintptr_t>(residual) & 1) == 0))
        return launch_s2c32(a, d, s);
    if (!up && !yolo && !split && !out_f32 && f16_forced_variant(d) < 0 && si_conv_slab_f16_ok(d) && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
        (!d->has_bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0) && (!d->has_residual || (reinterpret_cast<uintptr_t>(residual) & 7) == 0))
        return si_conv_slab_f16_launch(d, in, a.wl, a.wl_nb, a.wl_ks, bias, residual, out, s);

    const int v = up ? 0 : f16_variant(d);   // (the dual-source form lives in the one-stage 64x64 kernel)
    if (f16_block(d) == 64) {
        switch (v) {
            case 1: return launch_h<128, 64, 2, 2, 64>(a, d->groups, s);
            case 2: return launch_h<128, 128, 2, 2, 64>(a, d->groups, s);
            case 3: return launch_bd<128, 128, 2, 2, 64>(a, d->groups, s);
            case 7: return launch_bd<128, 32, 4, 1, 64>(a, d->groups, s);
            case 9: return launch_bd<128, 128, 1, 4, 64>(a, d->groups, s);
            case 10: return launch_bd<64, 128, 1, 4, 64>(a, d->groups, s);
            case 11: return launch_bd<128, 128, 1, 4, 64, 3>(a, d->groups, s);
            default: return launch_h<64, 64, 2, 2, 64>(a, d->groups, s);
        }
    }
    switch (v) {
        case 1: return launch_h<128, 64, 2, 2, 32>(a, d->groups, s);
        case 2: return launch_h<128, 128, 2, 2, 32>(a, d->groups, s);
        case 3: return launch_bd<128, 128, 2, 2, 32>(a, d->groups, s);
        case 7: return launch_bd<128, 32, 4, 1, 32>(a, d->groups, s);
        case 9: return launch_bd<128, 128, 1, 4, 32>(a, d->groups, s);
        case 10: return launch_bd<64, 128, 1, 4, 32>(a, d->groups, s);
        case 11: return launch_bd<128, 128, 1, 4, 32, 3>(a, d->groups, s);
        default: return launch_h<64, 64, 2, 2, 32>(a, d->groups, s);
    }
}

}  // namespace

extern "C" {

int si_hip_f32_to_f16_host(const float* src, void* dst, size_t n) {
    if ((!src || !dst) && n) return SI_E_BADARG;
    half_t* d = static_cast<half_t*>(dst);
    for (size_t i = 0; i < n; ++i) d[i] = (half_t)src[i];  // round to nearest even
    return 0;
}

int si_hip_f16_to_f32_host(const void* src, float* dst, size_t n) {
    if ((!src || !dst) && n) return SI_E_BADARG;
    const half_t* s = static_cast<const half_t*>(src);
    for (size_t i = 0; i < n; ++i) dst[i] = (float)s[i];
    return 0;
}

int si_hip_conv2d_f16_supported(const SiConv2dDesc* d) {
    if (!d) return 0;
    if (si_conv_stem_f16_ok(d)) return 2;  // stem: fp32 image in, fp16 out, weights from si_hip_conv2d_stem_f16_pack_weight_host
    if (si_conv_depthwise_f16_ok(d)) return 3;  // depthwise: si_hip_conv2d_depthwise_f16, fp32 weights in the fp32 depthwise layout
    return f16_shape_ok(d) ? 1 : 0;
}

size_t si_hip_conv2d_f16_weight_elems(const SiConv2dDesc* d) {
    if (!d || !f16_shape_ok(d)) return 0;
    return f16_row_elems(d) + f16_lane_elems(d);   // two images of the same weights: [oc][K] rows, then MFMA lane order
}

int si_hip_conv2d_f16_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, void* w_packed) {
    if (!d || !w_oihw || !w_packed) return SI_E_BADARG;
    if (!f16_shape_ok(d)) return SI_E_UNSUPPORTED;
    const int icg = d->ic / d->groups, icp = f16_icg_pad(d);
    const int ntaps = d->kh * d->kw;
    const int blk = f16_block(d);
    half_t* w = static_cast<half_t*>(w_packed);
    for (int o = 0; o < d->oc; ++o)
        for (int y = 0; y < d->kh; ++y)
            for (int x = 0; x < d->kw; ++x) {
                const int tap = y * d->kw + x;
                for (int c = 0; c < icp; ++c) {
                    const float v = c < icg ? w_oihw[(((size_t)o * icg + c) * d->kh + y) * d->kw + x] : 0.0f;
                    const size_t k = ((size_t)(c / blk) * ntaps + tap) * blk + (c % blk);  // (c/blk, kh, kw, c%blk)
                    w[(size_t)o * ntaps * icp + k] = (half_t)v;
                }
            }
    // second image, MFMA B-operand lane order (conv_igemm_f16_bd_kernel): per group [column block nb][k-step ks][lane][8 halves],
    // lane l = W[o = nb*32 + (l & 31)][k = ks*16 + 8*(l >> 5) .. + 7] in the same K order; channels behind the last one are zeros
    const int ocg = d->oc / d->groups, nb_n = f16_lane_nb(d), ks_n = f16_lane_ks(d);
    const size_t Kp = (size_t)ntaps * icp;
    half_t* wl = w + f16_row_elems(d);
    for (int g = 0; g < d->groups; ++g)
        for (int nb = 0; nb < nb_n; ++nb)
            for (int ks = 0; ks < ks_n; ++ks)
                for (int l = 0; l < 64; ++l) {
                    const int o = nb * 32 + (l & 31);
                    half_t* dst = wl + ((((size_t)g * nb_n + nb) * ks_n + ks) * 64 + l) * 8;
                    for (int j = 0; j < 8; ++j)
                        dst[j] = o < ocg ? w[((size_t)g * ocg + o) * Kp + (size_t)ks * 16 + 8 * (l >> 5) + j] : (half_t)0.0f;
                }
    return 0;
}

int si_hip_conv2d_f16_tile_variant(const SiConv2dDesc* d) {
    if (!d || !f16_shape_ok(d)) return -1;
    return f16_variant(d);
}

const char* si_hip_conv2d_f16_kernel_name(const SiConv2dDesc* d, int form) {
    // the instantiation dispatch_h launches for this problem, exactly as rocprofv3 prints it (minus the namespace); form 1: the
    // dual-source (upsampled) form, which lives in the one-stage 64x64 kernel
    if (!d || !f16_shape_ok(d)) return "";
    if (form == 0 && s2c32_on(d) && f16_forced_variant(d) < 0 && s2c32_shape_ok(d))
    {
        static const char* const names[16] = {
            "conv_c32_patch_f16_kernel<1, 1, 0, false, 32>", "conv_c32_patch_f16_kernel<1, 1, 0, true, 32>", "conv_c32_patch_f16_kernel<1, 1, 2, false, 32>", "conv_c32_patch_f16_kernel<1, 1, 2, true, 32>",
            "conv_c32_patch_f16_kernel<1, 2, 0, false, 32>", "conv_c32_patch_f16_kernel<1, 2, 0, true, 32>", "conv_c32_patch_f16_kernel<1, 2, 2, false, 32>", "conv_c32_patch_f16_kernel<1, 2, 2, true, 32>",
            "conv_c32_patch_f16_kernel<2, 1, 0, false, 32>", "conv_c32_patch_f16_kernel<2, 1, 0, true, 32>", "conv_c32_patch_f16_kernel<2, 1, 2, false, 32>", "conv_c32_patch_f16_kernel<2, 1, 2, true, 32>",
            "conv_c32_patch_f16_kernel<2, 2, 0, false, 32>", "conv_c32_patch_f16_kernel<2, 2, 0, true, 32>", "conv_c32_patch_f16_kernel<2, 2, 2, false, 32>", "conv_c32_patch_f16_kernel<2, 2, 2, true, 32>"};
        static const char* const names64[4] = {"conv_c32_patch_f16_kernel<1, 2, 0, false, 64>", "conv_c32_patch_f16_kernel<1, 2, 0, true, 64>",
                                               "conv_c32_patch_f16_kernel<1, 2, 2, false, 64>", "conv_c32_patch_f16_kernel<1, 2, 2, true, 64>"};
        const bool silu = d->act1 == SI_ACT_SILU && d->act2 == SI_ACT_NONE;
        static const char* const names64s2[4] = {"conv_c32_patch_f16_kernel<2, 4, 0, false, 64>", "conv_c32_patch_f16_kernel<2, 4, 0, true, 64>",
                                                 "conv_c32_patch_f16_kernel<2, 4, 2, false, 64>", "conv_c32_patch_f16_kernel<2, 4, 2, true, 64>"};
        if (d->ic == 64) return (d->sh == 2 ? names64s2 : names64)[(silu ? 2 : 0) + (d->has_residual ? 1 : 0)];
        return names[(d->sh - 1) * 8 + (d->oc == 64 ? 4 : 0) + (silu ? 2 : 0) + (d->has_residual ? 1 : 0)];
    }
    if (form == 0 && f16_forced_variant(d) < 0 && si_conv_slab_f16_ok(d)) return si_conv_slab_f16_name(d);
    const int v = form == 1 ? 0 : f16_variant(d);
    const bool b64 = f16_block(d) == 64;
    switch (v) {
        case 1: return b64 ? "conv_igemm_f16_kernel<128, 64, 2, 2, 64, false>" : "conv_igemm_f16_kernel<128, 64, 2, 2, 32, false>";
        case 2: return b64 ? "conv_igemm_f16_kernel<128, 128, 2, 2, 64, false>" : "conv_igemm_f16_kernel<128, 128, 2, 2, 32, false>";
        case 3: return b64 ? "conv_igemm_f16_bd_kernel<128, 128, 2, 2, 64, 1, 1, 2>" : "conv_igemm_f16_bd_kernel<128, 128, 2, 2, 32, 1, 1, 2>";
        case 7: return b64 ? "conv_igemm_f16_bd_kernel<128, 32, 4, 1, 64, 1, 1, 2>" : "conv_igemm_f16_bd_kernel<128, 32, 4, 1, 32, 1, 1, 2>";
        case 9: return b64 ? "conv_igemm_f16_bd_kernel<128, 128, 1, 4, 64, 1, 1, 2>" : "conv_igemm_f16_bd_kernel<128, 128, 1, 4, 32, 1, 1, 2>";
        case 10: return b64 ? "conv_igemm_f16_bd_kernel<64, 128, 1, 4, 64, 1, 1, 2>" : "conv_igemm_f16_bd_kernel<64, 128, 1, 4, 32, 1, 1, 2>";
        case 11: return b64 ? "conv_igemm_f16_bd_kernel<128, 128, 1, 4, 64, 3, 1, 2>" : "conv_igemm_f16_bd_kernel<128, 128, 1, 4, 32, 3, 1, 2>";
        default:
            if (form == 1) return b64 ? "conv_igemm_f16_kernel<64, 64, 2, 2, 64, true>" : "conv_igemm_f16_kernel<64, 64, 2, 2, 32, true>";
            return b64 ? "conv_igemm_f16_kernel<64, 64, 2, 2, 64, false>" : "conv_igemm_f16_kernel<64, 64, 2, 2, 32, false>";
    }
}

int si_hip_conv2d_f16(const SiConv2dDesc* d, const void* in, const void* w_packed, const float* bias, const void* residual,
                      void* out, int out_is_f32, si_stream_t stream) {
    return dispatch_h(d, in, w_packed, bias, residual, out, out_is_f32 ? 1 : 0, stream, nullptr, nullptr, nullptr, nullptr);
}

int si_hip_conv2d_split_f16(const SiConv2dDesc* d, const void* in, const void* w_packed, const float* bias, void* out,
                            int split_oc, void* out2, int out2_ld, si_stream_t stream) {
    if (!d || d->has_residual) return SI_E_BADARG;
    SplitOutH sp{static_cast<half_t*>(out2), out2_ld, split_oc};
    return dispatch_h(d, in, w_packed, bias, nullptr, out, 0, stream, nullptr, nullptr, nullptr, &sp);
}

int si_hip_conv2d_upcat_f16(const SiConv2dDesc* d, const void* in, const SiConv2dUpsampledSource* up, const void* w_packed,
                            const float* bias, void* out, int split_oc, void* out2, int out2_ld, si_stream_t stream) {
    if (!d || !up || d->has_residual) return SI_E_BADARG;
    if (split_oc > 0) {
        SplitOutH sp{static_cast<half_t*>(out2), out2_ld, split_oc};
        return dispatch_h(d, in, w_packed, bias, nullptr, out, 0, stream, nullptr, nullptr, nullptr, &sp, up);
    }
    return dispatch_h(d, in, w_packed, bias, nullptr, out, 0, stream, nullptr, nullptr, nullptr, nullptr, up);
}

int si_hip_conv2d_upcat_f16_supported(const SiConv2dDesc* d, const SiConv2dUpsampledSource* up) { return f16_upcat_shape_ok(d, up) ? 1 : 0; }

int si_hip_conv2d_yolo_f16(const SiConv2dDesc* d, const void* in, const void* w_packed, const float* bias,
                           const SiYoloLevel* level, const float* grid_hwa2, const float* anchor_hwa2, float* detect_out,
                           si_stream_t stream) {
    if (!level || !grid_hwa2 || !anchor_hwa2 || level->ne < 4 || level->na <= 0) return SI_E_BADARG;
    return dispatch_h(d, in, w_packed, bias, nullptr, detect_out, 0, stream, level, grid_hwa2, anchor_hwa2, nullptr);
}

int si_hip_conv2d_yolo_f16_tile(const SiConv2dDesc* d, const SiYoloLevel* level) {
    if (!d || !level) return 0;
    return (detect_tile_on(d) && f16_shape_ok(d) && detect_tile_shape_ok(d, level)) ? 1 : 0;
}

}  // extern "C"
