// conv_igemm.hip -- NHWC fp32 convolution as an implicit GEMM on the gfx950
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
// Replaces Conv2d::ForwardIm2Col / ForwardIm2ColWithGroup / ForwardWinograd23
// of the reference (src/layer/conv_2d.cpp:207-283, :285-380, :382-487) plus
// the passes it runs afterwards (AddBiasNHWC src/layer/simd/binary.cpp:38-53,
// SiLU/ReLU src/layer/silu.cpp:49-62, residual BinaryOp
// src/layer/binary_op.cpp:52-94) with ONE kernel:
//
//     out[m, o] = act2( act1( sum_k A[m,k] * W[o,k] + bias[o] ) + res[m, o] )
//
//   m = (n, oh, ow) flattened       M = N*OH*OW
//   k = (kh, kw, c) flattened       K = KH*KW*ICGp   (ICGp = ic/groups padded to x4)
//   A[m,k] = in[n, oh*sh-pt+kh*dh, ow*sw-pl+kw*dw, g*icg + c]   (0 outside the image)
//
// The im2col matrix is never materialised: each workgroup gathers a BM x 32
// slice of A and a BN x 32 slice of W with 16-byte loads (channels are the
// fastest NHWC axis, so a K-vector of 4 is 4 consecutive channels of one tap),
// stages both through LDS as [row][32+4] (the +4 pad makes ds_write_b128 and
// the ds_read_b128 fragment reads conflict free: row stride 36 dwords puts 16
// consecutive rows on 16 distinct 4-bank groups of the 64-bank array), and
// runs 32x32x2 MFMAs on them.  Global loads for K-tile t+1 are issued before
// the MFMAs of tile t (register prefetch + two LDS buffers, one barrier per
// K-tile); 2 workgroups per CU cover each other's barrier.
//
// Fragment trick: one ds_read_b128 per lane fetches A[row][8q + 4h .. +3]
// (h = lane>>5).  MFMA j (j=0..3) of that group consumes register j, i.e. the
// k pair {8q+j, 8q+4+j}: lanes 0-31 supply the first k, lanes 32-63 the second,
// for A and W alike, so the products line up (a K permutation inside the
// tile, not a different sum).
#include <hip/hip_runtime.h>

#include <array>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "si_hip.h"
#include "si_hip_internal.h"

// No floating-point contraction in this file: the epilogue exists in several template instantiations (interior / edge
// tile, with / without residual, activation known or not) and an image's result must not depend on which one a pixel
// happens to go through (bit-exact batch sharding).  The MFMA builtins are unaffected.
#pragma clang fp contract(off)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// conv_smallc.hip: stem kernel for 1..3 input channels
bool si_conv_smallc_ok(const SiConv2dDesc* d);
size_t si_conv_smallc_weight_elems(const SiConv2dDesc* d);
void si_conv_smallc_pack(const SiConv2dDesc* d, const float* w_oihw, float* w_packed);
const char* si_conv_smallc_name(const SiConv2dDesc* d);
int si_conv_smallc_launch(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                          const float* residual, float* out, hipStream_t s);

// conv_depthwise.hip: groups == ic == oc (HBM-bound, no MFMA)
bool si_conv_depthwise_ok(const SiConv2dDesc* d);
size_t si_conv_depthwise_weight_elems(const SiConv2dDesc* d);
void si_conv_depthwise_pack(const SiConv2dDesc* d, const float* w_oihw, float* w_packed);
const char* si_conv_depthwise_name(const SiConv2dDesc* d);
int si_conv_depthwise_launch(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                             const float* residual, float* out, hipStream_t s);

namespace {

struct ConvArgs {
    const float* in;
    const float* w;
    const float* bias;
    const float* res;
    float* out;
    int ih, iw, in_ld;
    int oh, ow, out_ld, res_ld;
    int kh, kw, sh, sw, dh, dw, pt, pl;
    int icg, icg_pad, ocg, oc;  // per-group channels
    int Kp;                     // kh*kw*icg_pad
    int M;                      // n*oh*ow
    int ohow;
    unsigned mg_ohow, mg_ow;    // floor(2^32 / d) for the two per-row divisions of the fast kernel's prologue (see fast_div)
    unsigned mg_chunk;          // ... and for the block decode's division by 8 * n_tiles
    int m_tiles, n_tiles;
    int act1, act2;
    float act_param;
    int cb_major;               // K order: 0 = (tap, c), 1 = (c/32, tap, c%32)
    int pointwise;              // 1x1 stride 1 pad 0: the A matrix is the input tensor itself
    int ntaps;                  // kh*kw
    unsigned in_bytes, w_bytes; // buffer-resource extents for the fast path (tensor < 4 GB)
    // split output (two sibling convs on one input fused along oc): channels >= split go to out2 (stride out2_ld)
    float* out2;
    int out2_ld, split;
    // YOLOv5 Detect decode fused into the epilogue (ymode != 0): out is the [N][rows_total][ne] detect tensor
    int ymode, yna, yne, yrows_total, yrow_off;
    float ystride;
    const float* ygrid;         // [oh*ow*na][2]
    const float* yanchor;       // [oh*ow*na][2]
    // dual-source input of a 1x1 conv that consumes cat(..., nn.Upsample(x), ...): 32-channel blocks [up_cb0, up_cb1) of the K
    // axis are read from the LOW-RESOLUTION tensor `up` at the nearest-neighbour source pixel (reference index rule,
    // src/layer/upsample.cpp:85-92) instead of from `in`; the upsampled tensor is never written
    const float* up;
    int up_ih, up_iw, up_ld, up_cb0, up_cb1;
    float up_inv_h, up_inv_w;
    unsigned up_bytes;
};

__device__ __forceinline__ float apply_act(int act, float v, float p) {
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

// The two fp32 MFMA tiles the kernels use.  Both round like ONE sequential fma chain over k (tools/mfma_chain_test.hip, measured:
// v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32 and a scalar fmaf loop agree bit for bit), so a kernel on 16x16 tiles that feeds
// k in the same order as one on 32x32 tiles produces the same bits per output element -- the tile policy may follow the batch
// size without breaking the engine's batch-invariance contract.
//   32x32x2: A/B operand of lane l = row (l & 31), k = l >> 5;   C/D: col = l & 31, row = (e&3) + 8*(e>>2) + 4*(l>>5), 16 registers
//   16x16x4: A/B operand of lane l = row (l & 15), k = l >> 4;   C/D: col = l & 15, row = e + 4*(l>>4),             4 registers
template <int MT> struct Mma;
template <> struct Mma<32> {
    typedef f32x16 acc_t;
    static constexpr int NE = 16;
    __device__ static __forceinline__ constexpr int row(int e) { return (e & 3) + 8 * (e >> 2); }
};
template <> struct Mma<16> {
    typedef f32x4 acc_t;
    static constexpr int NE = 4;
    __device__ static __forceinline__ constexpr int row(int e) { return e; }
};

// ---- epilogue shared by the kernels.  32x32 C/D map: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
// Each store instruction writes two 128-byte row segments (lanes 0-31 / 32-63).  The activation switch
// is resolved once per workgroup, not per element.
// v_rcp_f32 / v_exp_f32 are accurate to 1 ulp: SiLU and sigmoid are within ~2e-7 relative of the exact value, three orders
// inside the parity bar, at 5 instructions instead of the 15 of an IEEE division.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <int ACT>
__device__ __forceinline__ float act_fn(float v, float p) {
    if (ACT == SI_ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == SI_ACT_SILU) return v * fast_rcp(1.0f + __expf(-v));
    if (ACT == SI_ACT_SIGMOID) return fast_rcp(1.0f + __expf(-v));
    if (ACT == SI_ACT_HARDSIGMOID) return fminf(fmaxf(v * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f);
    if (ACT == SI_ACT_HARDSWISH) return v * fminf(fmaxf(v * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f);
    if (ACT == SI_ACT_LEAKYRELU) return v > 0.0f ? v : v * p;
    return v;
}

// The straight-line epilogue: one activation known at compile time, residual yes / no and "every row of the tile is inside
// M" (all tiles but the last) resolved once per workgroup, row offsets c*ld computed once -- per element this leaves
// bias add, activation, (residual load + add), one store.
template <int MT, int TM, int TN, int ACT1, int ACT2, bool HAS_RES, bool INTERIOR>
__device__ __forceinline__ void epilogue_lean(const ConvArgs& a, typename Mma<MT>::acc_t (&acc)[TM][TN], int g, int mrow0, int ocol0,
                                              const float* bias_pre) {
    constexpr int NE = Mma<MT>::NE;
    if constexpr (MT == 16 && TN % 2 == 0) {
        // 16x16 tiles in PAIRS along the channels: a lane holds, per register, one value of the left tile (A) and one of the right
        // tile (B) for row e + 4g (g = its 16-lane group).  v_permlane16_swap trades A's odd groups with B's even groups, after
        // which a register holds 32 CONSECUTIVE channels of one row in lanes 0-31 and of another row in lanes 32-63: every store
        // instruction then writes two whole 128-byte row segments, as the 32x32 tile's does, instead of four 64-byte ones (which
        // cost the strided concat-slice outputs of the C3 blocks 8-10 % of the layer).  Same arithmetic per element, so same bits.
        const int lg = (int)(threadIdx.x & 63) >> 4;
        const int hsel = lg & 1;                  // after the swap this lane stores the left (0) / right (1) half of the 32 channels
        const int rbase = mrow0 - 4 * lg + 8 * (lg >> 1);   // ... of row rbase + e (first register) and rbase + 4 + e (second)
#pragma unroll
        for (int u = 0; u < TN; u += 2) {
            const int o_l = ocol0 + u * 16, o_r = o_l + 16;   // this lane's channels in the two tiles
            const float bv_l = bias_pre ? bias_pre[u] : ((a.bias && o_l < a.ocg) ? a.bias[g * a.ocg + o_l] : 0.0f);
            const float bv_r = bias_pre ? bias_pre[u + 1] : ((a.bias && o_r < a.ocg) ? a.bias[g * a.ocg + o_r] : 0.0f);
            const int o = o_l + 16 * hsel;        // the channel this lane stores
            const bool live = o < a.ocg;
            const int oc_abs = g * a.ocg + (live ? o : 0);
            const bool second = a.out2 != nullptr && oc_abs >= a.split;
            float* const obase = second ? a.out2 + (oc_abs - a.split) : a.out + oc_abs;
            const int old = second ? a.out2_ld : a.out_ld;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int mb = rbase + t * 16;
                float* const op = obase + (size_t)mb * old;
                const float* const rp = HAS_RES ? a.res + (size_t)mb * a.res_ld + oc_abs : nullptr;
                // the block's residual values, ALL requested before its first store: a load behind a store that may alias it is
                // not moved up by the compiler, and on gfx9 vmcnt counts the stores too, so "load, wait, add, store" per element
                // is a full memory round trip per element.  (A row behind M re-reads row M - 1: always a valid address.)
                float rx[4], ry[4];
                if (HAS_RES) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        rx[e] = rp[((INTERIOR || mb + e < a.M) ? e : a.M - 1 - mb) * a.res_ld];
                        ry[e] = rp[((INTERIOR || mb + 4 + e < a.M) ? 4 + e : a.M - 1 - mb) * a.res_ld];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float vl = act_fn<ACT1>(acc[t][u][e] + bv_l, a.act_param);
                    const float vr = act_fn<ACT1>(acc[t][u + 1][e] + bv_r, a.act_param);
                    // (inline assembly: with __builtin_amdgcn_permlane16_swap hipcc 7.2 used the FIRST result for both halves in this
                    // function -- the second result's register was reused as an address before its store; the s_nop covers the
                    // VALU-write -> permlane-read hazard the compiler otherwise pads)
                    float x = vl, y = vr;
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
                    if (live && (INTERIOR || mb + e < a.M)) {
                        if (HAS_RES) x += rx[e];
                        op[e * old] = act_fn<ACT2>(x, a.act_param);
                    }
                    if (live && (INTERIOR || mb + 4 + e < a.M)) {
                        if (HAS_RES) y += ry[e];
                        op[(4 + e) * old] = act_fn<ACT2>(y, a.act_param);
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int o = ocol0 + u * MT;  // channel inside the group
        if (o >= a.ocg) continue;
        const int oc_abs = g * a.ocg + o;
        const float bv = bias_pre ? bias_pre[u] : (a.bias ? a.bias[oc_abs] : 0.0f);
        const bool second = a.out2 != nullptr && oc_abs >= a.split;
        float* const obase = second ? a.out2 + (oc_abs - a.split) : a.out + oc_abs;
        const int old = second ? a.out2_ld : a.out_ld;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int mb = mrow0 + t * MT;
            float* const op = obase + (size_t)mb * old;
            const float* const rp = HAS_RES ? a.res + (size_t)mb * a.res_ld + oc_abs : nullptr;
            if (ACT1 == SI_ACT_SILU && ACT2 == SI_ACT_NONE && !HAS_RES) {
                // SiLU on PAIRS of rows with the packed fp32 instructions (v_pk_add / v_pk_mul: two values per issue slot, each
                // component rounded exactly like the scalar form, so the bits are the scalar path's): bias add, the -log2(e)
                // scale, the + 1 and the final product take 4 vector issues per pair instead of 8; v_exp / v_rcp stay scalar.
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int e = 0; e < NE; e += 2) {
                    const int c0 = Mma<MT>::row(e);   // rows c0 and c0 + 1 (e and e + 1 never straddle a group of 4)
                    const f32x2 v = f32x2{acc[t][u][e], acc[t][u][e + 1]} + f32x2{bv, bv};
                    const f32x2 x = v * f32x2{-1.44269504088896340736f, -1.44269504088896340736f};
                    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])} + f32x2{1.0f, 1.0f};
                    const f32x2 o = v * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                    if (INTERIOR || mb + c0 < a.M) op[c0 * old] = o[0];
                    if (INTERIOR || mb + c0 + 1 < a.M) op[(c0 + 1) * old] = o[1];
                }
            } else {
            float rv[NE];   // (all of the block's residual values before its first store, see above)
            if (HAS_RES) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int c = Mma<MT>::row(e);
                    rv[e] = rp[((INTERIOR || mb + c < a.M) ? c : a.M - 1 - mb) * a.res_ld];
                }
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int c = Mma<MT>::row(e);  // row of the C/D map (plus 4 * (lane >> 5) or 4 * (lane >> 4), already in mrow0)
                if (INTERIOR || mb + c < a.M) {
                    float v = act_fn<ACT1>(acc[t][u][e] + bv, a.act_param);
                    if (HAS_RES) v += rv[e];
                    op[c * old] = act_fn<ACT2>(v, a.act_param);
                }
            }
            }
        }
    }
}

template <int MT, int TM, int TN, int ACT1, int ACT2 = SI_ACT_NONE>
__device__ __forceinline__ void epilogue_pick(const ConvArgs& a, typename Mma<MT>::acc_t (&acc)[TM][TN], int g, int mrow0, int ocol0, bool interior,
                                              const float* bias_pre) {
    if (a.res) {
        if (interior) epilogue_lean<MT, TM, TN, ACT1, ACT2, true, true>(a, acc, g, mrow0, ocol0, bias_pre);
        else epilogue_lean<MT, TM, TN, ACT1, ACT2, true, false>(a, acc, g, mrow0, ocol0, bias_pre);
    } else {
        if (interior) epilogue_lean<MT, TM, TN, ACT1, ACT2, false, true>(a, acc, g, mrow0, ocol0, bias_pre);
        else epilogue_lean<MT, TM, TN, ACT1, ACT2, false, false>(a, acc, g, mrow0, ocol0, bias_pre);
    }
}

// generic epilogue: any act1 / act2 combination (runtime switches per element)
template <int MT, int TM, int TN>
__device__ __forceinline__ void epilogue_generic(const ConvArgs& a, typename Mma<MT>::acc_t (&acc)[TM][TN], int g, int mrow0, int ocol0) {
    const bool has_bias = a.bias != nullptr;
    const bool has_res = a.res != nullptr;
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int o = ocol0 + u * MT;
        if (o >= a.ocg) continue;
        const int oc_abs = g * a.ocg + o;
        const float bv = has_bias ? a.bias[oc_abs] : 0.0f;
        // split output: the boundary is a multiple of 32, so a 32-wide column tile goes to one destination
        const bool second = a.out2 != nullptr && oc_abs >= a.split;
        float* const obase = second ? a.out2 + (oc_abs - a.split) : a.out + oc_abs;
        const int old = second ? a.out2_ld : a.out_ld;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int mb = mrow0 + t * MT;
#pragma unroll
            for (int e = 0; e < Mma<MT>::NE; ++e) {
                const int m = mb + Mma<MT>::row(e);
                if (m < a.M) {
                    float v = acc[t][u][e] + bv;
                    v = apply_act(a.act1, v, a.act_param);
                    if (has_res) v += a.res[(size_t)m * a.res_ld + oc_abs];
                    v = apply_act(a.act2, v, a.act_param);
                    obase[(size_t)m * old] = v;
                }
            }
        }
    }
}

// Detect head: sigmoid, grid / anchor decode and the concat into [N][rows_total][ne] in the conv epilogue
// (reference src/layer/yolo_detect.cpp:223-266).  Output channel o = anchor*ne + e; pixel (y,x) of level rows
// [H][W][anchor], so one pixel's na*ne outputs are contiguous: offset = ((img*rows_total + row_off + pix*na)*ne + o.
template <int MT, int TM, int TN>
__device__ __forceinline__ void epilogue_yolo(const ConvArgs& a, typename Mma<MT>::acc_t (&acc)[TM][TN], int mrow0, int ocol0) {
    const int per_pix = a.yna * a.yne;
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int o = ocol0 + u * MT;
        if (o >= a.ocg) continue;
        const float bv = a.bias ? a.bias[o] : 0.0f;
        const int anc = o / a.yne;
        const int e_ = o - anc * a.yne;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int mb = mrow0 + t * MT;
            const int img0 = mb / a.ohow;
            const int pix0 = mb - img0 * a.ohow;
#pragma unroll
            for (int e = 0; e < Mma<MT>::NE; ++e) {
                const int dm = Mma<MT>::row(e);
                if (mb + dm < a.M) {
                    int pix = pix0 + dm, img = img0;
                    if (pix >= a.ohow) {  // tile rows may span several images when the level is tiny (2x2, 4x4 maps)
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
                    a.out[((size_t)img * a.yrows_total + a.yrow_off) * a.yne + (size_t)pix * per_pix + o] = v;
                }
            }
        }
    }
}

// `interior`: every row of the workgroup's tile is below M (workgroup-uniform)
// `yolo_img` >= 0: the tile also lies inside that one image (Detect decode takes its straight-line form)
// `bias_pre`: this lane's TN bias values, loaded before the K loop (their latency is otherwise paid at the head of the
// epilogue, by every workgroup of a round at the same time)
// YMODE: -1 the Detect form is a runtime switch (generic kernel), 0 never, 1 always (the Detect instantiation of the fast kernel)
template <int TM, int TN, int YMODE = -1, int MT = 32>
__device__ __forceinline__ void epilogue(const ConvArgs& a, typename Mma<MT>::acc_t (&acc)[TM][TN], int g, int mrow0, int ocol0, bool interior = false,
                                         int yolo_img = -1, const float* bias_pre = nullptr) {
    if (YMODE == 1 || (YMODE < 0 && a.ymode)) {
        if (yolo_img >= 0) si_yolo_tile_one_image<TM, TN, ConvArgs, MT, typename Mma<MT>::acc_t>(a, a.out, acc, mrow0, ocol0, yolo_img);
        else epilogue_yolo<MT, TM, TN>(a, acc, mrow0, ocol0);
        return;
    }
    // the shapes the YOLOv5 / ResNet graphs produce get straight-line code; the rest is generic
    if (a.act2 == SI_ACT_NONE && a.act1 == SI_ACT_SILU) {
        epilogue_pick<MT, TM, TN, SI_ACT_SILU>(a, acc, g, mrow0, ocol0, interior, bias_pre);
    } else if (a.act2 == SI_ACT_NONE && a.act1 == SI_ACT_NONE) {
        epilogue_pick<MT, TM, TN, SI_ACT_NONE>(a, acc, g, mrow0, ocol0, interior, bias_pre);
    } else if (a.act2 == SI_ACT_NONE && a.act1 == SI_ACT_RELU) {
        epilogue_pick<MT, TM, TN, SI_ACT_RELU>(a, acc, g, mrow0, ocol0, interior, bias_pre);
    } else if (a.act1 == SI_ACT_NONE && a.act2 == SI_ACT_RELU) {  // ResNet: conv -> add -> ReLU
        epilogue_pick<MT, TM, TN, SI_ACT_NONE, SI_ACT_RELU>(a, acc, g, mrow0, ocol0, interior, bias_pre);
    } else {
        epilogue_generic<MT, TM, TN>(a, acc, g, mrow0, ocol0);
    }
}

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;
#ifndef SI_IGEMM_LDL16
#define SI_IGEMM_LDL16 40   // development: 0 = the old pitch (36) for the 16x16x4 tiles too
#endif

SI_STAMP_ARRAY(si_diag_stamps);   // diagnostic build only (si_hip_internal.h)

template <int BM, int BN, int WM, int WN, bool VEC_A>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = BM / WM / 32;  // 32x32 accumulator tiles per wave along M
    constexpr int TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    constexpr int A_IT = BM / 32;  // 16-byte vectors per thread per K-tile
    constexpr int B_IT = BN / 32;

    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDS_LD];

    // ---- tile mapping: the n_tiles blocks that share one A panel get the same
    // blockIdx % 8, i.e. the same XCD / L2 (placement is a speed hint only).
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
    const int kv = tid & 7;   // which 4-wide K vector of the 32-wide tile
    const int r0 = tid >> 3;  // base row 0..31

    // ---- per-thread A rows (fixed for the whole K loop)
    int a_pix[A_IT], a_ih0[A_IT], a_iw0[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < a.M) {
            const int img = m / a.ohow;
            const int rem = m - img * a.ohow;
            const int oy = rem / a.ow;
            const int ox = rem - oy * a.ow;
            a_pix[i] = img * a.ih * a.iw;
            a_ih0[i] = oy * a.sh - a.pt;
            a_iw0[i] = ox * a.sw - a.pl;
        } else {
            a_pix[i] = 0;
            a_ih0[i] = -(1 << 28);
            a_iw0[i] = 0;
        }
    }
    const float* in_g = a.in + (size_t)g * a.icg;
    const float* w_g = a.w + (size_t)g * a.ocg * a.Kp;

    f32x4 pa[A_IT], pb[B_IT];

    auto load_tile = [&](int kt) {
        const int k = kt * BK + kv * 4;
        int tap, c;
        if (a.cb_major) {  // k = (cb * ntaps + tap) * 32 + c%32
            const int blk = k >> 5;
            const int cb = blk / a.ntaps;
            tap = blk - cb * a.ntaps;
            c = cb * 32 + (k & 31);
        } else {
            tap = k / a.icg_pad;
            c = k - tap * a.icg_pad;
        }
        const int ky = tap / a.kw;
        const int kx = tap - ky * a.kw;
        const bool kvalid = k < a.Kp;
        const int dy = ky * a.dh, dx = kx * a.dw;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int y = a_ih0[i] + dy, x = a_iw0[i] + dx;
            const bool ok = kvalid && c < a.icg && (unsigned)y < (unsigned)a.ih && (unsigned)x < (unsigned)a.iw;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const float* p = in_g + (size_t)(a_pix[i] + y * a.iw + x) * a.in_ld + c;
                if (VEC_A) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    if (c + 0 < a.icg) v.x = p[0];
                    if (c + 1 < a.icg) v.y = p[1];
                    if (c + 2 < a.icg) v.z = p[2];
                    if (c + 3 < a.icg) v.w = p[3];
                }
            }
            pa[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int o = n0 + r0 + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kvalid && o < a.ocg) v = *reinterpret_cast<const f32x4*>(w_g + (size_t)o * a.Kp + k);
            pb[i] = v;
        }
    };

    auto store_tile = [&](int buf) {
        float* As = lds[buf];
        float* Bs = lds[buf] + BM * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<f32x4*>(As + (r0 + 32 * i) * LDS_LD + kv * 4) = pa[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<f32x4*>(Bs + (r0 + 32 * i) * LDS_LD + kv * 4) = pb[i];
    };

    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.0f;

    const int nk = (a.Kp + BK - 1) / BK;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);

        const float* As = lds[cur] + (wm * TM * 32 + l31) * LDS_LD + lh * 4;
        const float* Bs = lds[cur] + BM * LDS_LD + (wn * TN * 32 + l31) * LDS_LD + lh * 4;
#pragma unroll
        for (int p = 0; p < BK / 16; ++p) {   // the canonical k order (see the fast kernel's K-tile loop)
            f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[h][t] = *reinterpret_cast<const f32x4*>(As + t * 32 * LDS_LD + p * 16 + h * 8);
#pragma unroll
                for (int u = 0; u < TN; ++u) fb[h][u] = *reinterpret_cast<const f32x4*>(Bs + u * 32 * LDS_LD + p * 16 + h * 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < TM; ++t)
#pragma unroll
                        for (int u = 0; u < TN; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][t][j], fb[h][u][j], acc[t][u], 0, 0, 0);
        }

        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    epilogue<TM, TN>(a, acc, g, m0 + wm * TM * 32 + 4 * lh, n0 + wn * TN * 32 + l31);
}

// ---- fast path: ic/groups a multiple of 32, 16-byte aligned tensors smaller than 4 GB -------------------
// * K runs channel-block major, (c/32, kh, kw, c%32): all taps of one 32-channel block are consumed
//   back to back, so the 3x3 halo re-reads of a pixel's 128-byte line come 1..3 K-tiles apart and hit L2
//   instead of 4..12 tiles apart (they were thrashing the 4 MiB XCD L2: 40-60 % hit rate measured).
// * the tap walk (cb, ky, kx) is wave-uniform scalar code; per row the thread only adds a scalar delta to a
//   fixed byte offset and looks its tap up in a 64-bit validity mask computed once.
// * loads are raw buffer loads: an out-of-image tap (or a row past M / past OC) is an out-of-range offset
//   and the hardware returns zeros -- no branches around the loads.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// n / d for 0 <= n < 2^32 with mg = floor(2^32 / d) (0xFFFFFFFF for d = 1): mulhi gives the quotient or one less, one
// correction step makes it exact -- 5 instructions instead of the ~40 of an emulated integer division, twice per staged row
__device__ __forceinline__ int fast_div(int n, int d, unsigned mg) {
    unsigned q = __umulhi((unsigned)n, mg);
    if ((unsigned)n - q * (unsigned)d >= (unsigned)d) ++q;
    return (int)q;
}
constexpr unsigned OOB_A = 0xFFFFFF00u;  // >= any legal num_records
constexpr unsigned OOB_B = 0x80000000u;  // weights are < 2 GB; + kt*128 cannot wrap

// PADK: the K axis of a 1x1 conv is zero-padded to whole 32-channel blocks (conv_icg_pad); a separate instantiation so the
// channel check costs the ordinary layers nothing (it was worth 0.5 % of the YOLOv5s step inside the shared loop)
// UPS: the dual-source pointwise form (ConvArgs::up); its own instantiation, so the ordinary layers carry neither the extra row
// offsets nor the per-K-tile source select
// YOLO: the Detect form (decode + concat in the epilogue); its own instantiation, so a profile lists Detect's three launches --
// which run on the engine's second stream beside other layers -- apart from the ordinary ones
// PW: plain pointwise layers (1x1, stride 1, no padding, whole 32-channel K blocks): a row's offset is fixed and never before
// the tensor, so the K-tile's channel offset rides in the load's SCALAR offset and a K-tile costs no vector instruction for
// addressing (the general form spends ~12 per K-tile on tap masks and offset adds -- on the vector issue port the short-K layers
// are bound by)
// MT: the MFMA tile, 32 (v_mfma_f32_32x32x2_f32) or 16 (v_mfma_f32_16x16x4_f32: workgroup tiles of 32 rows for launches that
// would otherwise leave most of the chip without a wave, i.e. small batches; same bits, see Mma / the K-tile loop)
// KU: K-tiles per barrier round.  2 = two K-tiles are staged side by side and consumed back to back between ONE pair of barriers
// (NBUF = 1 only): the same MFMA sequence per output element, half the barrier / LDS-write / LDS-read round trips per K.  For
// launches that leave a CU with one or two workgroups (small batches), where that round trip, not the matrix pipe, sets the pace.
template <int BM, int BN, int WM, int WN, int NBUF, bool PADK = false, bool UPS = false, bool YOLO = false, bool PW = false, int MT = 32, int KU = 1>
__global__ __launch_bounds__(256) void conv_igemm_f32_fast_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(NBUF == 1 || NBUF == 2, "one or two LDS stages");
    static_assert(KU == 1 || ((KU == 2 || KU == 4) && NBUF == 1), "several K-tiles per barrier round: one-stage form only");
    static_assert(MT == 32 || MT == 16, "MFMA tile");
    static_assert(BM % 32 == 0 && BN % 32 == 0 && BM % (WM * MT) == 0 && BN % (WN * MT) == 0, "tile shape");
    constexpr int TM = BM / WM / MT;
    constexpr int TN = BN / WN / MT;
    constexpr int A_IT = BM / 32;
    constexpr int B_IT = BN / 32;

    // NBUF = 2: one barrier per K-tile.  NBUF = 1: half the LDS (more workgroups per CU), two barriers
    // per K-tile; the other resident workgroups cover them.
    // LDS row pitch: 36 floats put the 16 rows of a ds_read_b128 lane group of the 32x32x2 operand layout on 16 distinct 4-bank
    // groups; the 16x16x4 layout (row = lane & 15, k group = lane >> 4) mixes two k groups in one lane group and needs 40 for the
    // same (with 36 every such read took two LDS cycles per lane group: SQ_LDS_BANK_CONFLICT = a third of SQ_LDS_IDX_ACTIVE,
    // profiles/r03_pmc_lds.txt)
    constexpr int LDL = SI_IGEMM_LDL16 > 0 && MT == 16 ? SI_IGEMM_LDL16 : LDS_LD;
    __shared__ __attribute__((aligned(16))) float lds[NBUF * KU][(BM + BN) * LDL];
    SI_STAMP_DECL;
    SI_STAMP_RT(0);
    SI_STAMP(1);

    const int g = blockIdx.y;
    const int per_chunk = 8 * a.n_tiles;
    const int chunk = fast_div((int)blockIdx.x, per_chunk, a.mg_chunk);
    const int r = blockIdx.x - chunk * per_chunk;
    const int m_tile = chunk * 8 + (r & 7);
    const int n_tile = r >> 3;
    if (m_tile >= a.m_tiles) return;
    const int m0 = m_tile * BM;
    const int n0 = n_tile * BN;

    const int tid = threadIdx.x;
    const int kv = tid & 7;
    const int r0 = tid >> 3;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.in + (size_t)g * a.icg), 0, a.in_bytes - (unsigned)g * a.icg * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w + (size_t)g * a.ocg * a.Kp), 0, (unsigned)a.ocg * a.Kp * 4u, 0x00020000);

    const __amdgpu_buffer_rsrc_t rs_up = UPS ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.up), 0, a.up_bytes, 0x00020000) : rs_in;

    // per-thread A rows: byte offset of (tap 0, channel kv*4) and the tap validity mask
    unsigned a_off[A_IT];
    unsigned u_off[UPS ? A_IT : 1];   // UPS: byte offset of the row's source pixel in the low-resolution tensor
    unsigned long long a_mask[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_off[i] = 0;
        a_mask[i] = 0ull;
        if (UPS) {
            u_off[i] = OOB_A;
            if (m < a.M) {
                const int img = fast_div(m, a.ohow, a.mg_ohow);
                const int rem = m - img * a.ohow;
                const int oy = fast_div(rem, a.ow, a.mg_ow);
                const int ox = rem - oy * a.ow;
                // upsample.cpp:85-92: src = clamp(int(float(dst) * (1 / scale)), 0, in - 1)
                int sy = (int)((float)oy * a.up_inv_h), sx = (int)((float)ox * a.up_inv_w);
                sy = max(0, min(a.up_ih - 1, sy));
                sx = max(0, min(a.up_iw - 1, sx));
                u_off[i] = (unsigned)((img * a.up_ih + sy) * a.up_iw + sx) * (unsigned)(a.up_ld * 4) + (unsigned)(kv * 16);
            }
        }
        if (PW || UPS || YOLO) {   // (the dual-source and Detect forms are pointwise by construction)
            a_off[i] = m < a.M ? (unsigned)m * (unsigned)(a.in_ld * 4) + (unsigned)(kv * 16) : OOB_A;
        } else if (m < a.M && a.pointwise) {
            // 1x1, stride 1, no padding: output pixel m IS input pixel m -- no index decomposition, one always-valid tap
            a_off[i] = (unsigned)m * (unsigned)(a.in_ld * 4) + (unsigned)(kv * 16);
            a_mask[i] = 1ull;
        } else if (m < a.M) {
            const int img = fast_div(m, a.ohow, a.mg_ohow);
            const int rem = m - img * a.ohow;
            const int oy = fast_div(rem, a.ow, a.mg_ow);
            const int ox = rem - oy * a.ow;
            const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
            // modulo-2^32 arithmetic: (tap-0 pixel may lie before the tensor start; adding the tap delta brings
            // every VALID tap back into [0, in_bytes))
            a_off[i] = (unsigned)((img * a.ih + y0) * a.iw + x0) * (unsigned)(a.in_ld * 4) + (unsigned)(kv * 16);
            unsigned long long mk = 0ull;
            if (a.kh == 3 && a.kw == 3) {
                // the 3x3 case straight-line: three row bits, three column bits (the generic loop below is ~15 instructions per
                // tap with 64-bit shifts; tools/conv_diag.py: prologue 55 k -> 30 k cycles for a workgroup that starts beside
                // seven others in their K loops -- a shorter launch ramp, no measurable change in steady state)
                unsigned rowbits = 0, colbits = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    rowbits |= ((unsigned)(y0 + t * a.dh) < (unsigned)a.ih ? 1u : 0u) << t;
                    colbits |= ((unsigned)(x0 + t * a.dw) < (unsigned)a.iw ? 1u : 0u) << t;
                }
                unsigned m9 = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) m9 |= ((rowbits >> t) & 1u) ? (colbits << (3 * t)) : 0u;
                mk = m9;
            } else
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
        const int o = n0 + r0 + 32 * i;
        b_off[i] = o < a.ocg ? (unsigned)(o * a.Kp * 4 + kv * 16) : OOB_B;
    }

    u32x4 pa_[KU][A_IT], pb_[KU][B_IT];
    // wave-uniform K walk
    int cb = 0, ky = 0, kx = 0;

    auto load_tile = [&](u32x4 (&pa)[A_IT], u32x4 (&pb)[B_IT], int kt) {
        if (PW || UPS || YOLO) {
            const unsigned kb = (unsigned)kt * (BK * 4);   // the K-tile's 32 channels: the same 128 bytes further in both operands
            // UPS: K-tiles [up_cb0, up_cb1) come from the low-resolution tensor at the row's source pixel (wave-uniform choice)
            const bool from_up = UPS && kt >= a.up_cb0 && kt < a.up_cb1;
            if (from_up) {
                const unsigned du = (unsigned)(kt - a.up_cb0) * 128u;
#pragma unroll
                for (int i = 0; i < A_IT; ++i) pa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_up, u_off[i], du, 0);
            } else {
#pragma unroll
                for (int i = 0; i < A_IT; ++i) pa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_off[i], kb, 0);
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) pb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[i], kb, 0);
            return;
        }
        const unsigned delta = (unsigned)((ky * a.dh * a.iw + kx * a.dw) * a.in_ld + cb * 32) * 4u;
        const int tapbit = ky * a.kw + kx;
        const bool cok = !PADK || cb * 32 + kv * 4 < a.icg;  // false only in the zero-padded tail block of a 1x1 conv
        const bool from_up = UPS && cb >= a.up_cb0 && cb < a.up_cb1;   // wave-uniform: this K-tile's channels are upsampled ones
        if (from_up) {
            const unsigned du = (unsigned)(cb - a.up_cb0) * 128u;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) pa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_up, u_off[i] == OOB_A ? OOB_A : u_off[i] + du, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const bool ok = ((a_mask[i] >> tapbit) & 1ull) && cok;
                const unsigned off = ok ? a_off[i] + delta : OOB_A;
                pa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
            }
        }
        const unsigned kb = (unsigned)kt * (BK * 4);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) pb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[i], kb, 0);
        if (++kx == a.kw) {
            kx = 0;
            if (++ky == a.kh) {
                ky = 0;
                ++cb;
            }
        }
    };

    auto store_tile = [&](const u32x4 (&pa)[A_IT], const u32x4 (&pb)[B_IT], int buf) {
        float* As = lds[buf];
        float* Bs = lds[buf] + BM * LDL;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<u32x4*>(As + (r0 + 32 * i) * LDL + kv * 4) = pa[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<u32x4*>(Bs + (r0 + 32 * i) * LDL + kv * 4) = pb[i];
    };

    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lrow = lane & (MT - 1), lk = lane / MT;   // operand row and k slot of this lane (Mma<MT>)

    typename Mma<MT>::acc_t acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int e = 0; e < Mma<MT>::NE; ++e) acc[t][u][e] = 0.0f;

#if defined(SI_EXP_NOKLOOP)
    // experiment build only (tools/r06_small_batch_floor.sh): the K loop compiled out -- what is left is the launch,
    // the prologue (index arithmetic, the first K-tile's loads, its LDS round trip, one barrier) and the epilogue
    const int nk = 0;
#else
    const int nk = a.Kp / BK;  // icg_pad % 32 == 0
#endif
#if defined(SI_EXP_EMPTY)
    if (a.M >= 0) return;      // experiment build only: the launch alone
#endif

    load_tile(pa_[0], pb_[0], 0);
#pragma unroll
    for (int u = 1; u < KU; ++u)
        if (u < nk) load_tile(pa_[u], pb_[u], u);
    // this lane's bias values ride along with the first tile's loads
    float bias_pre[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int o = n0 + wn * TN * MT + lrow + u * MT;
        bias_pre[u] = (a.bias && o < a.ocg) ? a.bias[g * a.ocg + o] : 0.0f;
    }
    SI_STAMP(2);
    store_tile(pa_[0], pb_[0], 0);
#pragma unroll
    for (int u = 1; u < KU; ++u)
        if (u < nk) store_tile(pa_[u], pb_[u], u);
    __syncthreads();
    SI_STAMP(3);

    // one K-tile from LDS stage `buf`
    auto compute_tile = [&](int buf) {
        const float* As = lds[buf] + (wm * TM * MT + lrow) * LDL + lk * 4;
        const float* Bs = lds[buf] + BM * LDL + (wn * TN * MT + lrow) * LDL + lk * 4;
        // CANONICAL K ORDER (every fp32 implicit-GEMM kernel of this file, so that they agree bit for bit): inside each 16-wide
        // block of a K-tile an output element accumulates k = j, 4+j, 8+j, 12+j for j = 0..3.  On the 16x16x4 MFMA that is one
        // 16-byte read per operand row (lane group g = lane >> 4 reads k = 4g..4g+3; MFMA j takes register j and chains the four
        // groups in order); on the 32x32x2 MFMA it is two reads (k = 4h.. and 8+4h.., h = lane >> 5) whose MFMAs alternate.
#pragma unroll
        for (int p = 0; p < BK / 16; ++p) {
            if constexpr (MT == 32) {
                f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int t = 0; t < TM; ++t) fa[h][t] = *reinterpret_cast<const f32x4*>(As + t * 32 * LDL + p * 16 + h * 8);
#pragma unroll
                    for (int u = 0; u < TN; ++u) fb[h][u] = *reinterpret_cast<const f32x4*>(Bs + u * 32 * LDL + p * 16 + h * 8);
                }
                // raised priority around the MFMA cluster: +0.5-1.2 % on YOLOv5s (same-box A/B; per cdna_hip_programming.md T5 the
                // effect is on how hipcc places the cluster relative to the LDS reads and barriers, not the s_setprio itself)
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < TM; ++t)
#pragma unroll
                            for (int u = 0; u < TN; ++u)
                                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][t][j], fb[h][u][j], acc[t][u], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            } else {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[t] = *reinterpret_cast<const f32x4*>(As + t * 16 * LDL + p * 16);
#pragma unroll
                for (int u = 0; u < TN; ++u) fb[u] = *reinterpret_cast<const f32x4*>(Bs + u * 16 * LDL + p * 16);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < TM; ++t)
#pragma unroll
                        for (int u = 0; u < TN; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t][j], fb[u][j], acc[t][u], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
        }
    };

    if constexpr (KU > 1) {
        for (int kt = 0; kt < nk; kt += KU) {
#pragma unroll
            for (int u = 0; u < KU; ++u)
                if (kt + KU + u < nk) load_tile(pa_[u], pb_[u], kt + KU + u);
#pragma unroll
            for (int u = 0; u < KU; ++u)
                if (kt + u < nk) compute_tile(u);
            if (kt + KU < nk) {
                __syncthreads();  // everyone is done reading tiles kt .. kt+KU-1
#pragma unroll
                for (int u = 0; u < KU; ++u)
                    if (kt + KU + u < nk) store_tile(pa_[u], pb_[u], u);
                __syncthreads();
            }
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = NBUF == 2 ? (kt & 1) : 0;
            if (kt + 1 < nk) load_tile(pa_[0], pb_[0], kt + 1);
            compute_tile(cur);
            if (NBUF == 2) {
                if (kt + 1 < nk) store_tile(pa_[0], pb_[0], cur ^ 1);
                __syncthreads();
            } else {
                __syncthreads();  // everyone is done reading tile kt
                if (kt + 1 < nk) {
                    store_tile(pa_[0], pb_[0], 0);
                    __syncthreads();
                }
            }
        }
    }

    SI_STAMP(4);

    int yolo_img = -1;
    if (YOLO && m0 + BM <= a.M) {
        const int img = m0 / a.ohow;
        if (m0 - img * a.ohow + BM <= a.ohow) yolo_img = img;
    }
    epilogue<TM, TN, YOLO ? 1 : 0, MT>(a, acc, g, m0 + wm * TM * MT + 4 * lk, n0 + wn * TN * MT + lrow, m0 + BM <= a.M, yolo_img, bias_pre);
    SI_STAMP(5);
    SI_STAMP_RT(6);
    SI_STAMP_FLUSH(si_diag_stamps);
}

#ifdef SI_DIAG_STAMPS
}  // namespace
SI_STAMP_ACCESSORS(si_diag_stamps, si_hip_diag_stamps_read, si_hip_diag_stamps_clear)
namespace {
#endif

template <int BM, int BN, int WM, int WN, int NBUF, int MT = 32, int KU = 1>
int launch_fast(const ConvArgs& a, int groups, hipStream_t s) {
    ConvArgs b = a;
    b.m_tiles = (a.M + BM - 1) / BM;
    b.n_tiles = (a.ocg + BN - 1) / BN;
    b.mg_chunk = (unsigned)(0x100000000ull / (unsigned)(8 * b.n_tiles));
    const int chunks = (b.m_tiles + 7) / 8;
    dim3 grid(chunks * 8 * b.n_tiles, groups, 1);
    // the tiles that carry every instantiation (pointwise / dual-source / zero-padded K): the two defaults and the 16x16-MFMA ones
    constexpr bool kFull = (MT == 32 && NBUF == 1 && ((BM == 64 && BN == 64) || (BM == 128 && BN == 32))) || MT == 16;
    if (a.ymode) {
        // Detect: the default tile and the 16x16-MFMA ones
        if constexpr ((MT == 32 && NBUF == 1 && BM == 64 && BN == 64) || MT == 16) {
            if (a.icg % 32 != 0 || a.up) return SI_E_UNSUPPORTED;
            hipLaunchKernelGGL((conv_igemm_f32_fast_kernel<BM, BN, WM, WN, NBUF, false, false, true, false, MT, KU>), grid, dim3(256), 0, s, b);
        } else {
            return SI_E_UNSUPPORTED;
        }
    } else if (a.up) {
        // dual-source pointwise conv (consumer of cat(upsample(x), skip))
        if constexpr (kFull) {
            if (a.icg % 32 != 0 || !a.pointwise) return SI_E_UNSUPPORTED;
            hipLaunchKernelGGL((conv_igemm_f32_fast_kernel<BM, BN, WM, WN, NBUF, false, true, false, false, MT, KU>), grid, dim3(256), 0, s, b);
        } else {
            return SI_E_UNSUPPORTED;
        }
    } else if (a.icg % 32 != 0) {
        // zero-padded K (1x1 convs with a channel count that is not a multiple of 32)
        if constexpr (kFull)
            hipLaunchKernelGGL((conv_igemm_f32_fast_kernel<BM, BN, WM, WN, NBUF, true, false, false, false, MT, KU>), grid, dim3(256), 0, s, b);
        else
            return SI_E_UNSUPPORTED;
    } else {
        if constexpr (kFull) {
            if (a.pointwise) {
                hipLaunchKernelGGL((conv_igemm_f32_fast_kernel<BM, BN, WM, WN, NBUF, false, false, false, true, MT, KU>), grid, dim3(256), 0, s, b);
                return (int)hipGetLastError();
            }
        }
        hipLaunchKernelGGL((conv_igemm_f32_fast_kernel<BM, BN, WM, WN, NBUF, false, false, false, false, MT, KU>), grid, dim3(256), 0, s, b);
    }
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN>
int launch(const ConvArgs& a, int groups, bool vec_a, hipStream_t s) {
    ConvArgs b = a;
    b.m_tiles = (a.M + BM - 1) / BM;
    b.n_tiles = (a.ocg + BN - 1) / BN;
    const int chunks = (b.m_tiles + 7) / 8;
    dim3 grid(chunks * 8 * b.n_tiles, groups, 1);
    if (vec_a)
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, true>), grid, dim3(256), 0, s, b);
    else
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false>), grid, dim3(256), 0, s, b);
    return (int)hipGetLastError();
}

inline int round_up4(int v) { return (v + 3) & ~3; }

}  // namespace

// Tile choice.  Measured on MI355X over every YOLOv5s / ResNet18 conv shape at batch 32
// (tools/conv_bench.py, SI_CONV_VARIANT sweep, profiles/r01_conv_variants.txt): the fp32 MFMA is slow
// enough (64 cycles per 32x32x2) that operand reuse is not the limit -- residency is.  A 64x64 tile with
// ONE LDS stage (18 KB) lets 7-8 workgroups share a CU, which hides HBM/L2 latency and evens out the tail
// when a layer only has a few hundred tiles; it beats the 128x128 double-buffered tile on every shape
// (6.7 ms vs 10.6 ms summed over the net).  Layers with <= 32 output channels use 128x32.
//   id: 0 128x128x2  1 128x64x2  2 64x64x2  3 128x32x2  4 64x64x1  5 64x128x1  6 128x64x1  7 128x128x1
//       8 64x128x2  10 128x32x1        (BM x BN x LDS stages; 32x32x2 MFMA)
//       11 32x64x2  12 32x32x2  13 32x64x1  14 32x32x1  16 64x32x1  17 64x64x1  19 64x32x1 as 4x1 waves     (16x16x4 MFMA, round 3;
//       15 / 18 / 20 / 21 -- 64x32x2, 32x128, 128x64, 64x128 -- were measured and removed)
//       22 32x32x1 with TWO K-tiles per barrier round (KU = 2; the same on 32x64 / 64x32 / 64x64 was measured and lost, four
//          K-tiles per round gain another 0.4 % at batch 1: profiles/r03_ku2_sweep.txt)
static constexpr int kConvVariants = 23;
static int si_cu_count() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}
// ids that were measured and removed again (15: 64x32 two-stage, 18: 32x128, 20: 128x64, 21: 64x128 on the 16x16x4 MFMA --
// profiles/r03_tile_sweep.txt) and the unused 9 are not accepted
static bool conv_variant_valid(int v) { return v >= 0 && v < kConvVariants && v != 9 && v != 15 && v != 18 && v != 20 && v != 21; }
// A forced tile comes with the call (SiConv2dDesc::plan, include/si_hip.h SiConvPlan::f32_tile: tests that hold every tile to the same bits,
// sweeps); -1 / no plan: the policy below.  (Until round 6 this was a process-global setter + an environment switch.)
static int conv_forced_variant(const SiConv2dDesc* d) {
    int v = (d && d->plan) ? d->plan->f32_tile : -1;
#ifdef SI_EXPERIMENT   // variant library only (simpleinfer_amd/build.py build_hip(defines=("SI_EXPERIMENT",), ...)): sweeps without a plan
    if (v < 0) {
        static const int env = [] { const char* e = getenv("SI_CONV_VARIANT"); return e ? atoi(e) : -1; }();
        v = env;
    }
#endif
    return conv_variant_valid(v) ? v : -1;
}
// The policy (round 3; tools/tile_sweep.py on MI355X, sustained, every YOLOv5s implicit-GEMM shape at batch 4 / 8 / 16 / 32:
// profiles/r03_tile_sweep.txt).  What decides is how many workgroups a launch has against the 256 CUs, counted in 64x64 tiles:
//   * >= 4800 tiles: the 64x64 tile, on the 16x16x4 MFMA (four accumulator chains per wave instead of one: the large 3x3
//     stride-2 layers run 3-6 % faster than on the 32x32x2 MFMA, the rest the same);
//   * 1200 .. 4800: half tiles -- 64x32 (as four 16x32 waves) for pointwise layers, 32x64 for the others (2-6 % over 64x64);
//   * below: 32x32 (a 20x20 layer at batch 32 has 800 64x64-tiles, at batch 4 a hundred: -14 % ... -40 %), except 3x3 layers
//     with 600+ tiles, which still prefer 32x64; with two K-tiles per barrier round for 3x3 layers under 450 tiles and pointwise
//     layers under 64;
//   * <= 32 output channels per group: 64x32 (four 16x32 waves: 32 consecutive channels per store, see epilogue_lean).
// In the network (same-box interleaved A/B of whole policies, tools/ab_policy.sh, profiles/r03_ab_policy.txt): YOLOv5s batch 32
// 7232-7308 img/s with the round-2 rule (64x64 / 128x32 on the 32x32x2 MFMA everywhere) -> 7513-7517 with this one.
// Whatever is chosen, the bits are the same (Mma, tests/test_gpu_tiles.py).
static int conv_variant(const SiConv2dDesc* d) {
    const int forced = conv_forced_variant(d);
    if (forced >= 0) return forced;
    static const std::array<int, 8> cls = [] {
        std::array<int, 8> c = {19, 17, 17, 13, 19, 13, 14, 22};
#ifdef SI_EXPERIMENT   // variant library only: SI_CONV_POLICY="oc32,bigG,bigP,midG,midP,smallG,small[,tiny]" overrides the classes' variants (G: general, P: pointwise)
        if (const char* e = getenv("SI_CONV_POLICY")) {
            int v[8] = {0, 0, 0, 0, 0, 0, 0, 22};
            if (sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]) >= 7)
                for (int i = 0; i < 8; ++i)
                    if (conv_variant_valid(v[i])) c[(size_t)i] = v[i];
        }
#endif
        return c;
    }();
    const int ocg = d->oc / d->groups;
    if (ocg <= 32) return cls[0];
    const long long M = (long long)d->n * d->oh * d->ow;
    const long long tiles64 = ((M + 63) / 64) * ((ocg + 63) / 64) * d->groups;
    const bool pointwise = d->kh == 1 && d->kw == 1;
    // (thresholds in units of the 256-CU part they were measured on)
    const long long cus = si_cu_count();
    if (tiles64 * 256 >= 4800 * cus) return pointwise ? cls[2] : cls[1];
    if (tiles64 * 256 >= 1200 * cus) return pointwise ? cls[4] : cls[3];
    if (!pointwise && tiles64 * 256 >= 600 * cus) return cls[5];
    // launches that leave most CUs with one workgroup or none (batch 1-4; the long-K 3x3 layers up to batch 8): the barrier /
    // LDS round trip of a K-tile, not the matrix pipe, sets the pace -- two K-tiles per round trip (profiles/r03_ku2_sweep.txt:
    // 3x3 s2 40x40x256->512 at batch 1 36.1 -> 24.8 us, the batch-1 sum over the implicit-GEMM shapes 336 -> 314 us)
    if (pointwise ? tiles64 * 256 < 64 * cus : tiles64 * 256 < 450 * cus) return cls[7];
    return cls[6];
}
// the generic kernel (any channel count) has four tiles: 0 128x128, 1 128x64, 2 64x64, 3 128x32; a forced variant maps to the
// nearest one, the policy is the round-1 rule (these layers are latency / HBM bound)
static int conv_generic_tile(const SiConv2dDesc* d) {
    static const int generic_of[kConvVariants] = {0, 1, 2, 3, 2, 2, 1, 0, 2, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 2, 2};
    const int forced = conv_forced_variant(d);
    if (forced >= 0) return generic_of[forced];
    return (d->oc / d->groups) <= 32 ? 3 : 2;
}
// the variants that have the pointwise / dual-source / zero-padded-K instantiations (launch_fast: kFull)
static bool conv_variant_full(int v) { return v == 4 || v == 10 || (v >= 11 && v < kConvVariants); }

// General grouped convolution with few channels per group (ForwardIm2ColWithGroup, reference src/layer/conv_2d.cpp:285-380: one
// Eigen expression per group): 4, 8 or 16 input channels per group.  G = 32 / (ic / groups) neighbouring groups are contiguous
// 32-channel blocks of the input and contiguous blocks of the output, so they run as ONE dense group on the fast MFMA kernel
// with a block-diagonal weight image (zeros where an output channel does not see an input channel of its super-group): G x the
// necessary multiplies, but on full-width 32x32x2 MFMAs with 16-byte loads -- these layers are memory-bound either way, and the
// alternative was the generic kernel's per-element bounds checks.  Shape-only (it fixes the weight layout).
static int conv_group_merge(const SiConv2dDesc* d) {
#ifdef SI_EXPERIMENT   // variant library only: A/B switch
    static const bool enabled = [] { const char* e = getenv("SI_GROUP_MERGE"); return !(e && e[0] == '0'); }();
    if (!enabled) return 1;
#endif
    if (d->groups <= 1 || d->ic % d->groups != 0 || d->oc % d->groups != 0) return 1;
    const int icg = d->ic / d->groups;
    if (icg != 4 && icg != 8 && icg != 16) return 1;
    const int G = 32 / icg;
    return d->groups % G == 0 ? G : 1;
}
// the descriptor the kernels see: the same tensors with groups / G dense super-groups of 32 input channels
static SiConv2dDesc conv_effective(const SiConv2dDesc* d) {
    SiConv2dDesc e = *d;
    e.groups = d->groups / conv_group_merge(d);
    return e;
}

// K order is a property of the weight buffer, so it depends on the layer shape only
static bool conv_cb_major(const SiConv2dDesc* d) {
    const int icg = d->ic / d->groups;
    return (icg % 32 == 0) && (d->kh * d->kw > 1);
}

// Channels per tap in the packed weights (shape-only, it fixes the weight layout).  A 1x1 ungrouped conv whose channel
// count is a multiple of 4 but not of 32 (MobileNet's 16 / 24 / 40 / 72 / 96 ... pointwise and squeeze-excite convs) pads
// its K axis with zero weights to whole 32-channel blocks, so the fast kernel serves it: the thread whose 4-channel vector
// lies behind the last channel gets an out-of-range buffer offset (zeros) instead of the neighbouring pixel.
static int conv_icg_pad(const SiConv2dDesc* d) {
    const int icg = d->ic / d->groups;
    if (d->kh * d->kw == 1 && d->groups == 1 && icg % 4 == 0 && icg % 32 != 0 && icg >= 8) return (icg + 31) / 32 * 32;
    return round_up4(icg);
}

static bool conv_fast_ok(const SiConv2dDesc* d, const float* in) {
    const int icg = d->ic / d->groups;
    if (conv_icg_pad(d) % 32 != 0 || d->in_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(in) & 15) != 0) return false;
    if (d->kh * d->kw > 64) return false;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->ih * d->iw * d->in_ld * 4ull;
    const unsigned long long w_bytes = (unsigned long long)d->oc * d->kh * d->kw * conv_icg_pad(d) * 4ull;
    return in_bytes < 0xFFFFFF00ull && w_bytes < 0x40000000ull;
}

static bool conv_vec_a(const SiConv2dDesc* d, const float* in) {
    const int icg = d->ic / d->groups;
    return (icg % 4 == 0) && (d->in_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
}

extern "C" size_t si_hip_conv2d_weight_elems(const SiConv2dDesc* d) {
    if (!d || d->groups <= 0) return 0;
    if (si_conv_smallc_ok(d)) return si_conv_smallc_weight_elems(d);  // stem layout, see conv_smallc.hip
    if (si_conv_depthwise_ok(d)) return si_conv_depthwise_weight_elems(d);
    const SiConv2dDesc e = conv_effective(d);
    return (size_t)e.oc * e.kh * e.kw * conv_icg_pad(&e);
}

extern "C" int si_hip_conv2d_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, float* w_packed) {
    if (!d || !w_oihw || !w_packed || d->groups <= 0) return SI_E_BADARG;
    if (si_conv_smallc_ok(d)) {
        si_conv_smallc_pack(d, w_oihw, w_packed);
        return 0;
    }
    if (si_conv_depthwise_ok(d)) {
        si_conv_depthwise_pack(d, w_oihw, w_packed);
        return 0;
    }
    const int ntaps = d->kh * d->kw;
    if (const int G = conv_group_merge(d); G > 1) {
        // block-diagonal image of G merged groups: output channel o (group g, position g % G inside its super-group) sees the
        // icg channels [ (g % G) * icg, +icg ) of the super-group's 32, zeros elsewhere; K order of the dense 32-channel case
        const SiConv2dDesc e = conv_effective(d);
        const int icg = d->ic / d->groups, ocg = d->oc / d->groups;
        (void)e;   // one 32-channel block per tap: channel-block-major and tap-major K orders coincide (k = tap * 32 + c)
        for (int o = 0; o < d->oc; ++o) {
            const int gi = (o / ocg) % G;
            for (int tap = 0; tap < ntaps; ++tap)
                for (int c = 0; c < 32; ++c) {
                    const int cl = c - gi * icg;
                    const float v = (cl >= 0 && cl < icg) ? w_oihw[((size_t)o * icg + cl) * ntaps + tap] : 0.0f;
                    w_packed[((size_t)o * ntaps + tap) * 32 + c] = v;
                }
        }
        return 0;
    }
    const int icg = d->ic / d->groups, icp = conv_icg_pad(d);
    const bool cbm = conv_cb_major(d);
    for (int o = 0; o < d->oc; ++o)
        for (int y = 0; y < d->kh; ++y)
            for (int x = 0; x < d->kw; ++x) {
                const int tap = y * d->kw + x;
                for (int c = 0; c < icp; ++c) {
                    const float v = c < icg ? w_oihw[(((size_t)o * icg + c) * d->kh + y) * d->kw + x] : 0.0f;
                    const size_t k = cbm ? ((size_t)(c >> 5) * ntaps + tap) * 32 + (c & 31) : (size_t)tap * icp + c;
                    w_packed[(size_t)o * ntaps * icp + k] = v;
                }
            }
    return 0;
}

// everything si_hip_conv2d_upcat_f32 requires of a problem EXCEPT the pointers' alignment (shape-only, so a scheduler can ask
// before any buffer exists): a plain pointwise conv on the fast path, 32-channel granularity, both tensors below 4 GiB
static bool conv_upcat_shape_ok(const SiConv2dDesc* d, const SiConv2dUpsampledSource* up) {
    if (!d || !up || d->groups != 1 || d->ic <= 0 || d->oc <= 0) return false;
    const bool pointwise = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->pl == 0 && d->ih == d->oh && d->iw == d->ow;
    if (!pointwise || d->has_residual) return false;
    if (up->c <= 0 || up->c % 32 != 0 || up->c0 % 32 != 0 || up->c0 + up->c > d->ic || up->ld % 4 != 0 || up->ih <= 0 || up->iw <= 0) return false;
    if (!conv_fast_ok(d, nullptr)) return false;   // (a null pointer is 16-byte aligned: sizes, strides and channel counts only)
    const unsigned long long ub = (unsigned long long)d->n * up->ih * up->iw * up->ld * 4ull;
    return ub < 0xFFFFFF00ull;
}

extern "C" int si_hip_conv2d_upcat_supported(const SiConv2dDesc* d, const SiConv2dUpsampledSource* up) {
    return conv_upcat_shape_ok(d, up) ? 1 : 0;
}

struct SplitOut {
    float* out2;
    int out2_ld, split;
};

static int conv2d_dispatch(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                           const float* residual, float* out, si_stream_t stream, const SiYoloLevel* yolo,
                           const float* ygrid, const float* yanchor, const SplitOut* split = nullptr,
                           const SiConv2dUpsampledSource* up = nullptr) {
    if (!d || !in || !w_packed || !out) return SI_E_BADARG;
    if (d->groups <= 0 || d->ic % d->groups != 0 || d->oc % d->groups != 0) return SI_E_BADARG;
    if (d->plan && d->plan->f32_tile >= 0 && !conv_variant_valid(d->plan->f32_tile)) return SI_E_BADARG;   // an unknown or retired tile id
    SiConv2dDesc eff;
    if (!si_conv_smallc_ok(d) && !si_conv_depthwise_ok(d) && conv_group_merge(d) > 1) {
        eff = conv_effective(d);   // merged groups: from here on a dense-per-super-group conv (the weights were packed for it)
        d = &eff;
    }
    if (d->n <= 0 || d->oh <= 0 || d->ow <= 0) return SI_E_BADARG;
    if (d->has_bias && !bias) return SI_E_BADARG;
    if (d->has_residual && !residual) return SI_E_BADARG;
    if ((long long)d->n * d->oh * d->ow > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    if ((long long)d->n * d->ih * d->iw > 0x7fffffffLL) return SI_E_UNSUPPORTED;

    // 1..3 input channels: the stem kernel (its weight layout is its own, so there is no falling through)
    if (si_conv_smallc_ok(d)) {
        if (yolo || split) return SI_E_UNSUPPORTED;
        return si_conv_smallc_launch(d, in, w_packed, bias, residual, out, static_cast<hipStream_t>(stream));
    }
    // depthwise: one channel per group, HBM-bound (its weight layout is its own as well)
    if (si_conv_depthwise_ok(d)) {
        if (yolo || split) return SI_E_UNSUPPORTED;
        return si_conv_depthwise_launch(d, in, w_packed, bias, residual, out, static_cast<hipStream_t>(stream));
    }

    ConvArgs a;
    a.in = in;
    a.w = w_packed;
    a.bias = d->has_bias ? bias : nullptr;
    a.res = d->has_residual ? residual : nullptr;
    a.out = out;
    a.ih = d->ih; a.iw = d->iw; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.dh = d->dh; a.dw = d->dw;
    a.pt = d->pt; a.pl = d->pl;
    a.icg = d->ic / d->groups;
    a.icg_pad = conv_icg_pad(d);
    a.ocg = d->oc / d->groups;
    a.oc = d->oc;
    a.Kp = d->kh * d->kw * a.icg_pad;
    a.M = d->n * d->oh * d->ow;
    a.ohow = d->oh * d->ow;
    a.mg_ohow = a.ohow > 1 ? (unsigned)(0x100000000ull / (unsigned)a.ohow) : 0xFFFFFFFFu;
    a.mg_ow = d->ow > 1 ? (unsigned)(0x100000000ull / (unsigned)d->ow) : 0xFFFFFFFFu;
    a.m_tiles = a.n_tiles = 0;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    a.cb_major = conv_cb_major(d) ? 1 : 0;
    a.ntaps = d->kh * d->kw;
    a.pointwise = (d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->pl == 0 && d->ih == d->oh && d->iw == d->ow) ? 1 : 0;
    a.in_bytes = 0; a.w_bytes = 0;
    a.ymode = 0; a.yna = a.yne = a.yrows_total = a.yrow_off = 0; a.ystride = 0.f; a.ygrid = a.yanchor = nullptr;
    a.out2 = nullptr; a.out2_ld = 0; a.split = 0;
    a.up = nullptr; a.up_ih = a.up_iw = a.up_ld = a.up_cb0 = a.up_cb1 = 0; a.up_inv_h = a.up_inv_w = 0.f; a.up_bytes = 0;
    if (up) {
        // channels [c0, c0 + c) of this 1x1 conv's input are nn.Upsample(nearest) of up->src: read them at the source
        if (!up->src || yolo || !conv_upcat_shape_ok(d, up) || (reinterpret_cast<uintptr_t>(up->src) & 15) != 0 || !conv_fast_ok(d, in))
            return SI_E_UNSUPPORTED;
        const unsigned long long ub = (unsigned long long)d->n * up->ih * up->iw * up->ld * 4ull;
        a.up = up->src; a.up_ih = up->ih; a.up_iw = up->iw; a.up_ld = up->ld; a.up_cb0 = up->c0 / 32; a.up_cb1 = (up->c0 + up->c) / 32;
        a.up_inv_h = up->inv_scale_h; a.up_inv_w = up->inv_scale_w; a.up_bytes = (unsigned)ub;
    }
    if (split) {
        if (d->groups != 1 || split->split <= 0 || split->split >= d->oc || split->split % 32 != 0 || !split->out2) return SI_E_BADARG;
        a.out2 = split->out2; a.out2_ld = split->out2_ld; a.split = split->split;
    }
    if (yolo) {
        // the decode epilogue lives in the fast kernel only; the layer falls back to conv + decode otherwise
        if (!conv_fast_ok(d, in) || d->groups != 1 || d->has_residual || yolo->na * yolo->ne != d->oc || !a.pointwise || d->ic % 32 != 0) return SI_E_UNSUPPORTED;
        a.ymode = 1; a.yna = yolo->na; a.yne = yolo->ne; a.yrows_total = yolo->rows_total; a.yrow_off = yolo->row_off;
        a.ystride = yolo->stride; a.ygrid = ygrid; a.yanchor = yanchor;
    }

    if (conv_fast_ok(d, in)) {
        a.in_bytes = (unsigned)((unsigned long long)d->n * d->ih * d->iw * d->in_ld * 4ull);
        hipStream_t fs = static_cast<hipStream_t>(stream);
        // a zero-padded K axis only exists in the two default tiles (SI_CONV_VARIANT is ignored for those layers)
        int variant = conv_variant(d);
        if (a.ymode && !(variant == 4 || variant >= 11)) variant = 4;   // Detect: the default tile or a 16x16-MFMA one
        else if ((a.icg % 32 != 0 || a.up) && !conv_variant_full(variant)) variant = (d->oc / d->groups) <= 32 ? 10 : 4;
        switch (variant) {
            case 0: return launch_fast<128, 128, 2, 2, 2>(a, d->groups, fs);
            case 1: return launch_fast<128, 64, 2, 2, 2>(a, d->groups, fs);
            case 2: return launch_fast<64, 64, 2, 2, 2>(a, d->groups, fs);
            case 3: return launch_fast<128, 32, 4, 1, 2>(a, d->groups, fs);
            case 4: return launch_fast<64, 64, 2, 2, 1>(a, d->groups, fs);
            case 5: return launch_fast<64, 128, 2, 2, 1>(a, d->groups, fs);
            case 6: return launch_fast<128, 64, 2, 2, 1>(a, d->groups, fs);
            case 7: return launch_fast<128, 128, 2, 2, 1>(a, d->groups, fs);
            case 8: return launch_fast<64, 128, 2, 2, 2>(a, d->groups, fs);
            case 11: return launch_fast<32, 64, 2, 2, 2, 16>(a, d->groups, fs);
            case 12: return launch_fast<32, 32, 2, 2, 2, 16>(a, d->groups, fs);
            case 13: return launch_fast<32, 64, 2, 2, 1, 16>(a, d->groups, fs);
            case 14: return launch_fast<32, 32, 2, 2, 1, 16>(a, d->groups, fs);
            case 16: return launch_fast<64, 32, 2, 2, 1, 16>(a, d->groups, fs);
            case 17: return launch_fast<64, 64, 2, 2, 1, 16>(a, d->groups, fs);
            case 19: return launch_fast<64, 32, 4, 1, 1, 16>(a, d->groups, fs);
            case 22: return launch_fast<32, 32, 2, 2, 1, 16, 2>(a, d->groups, fs);
            default: return launch_fast<128, 32, 4, 1, 1>(a, d->groups, fs);
        }
    }

    const bool vec_a = conv_vec_a(d, in);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int G = d->groups;
    switch (conv_generic_tile(d)) {
        case 0: return launch<128, 128, 2, 2>(a, G, vec_a, s);
        case 1: return launch<128, 64, 2, 2>(a, G, vec_a, s);
        case 2: return launch<64, 64, 2, 2>(a, G, vec_a, s);
        default: return launch<128, 32, 4, 1>(a, G, vec_a, s);
    }
}

extern "C" int si_hip_conv2d_f32(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                                 const float* residual, float* out, si_stream_t stream) {
    return conv2d_dispatch(d, in, w_packed, bias, residual, out, stream, nullptr, nullptr, nullptr);
}

extern "C" int si_hip_conv2d_split_f32(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                                       float* out, int split_oc, float* out2, int out2_ld, si_stream_t stream) {
    if (!d || d->has_residual) return SI_E_BADARG;
    SplitOut sp{out2, out2_ld, split_oc};
    return conv2d_dispatch(d, in, w_packed, bias, nullptr, out, stream, nullptr, nullptr, nullptr, &sp);
}

extern "C" int si_hip_conv2d_upcat_f32(const SiConv2dDesc* d, const float* in, const SiConv2dUpsampledSource* up, const float* w_packed,
                                       const float* bias, float* out, int split_oc, float* out2, int out2_ld, si_stream_t stream) {
    if (!d || !up || d->has_residual) return SI_E_BADARG;
    if (split_oc > 0) {
        SplitOut sp{out2, out2_ld, split_oc};
        return conv2d_dispatch(d, in, w_packed, bias, nullptr, out, stream, nullptr, nullptr, nullptr, &sp, up);
    }
    return conv2d_dispatch(d, in, w_packed, bias, nullptr, out, stream, nullptr, nullptr, nullptr, nullptr, up);
}

extern "C" int si_hip_conv2d_yolo_f32(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                                      const SiYoloLevel* level, const float* grid_hwa2, const float* anchor_hwa2,
                                      float* detect_out, si_stream_t stream) {
    if (!level || !grid_hwa2 || !anchor_hwa2 || level->ne < 4 || level->na <= 0) return SI_E_BADARG;
    return conv2d_dispatch(d, in, w_packed, bias, nullptr, detect_out, stream, level, grid_hwa2, anchor_hwa2);
}

// The kernel instantiation a launch of this problem runs, exactly as rocprofv3 prints it (minus the namespace):
// conv_igemm_f32_fast_kernel<BM, BN, WM, WN, NBUF, PADK, UPS, YOLO, PW, MT>.  form: 0 si_hip_conv2d_f32 / _split_f32,
// 1 si_hip_conv2d_upcat_f32 (dual-source), 2 si_hip_conv2d_yolo_f32 (Detect epilogue).
extern "C" const char* si_hip_conv2d_kernel_name_form(const SiConv2dDesc* d, const float* in, int form) {
    if (!d || d->groups <= 0) return "invalid";
    static const char* names[4][2] = {
        {"conv_igemm_f32_kernel<128, 128, 2, 2, false>", "conv_igemm_f32_kernel<128, 128, 2, 2, true>"},
        {"conv_igemm_f32_kernel<128, 64, 2, 2, false>", "conv_igemm_f32_kernel<128, 64, 2, 2, true>"},
        {"conv_igemm_f32_kernel<64, 64, 2, 2, false>", "conv_igemm_f32_kernel<64, 64, 2, 2, true>"},
        {"conv_igemm_f32_kernel<128, 32, 4, 1, false>", "conv_igemm_f32_kernel<128, 32, 4, 1, true>"}};
    // {BM, BN, WM, WN, NBUF, MT, KU}
    static const int tile_of[kConvVariants][7] = {
        {128, 128, 2, 2, 2, 32, 1}, {128, 64, 2, 2, 2, 32, 1}, {64, 64, 2, 2, 2, 32, 1},  {128, 32, 4, 1, 2, 32, 1}, {64, 64, 2, 2, 1, 32, 1},
        {64, 128, 2, 2, 1, 32, 1},  {128, 64, 2, 2, 1, 32, 1}, {128, 128, 2, 2, 1, 32, 1}, {64, 128, 2, 2, 2, 32, 1}, {128, 32, 4, 1, 1, 32, 1},
        {128, 32, 4, 1, 1, 32, 1},  {32, 64, 2, 2, 2, 16, 1},  {32, 32, 2, 2, 2, 16, 1},  {32, 64, 2, 2, 1, 16, 1},  {32, 32, 2, 2, 1, 16, 1},
        {64, 32, 2, 2, 2, 16, 1},   {64, 32, 2, 2, 1, 16, 1},  {64, 64, 2, 2, 1, 16, 1},  {32, 128, 2, 2, 1, 16, 1}, {64, 32, 4, 1, 1, 16, 1},
        {128, 64, 2, 2, 1, 16, 1},  {64, 128, 2, 2, 1, 16, 1}, {32, 32, 2, 2, 1, 16, 2}};
    // [variant][instantiation]: 0 general, 1 pointwise, 2 zero-padded K, 3 dual-source, 4 Detect
    static char fast_names[kConvVariants][5][112];
    static const bool named = [] {
        static const char* flags[5] = {"false, false, false, false", "false, false, false, true", "true, false, false, false",
                                       "false, true, false, false", "false, false, true, false"};
        for (int v = 0; v < kConvVariants; ++v)
            for (int k = 0; k < 5; ++k)
                snprintf(fast_names[v][k], sizeof(fast_names[v][k]), "conv_igemm_f32_fast_kernel<%d, %d, %d, %d, %d, %s, %d, %d>", tile_of[v][0],
                         tile_of[v][1], tile_of[v][2], tile_of[v][3], tile_of[v][4], flags[k], tile_of[v][5], tile_of[v][6]);
        return true;
    }();
    (void)named;
    if (si_conv_smallc_ok(d)) return si_conv_smallc_name(d);
    if (si_conv_depthwise_ok(d)) return si_conv_depthwise_name(d);
    SiConv2dDesc eff = conv_effective(d);
    d = &eff;
    if (conv_fast_ok(d, in)) {
        // the dispatch rule of conv2d_dispatch
        int v = conv_variant(d);
        const bool padk = (d->ic / d->groups) % 32 != 0;
        const bool pointwise = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->pl == 0 && d->ih == d->oh && d->iw == d->ow;
        if (form == 2) return fast_names[(v == 4 || v >= 11) ? v : 4][4];
        if ((padk || form == 1) && !conv_variant_full(v)) v = (d->oc / d->groups) <= 32 ? 10 : 4;
        if (form == 1) return fast_names[v][3];
        if (padk) return fast_names[v][2];
        return fast_names[v][(pointwise && conv_variant_full(v)) ? 1 : 0];
    }
    return names[conv_generic_tile(d)][conv_vec_a(d, in) ? 1 : 0];
}

extern "C" const char* si_hip_conv2d_kernel_name(const SiConv2dDesc* d, const float* in) { return si_hip_conv2d_kernel_name_form(d, in, 0); }
