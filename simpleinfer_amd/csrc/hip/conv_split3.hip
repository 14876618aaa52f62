// conv_split3.hip -- fp32 convolution on the fp16 matrix cores by operand splitting (round 5; opt-in, never the default path).
//
// The fp32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2) runs at 1/16 of the fp16 rate and the fp32 implicit GEMM has sat at 0.65 of it for
// three rounds.  The published way past an fp32 matrix peak (Ootomo & Yokota 2022, "Recovering single precision accuracy from Tensor
// Cores while surpassing the FP32 theoretical peak performance") splits every operand into two fp16 halves,
//     a = a_hi + 2^-11 a_lo,   a_hi = fp16(a),   a_lo = fp16((a - a_hi) * 2^11)          (22 significant bits; a - a_hi is exact in fp32)
// and forms a * b from THREE fp16 products with exact 22-bit results accumulated in fp32:
//     a b  ~  a_hi b_hi  +  2^-11 (a_hi b_lo + a_lo b_hi)                                   (the a_lo b_lo term is 2^-22 relative: dropped)
// The two scales get their own accumulators (the small terms do not drown in the large sum) and meet once, in the epilogue.
// Weights are split once at load (two lane-order images, as conv_igemm_f16.hip's "bd" kernels read them: L2 -> registers, one coalesced
// 16-byte load per lane and fragment); activations are fp32 in HBM -- the same bytes as the fp32 path -- and are split on their way into
// LDS.  Structure: conv_igemm_f16_bd_kernel's (A through two LDS stages, one barrier per 64-channel K-tile, 1 x 4 waves of 64 x 32).
// Not bit-compatible with the fp32 kernels (another arithmetic); error against the fp64 convolution is measured by tests / tools
// (tools/split3_check.py).  Operands must lie in fp16's range: weights are checked when they are split (the pack function refuses), activations by
// the kernel itself (split_range_report below, SiConv2dDesc::range_flag; include/si_hip.h has the contract); values below 6e-5 lose relative, not
// absolute, accuracy.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "si_hip.h"
#include "si_hip_internal.h"

#pragma clang fp contract(off)

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct Split3Args {
    const float* in;
    const half_t* wl_hi;        // lane-order images [nb][ks][64 lanes][8]: hi halves, then lo halves (scaled by 2^11)
    const half_t* wl_lo;
    int wl_nb, wl_ks;
    const float* bias;
    const float* res;
    float* out;
    int ih, iw, in_ld, oh, ow, out_ld, res_ld;
    int kh, kw, sh, sw, pt, pl;
    int ic, oc, Kp, M, ohow;
    unsigned mg_ohow, mg_ow;
    int m_tiles, n_tiles;
    int act1, act2;
    float act_param;
    unsigned in_bytes;
    // YOLO: the Detect decode in the epilogue (si_yolo_tile_one_image's argument names; conv_igemm.hip ConvArgs)
    int ocg;
    int yna, yne, yrows_total, yrow_off;
    float ystride;
    const float* ygrid;
    const float* yanchor;
    unsigned* range_flag;   // SiConv2dDesc::range_flag: set to 1 when an accumulator left the matrix cores non-finite (an operand overflowed fp16)
    // split destination (si_hip_conv2d_split3_split_f32: two sibling convs as one): output channels [split, oc) go to out2 (stride out2_ld)
    float* out2;
    int out2_ld, split;
    // dual-source rows (si_hip_conv2d_split3_upcat_f32, round 6): K-tiles [up_cb0, up_cb1) of this 1x1 conv's input are nn.Upsample(nearest) of
    // `up` -- read at the row's source pixel of the LOW-RESOLUTION tensor (conv_igemm.hip's UPS form: reference src/layer/upsample.cpp:85-92)
    const float* up;
    int up_ih, up_iw, up_ld, up_cb0, up_cb1;
    float up_inv_h, up_inv_w;
    unsigned up_bytes;
};

// An operand that rounds to fp16 infinity makes its hi half Inf and its lo half Inf / NaN, so every accumulator it feeds is Inf or NaN (the fp16
// MFMA follows IEEE for both): testing the COMBINED accumulators, before bias and activation (relu(NaN) = 0, sigmoid(Inf) = 1 would hide it), finds
// every overflow at one v_cmp_class per output element and nothing per input element.  Wave-level: one store by one lane, only when it trips.
__device__ __forceinline__ void split_range_report(bool bad, unsigned* flag) {
    if (flag && __builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) *reinterpret_cast<volatile unsigned*>(flag) = 1u;
}

__device__ __forceinline__ int fdiv(int n, int d, unsigned mg) {
    unsigned q = __umulhi((unsigned)n, mg);
    if ((unsigned)n - q * (unsigned)d >= (unsigned)d) ++q;
    return (int)q;
}

__device__ __forceinline__ float act_rt(int act, float v, float p) {
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

constexpr unsigned OOB_A = 0xFFFFFF00u;
constexpr unsigned OOB_B = 0x80000000u;
constexpr float kLoScale = 2048.0f;   // 2^11

// BM x (32 WN) workgroup tile, WM x WN waves (a wave: BM / WM rows x 32 columns of its own -- 1 x 4 for >= 128 output channels: every
// weight fragment fetched once per workgroup; 2 x 2 for 64), BKH-channel K-tiles (64; 32 for 32-channel layers)
// YOLO: a Detect level (1x1 conv to na * ne columns) with the decode of src/layer/yolo_detect.cpp:223-266 in the epilogue, written into the
// [n][rows_total][ne] detections -- the fp32 kernels' epilogue (si_yolo_tile_one_image) on the combined accumulators
// UPS: a pointwise conv some of whose K-tiles come from a low-resolution tensor at the row's nearest-neighbour source pixel (the upsample + concat
// in front of YOLOv5's PAN convs, read at the source)
template <int BM, int WM, int WN, int BKH, bool YOLO = false, bool UPS = false>
__global__ __launch_bounds__(256, 2) void conv_split3_f32_kernel(const Split3Args a) {
    static_assert(WM * WN == 4 && (BKH == 64 || BKH == 32), "4 waves; 64- or 32-channel K-tiles");
    constexpr int BN = 32 * WN, LDH = BKH + 8, QS = BKH / 16;
    constexpr int TM = BM / WM / 32;
    constexpr int VPR = BKH / 4, RPP = 256 / VPR;   // float4 vectors per row of a K-tile, rows per pass of the 256 threads
    constexpr int A_IT = BM / RPP;                  // vectors per thread and K-tile
    // LDS: two stages x {hi, lo} x [BM][BKH + 8] halves (dynamic: 72 KB for 128-row tiles of 64 channels)
    extern __shared__ __attribute__((aligned(16))) unsigned char split3_smem[];
    half_t (*lds)[2][BM * LDH] = reinterpret_cast<half_t (*)[2][BM * LDH]>(split3_smem);

    const int per_chunk = 8 * a.n_tiles;
    const int chunk = blockIdx.x / per_chunk;
    const int r = blockIdx.x - chunk * per_chunk;
    const int m_tile = chunk * 8 + (r & 7);
    const int n_tile = r >> 3;
    if (m_tile >= a.m_tiles) return;
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    const int tid = threadIdx.x;
    const int kv = tid % VPR, r0 = tid / VPR;   // this thread's 4-channel vector of a row, base row
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const unsigned wl_bytes = (unsigned)a.wl_nb * (unsigned)a.wl_ks * 1024u;
    const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.wl_hi), 0, wl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.wl_lo), 0, wl_bytes, 0x00020000);

    const __amdgpu_buffer_rsrc_t rs_up = UPS ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.up), 0, a.up_bytes, 0x00020000) : rs_in;
    unsigned a_off[A_IT];
    unsigned u_off[UPS ? A_IT : 1];   // UPS: byte offset of the row's source pixel in the low-resolution tensor
    unsigned a_mask[A_IT];   // tap validity bits (at most 32 taps)
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + RPP * i;
        a_off[i] = 0;
        a_mask[i] = 0u;
        if (UPS) u_off[i] = OOB_A;
        if (m < a.M) {
            const int img = fdiv(m, a.ohow, a.mg_ohow);
            const int rem = m - img * a.ohow;
            const int oy = fdiv(rem, a.ow, a.mg_ow);
            const int ox = rem - oy * a.ow;
            if (UPS) {
                // upsample.cpp:85-92: src = clamp(int(float(dst) * (1 / scale)), 0, in - 1)
                int sy = (int)((float)oy * a.up_inv_h), sx = (int)((float)ox * a.up_inv_w);
                sy = max(0, min(a.up_ih - 1, sy));
                sx = max(0, min(a.up_iw - 1, sx));
                u_off[i] = (unsigned)((img * a.up_ih + sy) * a.up_iw + sx) * (unsigned)(a.up_ld * 4) + (unsigned)(kv * 16);
            }
            const int y0 = oy * a.sh - a.pt, x0 = ox * a.sw - a.pl;
            a_off[i] = (unsigned)((img * a.ih + y0) * a.iw + x0) * (unsigned)(a.in_ld * 4) + (unsigned)(kv * 16);
            unsigned mk = 0u;
            for (int ky = 0; ky < a.kh; ++ky)
                for (int kx = 0; kx < a.kw; ++kx) {
                    const int y = y0 + ky, x = x0 + kx;
                    if ((unsigned)y < (unsigned)a.ih && (unsigned)x < (unsigned)a.iw) mk |= 1u << (ky * a.kw + kx);
                }
            a_mask[i] = mk;
        }
    }
    const int nb = (n0 >> 5) + wn;
    const unsigned b_off = nb < a.wl_nb ? (unsigned)nb * (unsigned)a.wl_ks * 1024u + (unsigned)lane * 16u : OOB_B;

    const int nk = a.Kp / BKH;
    int cb = 0, ky = 0, kx = 0;
    u32x4 ra[A_IT];
    // weight fragments: a ring of NBF k-steps (hi and lo), the step NBF - 1 ahead requested while this one multiplies; a step behind the
    // last one is an out-of-range offset
    constexpr int NBF = 4;
    static_assert(QS % NBF == 0 || NBF % QS == 0, "ring slots are compile-time per step of a K-tile");
    f16x8 rbh[NBF], rbl[NBF];
    const int ks_tot = nk * QS;
    auto load_b = [&](int slot, int ks) {
        const bool live = ks < ks_tot;
        const unsigned kb = (unsigned)(live ? ks : 0) * 1024u;
        const unsigned vo = live ? b_off : OOB_B;
        rbh[slot] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_hi, vo, kb, 0));
        rbl[slot] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_lo, vo, kb, 0));
    };
    auto load_a = [&](int kt) {
        const unsigned delta = (unsigned)((ky * a.iw + kx) * a.in_ld + cb * BKH) * 4u;
        const int tapbit = ky * a.kw + kx;
        const bool live = kt < nk;
        const bool from_up = UPS && cb >= a.up_cb0 && cb < a.up_cb1;   // wave-uniform: this K-tile's channels are upsampled ones
        if (from_up) {
            const unsigned du = (unsigned)(cb - a.up_cb0) * (unsigned)(BKH * 4);
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_up, (live && u_off[i] != OOB_A) ? u_off[i] + du : OOB_A, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const bool ok = ((a_mask[i] >> tapbit) & 1u) && live;
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? a_off[i] + delta : OOB_A, 0, 0);
            }
        }
        ++kx;
        const int wx = kx == a.kw ? 1 : 0;
        kx = wx ? 0 : kx;
        ky += wx;
        const int wy = ky == a.kh ? 1 : 0;
        ky = wy ? 0 : ky;
        cb += wy;
    };
    // fp32 -> (hi, lo) on the way into LDS
    auto store_a = [&](int stage) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const f32x4 x = __builtin_bit_cast(f32x4, ra[i]);
            f16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hi[j] = (half_t)x[j];
                lo[j] = (half_t)((x[j] - (float)hi[j]) * kLoScale);
            }
            const int row = r0 + RPP * i;
            *reinterpret_cast<f16x4*>(lds[stage][0] + row * LDH + kv * 4) = hi;
            *reinterpret_cast<f16x4*>(lds[stage][1] + row * LDH + kv * 4) = lo;
        }
    };

    f32x16 acc_h[TM], acc_x[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_h[t][e] = acc_x[t][e] = 0.0f;

    load_a(0);
#pragma unroll
    for (int j = 0; j + 1 < NBF; ++j) load_b(j, j);
    const int o = n0 + wn * 32 + l31;
    const float bv = (a.bias && o < a.oc) ? a.bias[o] : 0.0f;
    store_a(0);
    load_a(1);
    __syncthreads();

    auto k_tile = [&](int kt, auto ph) {
        constexpr int cur = decltype(ph)::value & 1;
        store_a(cur ^ 1);           // tile kt + 1 (requested one tile ago) into the other stage
        load_a(kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        const half_t* Ah = lds[cur][0] + (wm * TM * 32 + l31) * LDH + lh * 8;
        const half_t* Al = lds[cur][1] + (wm * TM * 32 + l31) * LDH + lh * 8;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            load_b((cur * QS + s + NBF - 1) % NBF, kt * QS + s + NBF - 1);
            f16x8 fh[TM], fl[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                fh[t] = *reinterpret_cast<const f16x8*>(Ah + t * 32 * LDH + s * 16);
                fl[t] = *reinterpret_cast<const f16x8*>(Al + t * 32 * LDH + s * 16);
            }
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                acc_h[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[t], rbh[(cur * QS + s) % NBF], acc_h[t], 0, 0, 0);
                acc_x[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[t], rbl[(cur * QS + s) % NBF], acc_x[t], 0, 0, 0);
                acc_x[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[t], rbh[(cur * QS + s) % NBF], acc_x[t], 0, 0, 0);
            }
        }
        __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        k_tile(kt, std::integral_constant<int, 0>{});
        k_tile(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < nk) k_tile(kt, std::integral_constant<int, 0>{});

    if constexpr (YOLO) {
        f32x16 accc[TM][1];
        bool bad = false;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                accc[t][0][e] = acc_h[t][e] + acc_x[t][e] * (1.0f / kLoScale);
                bad |= !__builtin_isfinite(accc[t][0][e]);
            }
        split_range_report(bad, a.range_flag);
        const int mrow0 = m0 + wm * TM * 32 + 4 * lh;
        const int img0 = fdiv(m0, a.ohow, a.mg_ohow);
        if (m0 + BM <= a.M && m0 - img0 * a.ohow + BM <= a.ohow) {   // (workgroup-uniform: the tile lies inside one image)
            si_yolo_tile_one_image<TM, 1>(a, a.out, accc, mrow0, o, img0);
        } else if (o < a.oc) {
            // a tile across image borders (the 20 x 20 level: 400 pixels per image) or the last, partial one: element by element
            const int per_pix = a.yna * a.yne;
            const int anc = o / a.yne, e_ = o - anc * a.yne;
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mrow0 + t * 32 + (e & 3) + 8 * (e >> 2);
                    if (m < a.M) {
                        const int img = fdiv(m, a.ohow, a.mg_ohow), pix = m - img * a.ohow;
                        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-(accc[t][0][e] + bv)));
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
        return;
    }
    // epilogue: the two scales meet, then bias / activation / shortcut / activation (C/D map: col = lane & 31, rows (e & 3) + 8 (e >> 2) + 4 lh)
    {
        // (rows / columns outside the problem multiply zeros: they are finite and cost nothing to include)
        bool bad = false;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc_h[t][e] = acc_h[t][e] + acc_x[t][e] * (1.0f / kLoScale);
                bad |= !__builtin_isfinite(acc_h[t][e]);
            }
        split_range_report(bad, a.range_flag);
    }
    if (o < a.oc) {
        // (split destination: this lane's column lives in one of the two tensors)
        const bool second = a.split > 0 && o >= a.split;
        float* const ob = second ? a.out2 + (o - a.split) : a.out + o;
        const int old = second ? a.out2_ld : a.out_ld;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int mb = m0 + (wm * TM + t) * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m < a.M) {
                    float v = acc_h[t][e];
                    v = act_rt(a.act1, v + bv, a.act_param);
                    if (a.res) v += a.res[(size_t)m * a.res_ld + o];
                    ob[(size_t)m * old] = act_rt(a.act2, v, a.act_param);
                }
            }
        }
    }
}


// ---- a Detect level as ONE tile shape of its own, on the split arithmetic (round 6, late) ----------------------------------------------------
// detect_f16_tile_kernel's structure (conv_igemm_f16.hip): a Detect level is a 1x1 conv to na * ne = 255 columns whose decoded fp32 result is
// 8.5 x (K = 128: 2 x) its fp32 input -- a WRITE stream.  Through the YOLO form of the kernel above level 0 (80x80x128 at batch 32: 105 MB in, 209 MB out)
// ran as 12 800 workgroups of two K-tiles, every one a load -> split -> LDS -> barrier -> MFMA -> decode chain ending in 4-byte stores 1020 bytes
// apart: 280 us beside the neck's launches against a 63 us stream.  Here a workgroup owns 64 CONSECUTIVE PIXELS of one image and ALL columns (one
// contiguous run of 64 * 255 floats of the output: src/layer/yolo_detect.cpp:223-266 writes the same bytes); the whole A tile is split once into two
// LDS images, the weights come from the lane-order images (L2 -> registers, four waves x 64 columns), and the decoded values are staged through LDS
// in two 32-pixel halves and leave as 16-byte-per-lane stores down the run.  The two halves are multiplied one after the other (two accumulator
// sets per 32 x 32 block: both halves at once would be 128 registers of accumulators).  Same k-steps in the same order on the same two chains and
// the same decode expressions as the YOLO form above: the same bits (tests/test_gpu_ops.py).  NCH = K / 128 (1, 2: the K = 512 level is 400 pixels
// per image and stays on the YOLO form -- its A images alone would be 133 KB of LDS).
template <int NCH>
__global__ __launch_bounds__(256, 2) void detect_split_tile_kernel(const Split3Args a) {
    constexpr int K = NCH * 128;
    constexpr int LDH = K + 8;            // halves per LDS row: (2K + 16) mod 128 = 16, the conflict-free pitch
    constexpr int VPR = K / 4;            // 16-byte fp32 vectors per row
    constexpr int A_IT = 64 * VPR / 256;  // vectors per thread
    constexpr int KS = K / 16;            // MFMA k-steps
    constexpr int RING = 4;               // weight fragment pairs in flight, in k-steps
    extern __shared__ __attribute__((aligned(16))) unsigned char det_smem[];
    half_t* const Ah = reinterpret_cast<half_t*>(det_smem);
    half_t* const Al = Ah + 64 * LDH;
    float* const stage = reinterpret_cast<float*>(det_smem);   // 32 pixels x 255 floats, over the A images once the MFMAs are done

    const int tiles_per_img = (a.ohow + 63) >> 6;
    const int img = blockIdx.x / tiles_per_img;
    const int pix0 = (blockIdx.x - img * tiles_per_img) * 64;
    const int valid = min(64, a.ohow - pix0);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const unsigned wl_bytes = (unsigned)a.wl_nb * (unsigned)a.wl_ks * 1024u;
    const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.wl_hi), 0, wl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.wl_lo), 0, wl_bytes, 0x00020000);

    // the A tile: fp32 -> (hi, lo) on the way into LDS, eight vectors per thread at a time
    static_assert(A_IT % 8 == 0, "A tile in batches of eight vectors per thread");
#pragma unroll
    for (int c = 0; c < A_IT; c += 8) {
        u32x4 ra[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * (c + i);
            const int row = idx / VPR, kv = idx - row * VPR;
            const unsigned off = (unsigned)(img * a.ohow + pix0 + row) * (unsigned)(a.in_ld * 4) + (unsigned)(kv * 16);
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, row < valid ? off : OOB_A, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * (c + i);
            const int row = idx / VPR, kv = idx - row * VPR;
            const f32x4 x = __builtin_bit_cast(f32x4, ra[i]);
            f16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hi[j] = (half_t)x[j];
                lo[j] = (half_t)((x[j] - (float)hi[j]) * kLoScale);
            }
            *reinterpret_cast<f16x4*>(Ah + row * LDH + kv * 4) = hi;
            *reinterpret_cast<f16x4*>(Al + row * LDH + kv * 4) = lo;
        }
    }
    unsigned b_off[2];
    float bv[2];
    int col[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int nb = wave * 2 + u;
        b_off[u] = nb < a.wl_nb ? (unsigned)nb * (unsigned)a.wl_ks * 1024u + (unsigned)lane * 16u : OOB_B;
        col[u] = wave * 64 + u * 32 + l31;
        bv[u] = (a.bias && col[u] < a.ocg) ? a.bias[col[u]] : 0.0f;
    }
    __syncthreads();

    f32x16 accc[2][2];   // the combined accumulators of the two 32-pixel halves
    bool bad = false;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x16 acc_h[2], acc_x[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_h[u][e] = acc_x[u][e] = 0.0f;
        f16x8 rbh[RING][2], rbl[RING][2];
#pragma unroll
        for (int s = 0; s < RING; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                rbh[s][u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_hi, b_off[u], (unsigned)(s * 1024), 0));
                rbl[s][u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_lo, b_off[u], (unsigned)(s * 1024), 0));
            }
        const half_t* const Arh = Ah + (32 * t + l31) * LDH + lh * 8;
        const half_t* const Arl = Al + (32 * t + l31) * LDH + lh * 8;
        f16x8 fh[2], fl[2];
        fh[0] = *reinterpret_cast<const f16x8*>(Arh);
        fl[0] = *reinterpret_cast<const f16x8*>(Arl);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
                fh[(s + 1) & 1] = *reinterpret_cast<const f16x8*>(Arh + (s + 1) * 16);
                fl[(s + 1) & 1] = *reinterpret_cast<const f16x8*>(Arl + (s + 1) * 16);
            }
            const f16x8 bh[2] = {rbh[s % RING][0], rbh[s % RING][1]}, bl[2] = {rbl[s % RING][0], rbl[s % RING][1]};
            if (s + RING < KS) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    rbh[s % RING][u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_hi, b_off[u], (unsigned)((s + RING) * 1024), 0));
                    rbl[s % RING][u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_lo, b_off[u], (unsigned)((s + RING) * 1024), 0));
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                acc_h[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[s & 1], bh[u], acc_h[u], 0, 0, 0);
                acc_x[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[s & 1], bl[u], acc_x[u], 0, 0, 0);
                acc_x[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[s & 1], bh[u], acc_x[u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                accc[t][u][e] = acc_h[u][e] + acc_x[u][e] * (1.0f / kLoScale);
                bad |= !__builtin_isfinite(accc[t][u][e]);
            }
    }
    split_range_report(bad, a.range_flag);

    // decode (si_yolo_tile_one_image's expressions) -> LDS -> the contiguous run, 32 pixels at a time (detect_f16_tile_kernel's epilogue)
    const int per_pix = a.yna * a.yne;
    const bool coco = a.yna == 3 && a.yne == 85 && valid == 64;   // workgroup-uniform
    float* const orun = a.out + ((size_t)img * a.yrows_total + a.yrow_off) * a.yne + (size_t)pix0 * per_pix;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        __syncthreads();   // t = 0: every wave is done reading A; t = 1: the first half has left the stage
        const int rows = min(32, valid - 32 * t);
        if (coco) {
#pragma clang fp contract(off)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int blk = wave * 2 + u;   // wave-uniform
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = __builtin_amdgcn_rcpf(1.0f + __expf(-(accc[t][u][e] + bv[u])));
                if (blk == 0 || blk == 2 || blk == 5) {   // (the 32-column blocks that hold x, y, w, h of an anchor: columns 0-3, 85-88, 170-173)
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
#pragma clang fp contract(off)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool live = col[u] < a.ocg;
                const int oo = live ? col[u] : 0;
                const int anc = oo / a.yne;
                const int e_ = oo - anc * a.yne;
                const bool is_xy = e_ < 2, is_box = e_ < 4;
                const float* const ap = (is_xy ? a.ygrid + e_ : a.yanchor + (is_box ? e_ - 2 : 0)) + anc * 2 + (size_t)(pix0 + 32 * t) * a.yna * 2;
                float auxv[16];
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
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-(accc[t][u][e] + bv[u])));
                    const float aux = auxv[e];
                    const float t2 = sg * 2.0f;
                    const float xy = (t2 + aux) * a.ystride;
                    const float wh = t2 * t2 * aux;
                    const float v = is_xy ? xy : (is_box ? wh : sg);
                    if (live && dm < rows) stage[dm * per_pix + oo] = v;
                }
            }
        }
        __syncthreads();
        if (rows > 0) {
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

bool split3_ok(const SiConv2dDesc* d) {
    return d && d->groups == 1 && d->ic > 0 && d->ic % 32 == 0 && d->oc > 0 && d->kh * d->kw <= 32 && d->dh == 1 && d->dw == 1 && d->in_ld % 4 == 0;
}

// channels per K-tile: the K order is (c / blk, ky, kx, c % blk)
int split3_blk(const SiConv2dDesc* d) { return d->ic % 64 == 0 ? 64 : 32; }

}  // namespace

extern "C" {

int si_hip_conv2d_split3_supported(const SiConv2dDesc* d) { return split3_ok(d) ? 1 : 0; }

// two lane-order fp16 images of the same weights, hi then lo (each [nb][ks][64][8] halves)
size_t si_hip_conv2d_split3_weight_elems(const SiConv2dDesc* d) {
    if (!split3_ok(d)) return 0;
    return (size_t)2 * ((d->oc + 31) / 32) * (size_t)(d->kh * d->kw * d->ic / 16) * 512;
}

int si_hip_conv2d_split3_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, void* w_packed) {
    if (!d || !w_oihw || !w_packed) return SI_E_BADARG;
    if (!split3_ok(d)) return SI_E_UNSUPPORTED;
    const int ntaps = d->kh * d->kw, nb_n = (d->oc + 31) / 32, ks_n = ntaps * d->ic / 16, B = split3_blk(d);
    half_t* hi = static_cast<half_t*>(w_packed);
    half_t* lo = hi + (size_t)nb_n * ks_n * 512;
    for (int nb = 0; nb < nb_n; ++nb)
        for (int ks = 0; ks < ks_n; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    // K order (c / B, ky, kx, c % B), B = 64 (32 for channel counts that are not multiples of 64): k = ks * 16 + 8 (l >> 5) + j
                    const int k = ks * 16 + 8 * (l >> 5) + j;
                    const int blk = k / B, cb = blk / ntaps, tap = blk - cb * ntaps, c = cb * B + (k % B);
                    const int o = nb * 32 + (l & 31);
                    float v = 0.0f;
                    if (o < d->oc) v = w_oihw[(((size_t)o * d->ic + c) * d->kh + tap / d->kw) * d->kw + tap % d->kw];
                    const half_t h = (half_t)v;
                    // a weight that is not finite or rounds to fp16 infinity cannot be split: the layer stays on the fp32 kernels
                    if (!(__builtin_fabsf((float)h) <= 65504.0f)) return SI_E_UNSUPPORTED;
                    const size_t idx = (((size_t)nb * ks_n + ks) * 64 + l) * 8 + j;
                    hi[idx] = h;
                    lo[idx] = (half_t)((v - (float)h) * 2048.0f);
                }
    return 0;
}

static bool split3_upcat_shape_ok(const SiConv2dDesc* d, const SiConv2dUpsampledSource* up) {
    if (!d || !up || !split3_ok(d) || d->ic % 64 != 0 || d->oc <= 64) return false;
    const bool pointwise = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->pl == 0 && d->ih == d->oh && d->iw == d->ow;
    if (!pointwise || d->has_residual) return false;
    if (up->c <= 0 || up->c % 64 != 0 || up->c0 % 64 != 0 || up->c0 + up->c > d->ic || up->ld % 4 != 0 || up->ih <= 0 || up->iw <= 0) return false;
    return (unsigned long long)d->n * up->ih * up->iw * up->ld * 4ull < 0xFFFFFF00ull;
}

static int split3_launch(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, const float* residual, float* out,
                         si_stream_t stream, const SiYoloLevel* yolo, const float* ygrid, const float* yanchor, int split_oc = 0, float* out2 = nullptr,
                         int out2_ld = 0, const SiConv2dUpsampledSource* up = nullptr) {
    if (!d || !in || !w_packed || !out) return SI_E_BADARG;
    if (!split3_ok(d) || (reinterpret_cast<uintptr_t>(in) & 15) != 0) return SI_E_UNSUPPORTED;
    if ((d->has_bias && !bias) || (d->has_residual && !residual)) return SI_E_BADARG;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->ih * d->iw * d->in_ld * 4ull;
    if (in_bytes >= 0xFFFFFF00ull || (long long)d->n * d->oh * d->ow > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    Split3Args a;
    a.in = in;
    a.wl_nb = (d->oc + 31) / 32;
    a.wl_ks = d->kh * d->kw * d->ic / 16;
    a.wl_hi = static_cast<const half_t*>(w_packed);
    a.wl_lo = a.wl_hi + (size_t)a.wl_nb * a.wl_ks * 512;
    a.bias = d->has_bias ? bias : nullptr;
    a.res = d->has_residual ? residual : nullptr;
    a.out = out;
    a.ih = d->ih; a.iw = d->iw; a.in_ld = d->in_ld; a.oh = d->oh; a.ow = d->ow; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.pl = d->pl;
    a.ic = d->ic; a.oc = d->oc; a.Kp = d->kh * d->kw * d->ic;
    a.M = d->n * d->oh * d->ow;
    a.ohow = d->oh * d->ow;
    a.mg_ohow = a.ohow > 1 ? (unsigned)(0x100000000ull / (unsigned)a.ohow) : 0xFFFFFFFFu;
    a.mg_ow = d->ow > 1 ? (unsigned)(0x100000000ull / (unsigned)d->ow) : 0xFFFFFFFFu;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    a.in_bytes = (unsigned)in_bytes;
    a.ocg = d->oc;
    a.yna = a.yne = a.yrows_total = a.yrow_off = 0; a.ystride = 0.0f; a.ygrid = a.yanchor = nullptr;
    a.range_flag = d->range_flag;
    a.out2 = nullptr; a.out2_ld = 0; a.split = 0;
    a.up = nullptr; a.up_ih = a.up_iw = a.up_ld = a.up_cb0 = a.up_cb1 = 0; a.up_inv_h = a.up_inv_w = 0.0f; a.up_bytes = 0;
    if (up) {
        if (yolo || !up->src || !split3_upcat_shape_ok(d, up) || (reinterpret_cast<uintptr_t>(up->src) & 15) != 0) return SI_E_UNSUPPORTED;
        a.up = up->src; a.up_ih = up->ih; a.up_iw = up->iw; a.up_ld = up->ld; a.up_cb0 = up->c0 / 64; a.up_cb1 = (up->c0 + up->c) / 64;
        a.up_inv_h = up->inv_scale_h; a.up_inv_w = up->inv_scale_w;
        a.up_bytes = (unsigned)((unsigned long long)d->n * up->ih * up->iw * up->ld * 4ull);
    }
    if (split_oc > 0) {
        if (yolo || d->has_residual || split_oc >= d->oc || split_oc % 32 != 0 || !out2) return SI_E_BADARG;
        a.out2 = out2; a.out2_ld = out2_ld; a.split = split_oc;
    }
    if (yolo) {
        const bool pointwise = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->pl == 0 && d->ih == d->oh && d->iw == d->ow;
        if (!pointwise || d->has_residual || yolo->na * yolo->ne != d->oc || !ygrid || !yanchor || d->ic % 64 != 0 || d->oc <= 64) return SI_E_UNSUPPORTED;
        a.yna = yolo->na; a.yne = yolo->ne; a.yrows_total = yolo->rows_total; a.yrow_off = yolo->row_off; a.ystride = yolo->stride;
        a.ygrid = ygrid; a.yanchor = yanchor;
    }
    int cus = 256;
    {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    }
    auto go = [&](auto kern, int BM, int BN, int BKH) {
        a.m_tiles = (a.M + BM - 1) / BM;
        a.n_tiles = (d->oc + BN - 1) / BN;
        const int chunks = (a.m_tiles + 7) / 8;
        const size_t lds = (size_t)2 * 2 * BM * (BKH + 8) * 2;
        const hipError_t e = si_allow_dynamic_lds(kern, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(chunks * 8 * a.n_tiles), dim3(256), lds, static_cast<hipStream_t>(stream), a);
        return (int)hipGetLastError();
    };
    // Tiles.  <= 64 output channels: 2 x 2 waves over 64 columns (the 1 x 4 form would leave two waves without columns), 64 rows (128 on request);
    // otherwise 1 x 4 waves over 128 columns and -- round 6 -- THIRTY-TWO rows: measured on YOLOv5s under the option, same box, interleaved
    // (profiles/r06_f32_split.txt), 32-row tiles everywhere beat round 5's 64 / 128-row choice at every batch (batch 32 +1.7 %, 16 +2.8 %, 8 +6 %,
    // 4 +14 %): twice the workgroups, half the LDS each (18 KB: more of them per CU), and the K loop of a 32-row tile is the same three MFMAs per
    // fragment pair.  An output element is the same two accumulator chains over the same k order whatever the tile -- the same bits
    // (tests/test_gpu_ops.py) -- so an image's result does not depend on its batch.  SiConvPlan::split3_bm forces 32 / 64 / 128 for a call.
    const int forced_bm = (d->plan && d->plan->split3_bm > 0) ? d->plan->split3_bm : SI_ENV_INT("SI_SPLIT3_BM", 0);
    if (up) return forced_bm == 64 ? go(conv_split3_f32_kernel<64, 1, 4, 64, false, true>, 64, 128, 64) : go(conv_split3_f32_kernel<32, 1, 4, 64, false, true>, 32, 128, 64);
    // (the Detect form: 201 vs 206 us on 32-row tiles; <= 64 columns: the 2 x 2-wave form on 64 rows -- YOLOv5s conv_1 252 vs 266 us on 128)
    // (launch-size policy -- the two forms give the same bits: FOUR 64-pixel tiles per CU or more; below that the generic kernel's 32-row tiles fill
    // the chip as well or better: unconditionally the tile kernel measured -0.5 % at batch 4 and at batch 8 (two per CU), +1.0 % at batch 32;
    // with this threshold batch 32 keeps +0.9 % (level 0 alone), batch 16 / 8 / 4 are flat: profiles/r06_f32_split.txt)
    const bool force_detect_tile = d->plan && d->plan->split3_bm == -1;   // (tests: the tile kernel on launches the policy would not give it)
    if (yolo && forced_bm == 0 && (d->ic == 128 || d->ic == 256) && d->oc <= 256 &&
        (force_detect_tile || ((long long)((a.ohow + 63) >> 6) * d->n >= 4LL * cus && SI_ENV_INT("SI_SPLIT3_DETECT_TILE", 1)))) {
        // (the level as 64-pixel runs of the output: detect_split_tile_kernel; a forced tile height takes the YOLO form of the generic kernel)
        auto tile = [&](auto kern, int K) {
            const size_t a_bytes = (size_t)2 * 64 * (K + 8) * 2, st_bytes = (size_t)32 * d->oc * 4;
            const size_t lds = a_bytes > st_bytes ? a_bytes : st_bytes;
            const hipError_t e = si_allow_dynamic_lds(kern, lds);
            if (e != hipSuccess) return (int)e;
            const long long tiles = (long long)((a.ohow + 63) >> 6) * d->n;
            if (tiles > 0x7fffffffLL) return (int)SI_E_UNSUPPORTED;
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, static_cast<hipStream_t>(stream), a);
            return (int)hipGetLastError();
        };
        return d->ic == 128 ? tile(detect_split_tile_kernel<1>, 128) : tile(detect_split_tile_kernel<2>, 256);
    }
    if (yolo) return forced_bm == 64 ? go(conv_split3_f32_kernel<64, 1, 4, 64, true>, 64, 128, 64) : go(conv_split3_f32_kernel<32, 1, 4, 64, true>, 32, 128, 64);
    if (d->oc <= 64) {
        if (split3_blk(d) == 32) return forced_bm == 128 ? go(conv_split3_f32_kernel<128, 2, 2, 32>, 128, 64, 32) : go(conv_split3_f32_kernel<64, 2, 2, 32>, 64, 64, 32);
        return forced_bm == 128 ? go(conv_split3_f32_kernel<128, 2, 2, 64>, 128, 64, 64) : go(conv_split3_f32_kernel<64, 2, 2, 64>, 64, 64, 64);
    }
    if (split3_blk(d) == 32) return forced_bm == 32 ? go(conv_split3_f32_kernel<32, 1, 4, 32>, 32, 128, 32) : go(conv_split3_f32_kernel<64, 1, 4, 32>, 64, 128, 32);
    if (forced_bm == 128) return go(conv_split3_f32_kernel<128, 1, 4, 64>, 128, 128, 64);
    if (forced_bm == 64) return go(conv_split3_f32_kernel<64, 1, 4, 64>, 64, 128, 64);
    return go(conv_split3_f32_kernel<32, 1, 4, 64>, 32, 128, 64);
}

int si_hip_conv2d_split3_f32(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, const float* residual, float* out,
                             si_stream_t stream) {
    return split3_launch(d, in, w_packed, bias, residual, out, stream, nullptr, nullptr, nullptr);
}

int si_hip_conv2d_split3_split_f32(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, float* out, int split_oc,
                                   float* out2, int out2_ld, si_stream_t stream) {
    if (!d || d->has_residual || split_oc <= 0) return SI_E_BADARG;
    return split3_launch(d, in, w_packed, bias, nullptr, out, stream, nullptr, nullptr, nullptr, split_oc, out2, out2_ld);
}

int si_hip_conv2d_split3_upcat_supported(const SiConv2dDesc* d, const SiConv2dUpsampledSource* up) { return split3_upcat_shape_ok(d, up) ? 1 : 0; }

int si_hip_conv2d_split3_upcat_f32(const SiConv2dDesc* d, const float* in, const SiConv2dUpsampledSource* up, const void* w_packed, const float* bias,
                                   float* out, int split_oc, float* out2, int out2_ld, si_stream_t stream) {
    if (!d || !up || d->has_residual) return SI_E_BADARG;
    return split3_launch(d, in, w_packed, bias, nullptr, out, stream, nullptr, nullptr, nullptr, split_oc, out2, out2_ld, up);
}

int si_hip_conv2d_split3_yolo_f32(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, const SiYoloLevel* level,
                                  const float* grid, const float* anchor_grid, float* out, si_stream_t stream) {
    if (!level) return SI_E_BADARG;
    return split3_launch(d, in, w_packed, bias, nullptr, out, stream, level, grid, anchor_grid);
}

}  // extern "C"
