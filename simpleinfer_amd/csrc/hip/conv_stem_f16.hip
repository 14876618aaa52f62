// conv_stem_f16.hip -- first-layer ("stem") convolution of the fp16-storage path: fp32 image in (1..3 channels, what
// Engine::Input hands over), fp16 activations out, fp16 MFMA inside.
//
// Same reference shapes as conv_smallc.hip (YOLOv5s 640x640x3 -> 320x320x32 6x6 s2 p2, test/test_layer/
// test_conv_2d.cpp:279-293; ResNet18 7x7 s2 p3; MobileNet 3x3 s2 p1), same idea -- in NHWC the KW taps x C channels of
// one kernel row are KW*C contiguous values of an input row -- but everything around the MFMAs is re-balanced, because
// on v_mfma_f32_32x32x16_f16 the 108-deep contraction of the YOLOv5 stem costs 9 MFMAs (288 cycles) per 32x32 tile
// instead of 54 x 64 cycles, and the kernel is then bound by getting rows in and activations out:
//   * input rows are fetched as 16-byte vectors (the staged window starts at a multiple of 4 floats; every row of a
//     dense image whose width*channels is a multiple of 4 starts on a 16-byte boundary, so a vector is either inside
//     its row or entirely outside and then an out-of-range buffer offset returns zeros -- no per-element masks),
//     converted once and staged in LDS as fp16: 8 input rows serve 2 output rows of 160 pixels;
//   * a kernel row is cut into groups of 8 consecutive values (zero weights behind the last tap); one MFMA contracts two
//     groups, lanes 0-31 feed group 2s and lanes 32-63 group 2s+1, each lane reading its 8 halves straight from the
//     staged row at pixel*stride*C + 8*group -- no im2col tile;
//   * the B fragments of all steps sit behind the rows in LDS (9 KB for the YOLOv5 stem), copied once per persistent
//     workgroup and read as one conflict-free ds_read_b128 per MFMA;
//   * loads of item i+1 are issued before the MFMAs of item i and committed to LDS after them.
// One wave owns 32 consecutive output pixels of one output row and 32 output channels.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "si_hip.h"
#include "si_hip_internal.h"

typedef _Float16 half_t;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct StemArgs {
    const float* in;
    const half_t* w;    // [step][lane half][ocp][8]: the B fragments, zero beyond kw*c and beyond the last kernel row
    const float* bias;
    half_t* out;
    int n, ih, iw, c, in_ld;
    int oh, ow, oc, ocp, out_ld;
    int kh, kw, sh, sw, pt, pl;
    int row_len;        // halves staged per input row (multiple of 8)
    int n_in_rows;
    int w_tiles, oc_tiles, row_blocks, items;
    unsigned in_bytes;
    int act1, act2;
    float act_param;
    int pair_store;     // channel count and strides even, output 4-byte aligned: lanes store channel pairs as dwords
    // the split form (conv_stem_split_f32_kernel below): the lo image of the weights, the fp32 output and its size, the range-guard word, how many
    // vertical segments a (image, column tile) strip is cut into and how many row blocks a segment holds
    const half_t* w_lo;
    float* out_f32;
    unsigned out_bytes;
    unsigned* range_flag;
    int segs, seg_len;
#ifdef SI_EXPERIMENT
    int exp;   // ablations (tools/stem_bench.py): 1 no stores, 2 no MFMA loop, 4 no lo halves staged
#endif
};
#ifdef SI_EXPERIMENT
#define STEM_EXP(a, bit) ((a).exp & (bit))
#else
#define STEM_EXP(a, bit) 0
#endif

__device__ __forceinline__ float act_any(int act, float v, float p) {
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

// NPW pixel waves x NOT output-channel waves per workgroup (32*NPW output pixels along W, 32*NOT channels), KH kernel
// rows of GPR 8-value groups, RB output rows per item.  EVEN: every fragment starts on an even half index (stride*C and
// pad*C even), so it is four aligned dwords; otherwise five dwords are read and funnel-shifted per lane.  VEC: the
// 16-byte row loads described above; otherwise element-wise loads with per-element bounds (odd widths, strided input).
template <int NPW, int NOT, int KH, int GPR, int RB, bool EVEN, bool VEC>
#ifndef STEM_MINW
#define STEM_MINW 4
#endif
__global__ __launch_bounds__(NPW * NOT * 64, STEM_MINW) void conv_stem_f16_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) half_t rows_h[];
    constexpr int NTHR = NPW * NOT * 64;
    constexpr int TOW = 32 * NPW;
    constexpr int NG = KH * GPR, NS = (NG + 1) / 2;
    constexpr int MAXR = (RB - 1) * 2 + KH;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int pw = wave % NPW, cw = wave / NPW;

    f32x4 pre[MAXR];

    // blockIdx.y picks the 32*NOT-channel tile (one for the usual 32 / 64-channel stems), so the B fragments are loaded
    // once and no global load other than the row prefetch is pending inside the item loop
    const int oc0 = blockIdx.y * 32 * NOT;
    auto decode = [&](int item, int& img, int& oy0, int& ox0) {
        int t = item;
        const int wt = t % a.w_tiles; t /= a.w_tiles;
        const int rbk = t % a.row_blocks; t /= a.row_blocks;
        img = t; oy0 = rbk * RB; ox0 = wt * TOW;
    };

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const int row_floats = a.iw * a.c;
    auto prefetch = [&](int item) {
        int img, oy0, ox0;
        decode(item, img, oy0, ox0);
        const int x0 = (ox0 * a.sw - a.pl) * a.c, iy0 = oy0 * a.sh - a.pt;
        const int e = (x0 & ~3) + 4 * tid;  // float index of this thread's vector within the image row
        const bool mine = 4 * tid < a.row_len;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int y = iy0 + r;
            const bool yok = mine && r < a.n_in_rows && (unsigned)y < (unsigned)a.ih;
            if (VEC) {
                // byte offset modulo 2^32; a vector outside its row gets an offset outside the buffer -> zeros
                unsigned off = ((unsigned)((img * a.ih + y) * row_floats + e)) * 4u;
                if (!(yok && e >= 0 && e + 3 < row_floats)) off = 0xFFFFFF00u;
                pre[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ee = e + t;
                    const int px = ee / a.c, ch = ee - px * a.c;
                    unsigned off = ((unsigned)(((img * a.ih + y) * a.iw + px) * a.in_ld + ch)) * 4u;
                    if (!(yok && ee >= 0 && ee < row_floats)) off = 0xFFFFFF00u;
                    pre[r][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, off, 0, 0));
                }
            }
        }
    };
    auto commit = [&]() {
        if (4 * tid < a.row_len) {
#pragma unroll
            for (int r = 0; r < MAXR; ++r)
                if (r < a.n_in_rows) {
                    f16x4 h;
#pragma unroll
                    for (int t = 0; t < 4; ++t) h[t] = (half_t)pre[r][t];
                    *reinterpret_cast<f16x4*>(rows_h + r * a.row_len + 4 * tid) = h;
                }
        }
    };

    int item = blockIdx.x;
    if (item >= a.items) return;
    prefetch(item);

    // the B fragments of this workgroup's channels, [step][lane half][32*NOT][8], behind the rows in LDS (kept in registers
    // they cost 36 VGPRs per wave and the row prefetch no longer fits beside them at four waves per SIMD)
    half_t* const wl = rows_h + a.n_in_rows * a.row_len + 8;
    for (int i = tid; i < NS * 2 * 32 * NOT; i += NTHR) {
        const int sh2 = i / (32 * NOT), oc_l = i - sh2 * (32 * NOT);
        *reinterpret_cast<f16x8*>(wl + (size_t)i * 8) = *reinterpret_cast<const f16x8*>(a.w + ((size_t)sh2 * a.ocp + oc0 + oc_l) * 8);
    }
    const half_t* const wfrag = wl + (lh * 32 * NOT + cw * 32 + l31) * 8;
    const int o = oc0 + cw * 32 + l31;
    const float bvv = (a.bias && o < a.oc) ? a.bias[o] : 0.0f;
    commit();
    // the bias (and the weight copy) have landed before the loop: inside it the only pending loads are the next item's
    // rows, which the compiler then does not wait for until commit()
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();

    const uint32_t* rows_w = reinterpret_cast<const uint32_t*>(rows_h);
    for (; item < a.items; item += gridDim.x) {
        const int next = item + gridDim.x;
        if (next < a.items) prefetch(next);

        int img, oy0, ox0;
        decode(item, img, oy0, ox0);
        const int shift = ((ox0 * a.sw - a.pl) * a.c) & 3;  // where this item's first pixel sits in the staged window
        const int px_h = shift + (pw * 32 + l31) * a.sw * a.c;

#pragma unroll 1
        for (int rb = 0; rb < RB; ++rb) {
            const int oy = oy0 + rb;
            if (oy >= a.oh) break;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;

            const int base_h = rb * a.sh * a.row_len + px_h;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // group 2s for the low lanes, 2s+1 for the high ones; a group behind the last one has zero weights and
                // re-reads the last group's values
                const int G0 = 2 * s, G1 = (2 * s + 1 < NG) ? 2 * s + 1 : NG - 1;  // constants once unrolled
                const int off0 = (G0 / GPR) * a.row_len + 8 * (G0 % GPR);
                const int off1 = (G1 / GPR) * a.row_len + 8 * (G1 % GPR);
                const int h0 = base_h + (lh ? off1 : off0);
                u32x4 fa;
                if (EVEN) {
                    const uint32_t* p = rows_w + (h0 >> 1);
                    fa[0] = p[0]; fa[1] = p[1]; fa[2] = p[2]; fa[3] = p[3];
                } else {
                    const uint32_t* p = rows_w + (h0 >> 1);
                    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
                    const unsigned sh16 = (h0 & 1) * 16;
                    fa[0] = __builtin_amdgcn_alignbit(d1, d0, sh16);
                    fa[1] = __builtin_amdgcn_alignbit(d2, d1, sh16);
                    fa[2] = __builtin_amdgcn_alignbit(d3, d2, sh16);
                    fa[3] = __builtin_amdgcn_alignbit(d4, d3, sh16);
                }
                const f16x8 fb = *reinterpret_cast<const f16x8*>(wfrag + s * (2 * 32 * NOT * 8));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa), fb, acc, 0, 0, 0);
            }

            // ---- epilogue: C/D map col = lane&31 (channel), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel)
            const int oxb = ox0 + pw * 32 + 4 * lh;
            half_t* const prow = a.out + ((size_t)(img * a.oh + oy) * a.ow + oxb) * a.out_ld;
            // the stores are expanded once per activation case, so no activated copy of the tile is kept across a branch
            auto store_tile = [&](auto act) {
                if (a.pair_store) {
                    // one dword (two channels) per lane and store instruction, see si_pair_halves
                    const bool odd = lane & 1;
                    half_t* const pcol = prow + (o & ~1);
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        const unsigned pk = si_pair_halves(act(acc[e] + bvv), act(acc[e + 1] + bvv), odd);
                        const int dx = (e & 3) + 8 * (e >> 2) + (odd ? 1 : 0);
                        if (o < a.oc && oxb + dx < a.ow) *reinterpret_cast<unsigned*>(pcol + dx * a.out_ld) = pk;
                    }
                } else if (o < a.oc) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int dx = (e & 3) + 8 * (e >> 2);
                        if (oxb + dx < a.ow) prow[dx * a.out_ld + o] = si_store_cast<half_t>(act(acc[e] + bvv));
                    }
                }
            };
            if (a.act1 == SI_ACT_SILU && a.act2 == SI_ACT_NONE)
                store_tile([](float t) { return t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)); });
            else if (a.act1 == SI_ACT_RELU && a.act2 == SI_ACT_NONE)
                store_tile([](float t) { return fmaxf(t, 0.0f); });
            else
                store_tile([&](float t) { return act_any(a.act2, act_any(a.act1, t, a.act_param), a.act_param); });
        }

        if (next < a.items) {
            __syncthreads();  // every wave is done reading this item's rows
            commit();         // waits for the row loads -- and, the counter being shared, for this item's stores
            __syncthreads();
        }
    }
}


// ---- the split form (round 6): the stem under the engine option f32_split -------------------------------------------------------------------
// fp32 image in, FP32 activations out, every product from three fp16 MFMA products on operands split hi + 2^-11 lo (conv_split3.hip's scheme; the
// RANGE contract of include/si_hip.h: the combined accumulators are tested in front of bias and activation).  Same fragment gather and B-fragment
// layout as the kernel above -- the staged rows and the fragments exist twice (hi halves, lo halves scaled by 2^11), a step is three MFMAs into two
// accumulator sets -- but 420 MB of fp32 activations leave (the fp16 form writes half of that), so the kernel is arranged around HBM and around
// keeping every wave busy:
//   * a WAVE is the unit: it owns 32 output pixels of a row, walks DOWN its column over consecutive row blocks and keeps ITS input rows (216 halves
//     each for the YOLOv5 stem) in a private ring in LDS -- an item loads only the RB*stride input rows it does not share with the item above it (the
//     kernel above re-stages all of them: 2x the image in fetches), and nothing in the item loop is shared between waves: no barrier (the kernel
//     above has two per item, and a workgroup of five waves on four SIMDs waits for the SIMD that got two of them).  A column is cut into `segs`
//     vertical segments so that every resident wave has one (the rows above a segment's first item are the only re-reads: KH - stride rows);
//   * loads and stores are raw buffer operations whose out-of-range cases are encoded in the OFFSET (loads return zeros, stores are dropped): the
//     item loop has no branch around a memory instruction, the number of stores between the row loads and their use is a constant, and the wait in
//     front of commit() is vmcnt(<stores>) -- it does not drain the item's own stores (the kernel above waits for vmcnt(0) there);
//   * the workgroup (W waves) only shares the B fragments (hi and lo images, all NCT channel tiles of 32), copied to LDS once.
template <int W, int NCT, int KH, int GPR, int RB, bool EVEN>
__global__ __launch_bounds__(W * 64, 4) void conv_stem_split_f32_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) half_t rows_h[];
    constexpr int NTHR = W * 64;
    constexpr int NG = KH * GPR, NS = (NG + 1) / 2;
    constexpr int MAXNR = RB * 2;   // new input rows per item (RB * stride, stride <= 2)
    constexpr int WL_ELEMS = NS * 2 * 32 * NCT * 8;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int R = a.n_in_rows, NR = RB * a.sh;
    const int ring_elems = R * a.row_len + 8;   // (+ the dwords the odd-index fragments over-read)

    // the B fragments of this workgroup's channels, [hi | lo][step][lane half][32*NCT][8], behind the W rings
    half_t* const wl = rows_h + W * 2 * ring_elems;
    const int oc0 = blockIdx.y * 32 * NCT;
    for (int i = tid; i < NS * 2 * 32 * NCT; i += NTHR) {
        const int sh2 = i / (32 * NCT), oc_l = i - sh2 * (32 * NCT);
        *reinterpret_cast<f16x8*>(wl + (size_t)i * 8) = *reinterpret_cast<const f16x8*>(a.w + ((size_t)sh2 * a.ocp + oc0 + oc_l) * 8);
        *reinterpret_cast<f16x8*>(wl + WL_ELEMS + (size_t)i * 8) = *reinterpret_cast<const f16x8*>(a.w_lo + ((size_t)sh2 * a.ocp + oc0 + oc_l) * 8);
    }
    __syncthreads();   // the only barrier: from here on a wave is on its own

    // this wave's segment: column = (image, 32-pixel tile), row blocks [rbk0, rbk1)
    const int unit = blockIdx.x * W + wave;
    const int col = unit / a.segs, seg = unit - col * a.segs;
    const int img = col / a.w_tiles, wt = col - img * a.w_tiles;
    const int rbk0 = seg * a.seg_len, rbk1 = min(a.row_blocks, rbk0 + a.seg_len);
    if (img >= a.n || rbk0 >= rbk1) return;
    const int ox0 = wt * 32;

    half_t* const ring_h = rows_h + wave * 2 * ring_elems;
    half_t* const ring_l = ring_h + ring_elems;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out_f32, 0, a.out_bytes, 0x00020000);
    const int row_floats = a.iw * a.c;
    const int x0 = (ox0 * a.sw - a.pl) * a.c;
    const int e0 = (x0 & ~3) + 4 * lane;   // float index of this lane's vector within an image row
    const bool stage = 4 * lane < a.row_len;
    const bool mine = stage && e0 >= 0 && e0 + 3 < row_floats;

    f32x4 pre[MAXNR];
    // rows y0 .. y0 + cnt - 1 of the image into pre[]; a row (or a vector) outside the image gets an offset outside the buffer -> zeros
    auto load_rows = [&](int y0, int cnt) {
#pragma unroll
        for (int i = 0; i < MAXNR; ++i) {
            const int y = y0 + i;
            unsigned off = ((unsigned)((img * a.ih + y) * row_floats + e0)) * 4u;
            if (!(mine && i < cnt && (unsigned)y < (unsigned)a.ih)) off = 0xFFFFFF00u;
            pre[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
        }
    };
    // ... and from pre[] into ring slots slot0 .. (mod R), hi halves and lo halves
    auto commit = [&](int slot0, int cnt) {
#pragma unroll
        for (int i = 0; i < MAXNR; ++i) {
            int slot = slot0 + i;
            if (slot >= R) slot -= R;
            f16x4 h, l;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                h[t] = (half_t)pre[i][t];
                l[t] = (half_t)((pre[i][t] - (float)h[t]) * 2048.0f);
            }
            if (stage && i < cnt) {
                *reinterpret_cast<f16x4*>(ring_h + slot * a.row_len + 4 * lane) = h;
                if (!STEM_EXP(a, 4)) *reinterpret_cast<f16x4*>(ring_l + slot * a.row_len + 4 * lane) = l;
            }
        }
    };

    // prologue: all R rows of the segment's first item (ring position 0), in chunks of what pre[] holds
    const int iy0 = rbk0 * RB * a.sh - a.pt;
    for (int r0 = 0; r0 < R; r0 += MAXNR) {
        const int cnt = min(MAXNR, R - r0);
        load_rows(iy0 + r0, cnt);
        commit(r0, cnt);
    }

    const uint32_t* rows_w = reinterpret_cast<const uint32_t*>(ring_h);
    const uint32_t* rows_wl = reinterpret_cast<const uint32_t*>(ring_l);
    const int px_h = (x0 & 3) + l31 * a.sw * a.c;   // (x0 & 3: where the column's first pixel sits in the staged window)
    const int oxb = ox0 + 4 * lh;
    bool bad = false;   // an accumulator left the matrix cores non-finite (an operand overflowed fp16)
    // the biases of this lane's channels, loaded once: inside the item loop the only pending loads are the next item's rows
    static_assert(NCT <= 2, "a lane keeps one bias per channel tile");
    const float bv0 = (a.bias && oc0 + l31 < a.oc) ? a.bias[oc0 + l31] : 0.0f;
    const float bv1 = (NCT > 1 && a.bias && oc0 + 32 + l31 < a.oc) ? a.bias[oc0 + 32 + l31] : 0.0f;
    // (the item loop is expanded once per activation case: no branch inside it, see above)
    auto run = [&](auto act) {
        int ring0 = 0;
        for (int rbk = rbk0; rbk < rbk1; ++rbk) {
            // the rows the next item does not share with this one (its last NR): issued now, committed behind this item's MFMAs and stores
            load_rows((rbk + 1) * RB * a.sh - a.pt + (R - NR), rbk + 1 < rbk1 ? NR : 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int oy = rbk * RB + rb;
#pragma unroll 1   // (unrolled, the second tile would keep the first one's gathered fragments: 72 registers)
                for (int ct = 0; ct < NCT; ++ct) {
                    if (NCT > 1) __asm__ volatile("" ::: "memory");   // (... and hoisted out of this loop they would be kept as well)
                    const half_t* const wfrag = wl + (lh * 32 * NCT + ct * 32 + l31) * 8;
                    const int o = oc0 + ct * 32 + l31;
                    const float bvv = ct ? bv1 : bv0;
                    f32x16 acc, accx;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = accx[e] = 0.0f;
#pragma unroll
                    for (int s = 0; s < (STEM_EXP(a, 2) ? 0 : NS); ++s) {
                        const int G0 = 2 * s, G1 = (2 * s + 1 < NG) ? 2 * s + 1 : NG - 1;  // constants once unrolled
                        int slot0 = ring0 + rb * a.sh + G0 / GPR, slot1 = ring0 + rb * a.sh + G1 / GPR;
                        if (slot0 >= R) slot0 -= R;
                        if (slot1 >= R) slot1 -= R;
                        const int off0 = slot0 * a.row_len + 8 * (G0 % GPR);
                        const int off1 = slot1 * a.row_len + 8 * (G1 % GPR);
                        const int h0 = px_h + (lh ? off1 : off0);
                        auto gather = [&](const uint32_t* base) {
                            u32x4 f;
                            const uint32_t* p = base + (h0 >> 1);
                            if (EVEN) {
                                f[0] = p[0]; f[1] = p[1]; f[2] = p[2]; f[3] = p[3];
                            } else {
                                const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
                                const unsigned sh16 = (h0 & 1) * 16;
                                f[0] = __builtin_amdgcn_alignbit(d1, d0, sh16);
                                f[1] = __builtin_amdgcn_alignbit(d2, d1, sh16);
                                f[2] = __builtin_amdgcn_alignbit(d3, d2, sh16);
                                f[3] = __builtin_amdgcn_alignbit(d4, d3, sh16);
                            }
                            return __builtin_bit_cast(f16x8, f);
                        };
                        const f16x8 fa = gather(rows_w), fl = gather(rows_wl);
                        const f16x8 fb = *reinterpret_cast<const f16x8*>(wfrag + s * (2 * 32 * NCT * 8));
                        const f16x8 fbl = *reinterpret_cast<const f16x8*>(wfrag + WL_ELEMS + s * (2 * 32 * NCT * 8));
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
                        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fbl, accx, 0, 0, 0);
                        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, fb, accx, 0, 0, 0);
                    }
                    // the two scales meet; the range guard reads the combined accumulators before bias / activation can hide an Inf or a NaN
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        acc[e] = acc[e] + accx[e] * (1.0f / 2048.0f);
                        bad |= !__builtin_isfinite(acc[e]);
                    }
                    // ---- epilogue: C/D map col = lane&31 (channel), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel); a lane is a channel, 32 lanes
                    // write 128 consecutive bytes of a pixel
                    const unsigned obase = ((unsigned)((img * a.oh + oy) * a.ow + oxb) * (unsigned)a.out_ld + (unsigned)o) * 4u;
                    const int ow_live = (o < a.oc && oy < a.oh && !STEM_EXP(a, 1)) ? a.ow : 0;   // (one compare per store: dead lanes / rows see an empty row)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int dx = (e & 3) + 8 * (e >> 2);
                        // the pixel's distance from the tile's first one travels in the scalar offset (not part of the range check: a dropped store stays dropped)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, act(acc[e] + bvv)), rs_out, oxb + dx < ow_live ? obase : 0xFFFFFF00u,
                                                              dx * a.out_ld * 4, 0);
                    }
                }
            }
            commit(ring0, NR);   // the next item's new rows over this item's oldest ones (LDS is in order within a wave; waits for the row loads only)
            ring0 += NR;
            if (ring0 >= R) ring0 -= R;
        }
    };
    if (a.act1 == SI_ACT_SILU && a.act2 == SI_ACT_NONE)
        run([](float t) { return t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)); });
    else if (a.act1 == SI_ACT_RELU && a.act2 == SI_ACT_NONE)
        run([](float t) { return fmaxf(t, 0.0f); });
    else if (a.act1 == SI_ACT_NONE && a.act2 == SI_ACT_NONE)
        run([](float t) { return t; });
    else
        run([&](float t) { return act_any(a.act2, act_any(a.act1, t, a.act_param), a.act_param); });
    if (a.range_flag && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) *reinterpret_cast<volatile unsigned*>(a.range_flag) = 1u;
}

inline int groups_per_row(const SiConv2dDesc* d) { return (d->kw * d->ic + 7) / 8; }
inline int steps_of(const SiConv2dDesc* d) { return (d->kh * groups_per_row(d) + 1) / 2; }
// whole channel tiles of the variant that will run: 32 channels per workgroup up to 32, 64 beyond
inline int padded_oc(const SiConv2dDesc* d) { return d->oc <= 32 ? 32 : (d->oc + 63) / 64 * 64; }

template <int NPW, int NOT, int KH, int GPR, int RB, bool EVEN>
int launch_stem(StemArgs a, hipStream_t s) {
    constexpr int TOW = 32 * NPW, NTHR = NPW * NOT * 64;
    a.w_tiles = (a.ow + TOW - 1) / TOW;
    a.oc_tiles = (a.oc + 32 * NOT - 1) / (32 * NOT);
    a.row_blocks = (a.oh + RB - 1) / RB;
    a.n_in_rows = (RB - 1) * a.sh + a.kh;
    a.row_len = (3 + (TOW - 1) * a.sw * a.c + 8 * GPR + 7) / 8 * 8;  // multiple of 8 halves: rows and weights stay 16-byte aligned
    if (a.row_len > 4 * NTHR || a.n_in_rows > (RB - 1) * 2 + KH) return SI_E_UNSUPPORTED;
    const long long items = (long long)a.n * a.row_blocks * a.w_tiles;
    if (items > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    a.items = (int)items;
    // rows (+ the dwords the odd-index fragments over-read, keeping the weights 16-byte aligned) + B fragments
    const size_t lds = ((size_t)a.n_in_rows * a.row_len + 8 + (size_t)((KH * GPR + 1) / 2) * 2 * 32 * NOT * 8) * sizeof(half_t);
    const bool vec = a.in_ld == a.c && (a.iw * a.c) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0;
    // persistent grid: exactly the workgroups that are resident at once (a workgroup that has to wait for a slot would
    // start its share of the items when the others are finishing theirs)
    auto launch = [&](auto kern) {
        const int per_cu = si_resident_blocks(kern, NTHR, lds);
        int grid = 256 * per_cu / a.oc_tiles;
        if (grid < 1) grid = 1;
        if ((long long)grid > items) grid = (int)items;
        hipLaunchKernelGGL(kern, dim3(grid, a.oc_tiles), dim3(NTHR), lds, s, a);
    };
    if (vec)
        launch(conv_stem_f16_kernel<NPW, NOT, KH, GPR, RB, EVEN, true>);
    else
        launch(conv_stem_f16_kernel<NPW, NOT, KH, GPR, RB, EVEN, false>);
    return (int)hipGetLastError();
}

template <int W, int NCT, int KH, int GPR, int RB, bool EVEN>
int launch_stem_split(StemArgs a, hipStream_t s) {
    a.w_tiles = (a.ow + 31) / 32;
    a.oc_tiles = (a.oc + 32 * NCT - 1) / (32 * NCT);
    a.row_blocks = (a.oh + RB - 1) / RB;
    a.n_in_rows = (RB - 1) * a.sh + a.kh;
    a.row_len = (3 + 31 * a.sw * a.c + 8 * GPR + 7) / 8 * 8;   // halves a wave stages per input row
    if (a.row_len > 4 * 64 || a.sh > 2 || a.kh < a.sh) return SI_E_UNSUPPORTED;
    // dense image rows on 16-byte boundaries (what Engine::Input hands over), offsets that fit the buffer descriptors
    if (!(a.in_ld == a.c && (a.iw * a.c) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0)) return SI_E_UNSUPPORTED;
    const unsigned long long out_bytes = ((unsigned long long)a.n * a.oh * a.ow - 1) * a.out_ld * 4ull + (unsigned long long)a.oc * 4ull;
    if (out_bytes > 0xFFFFFF00ull || (reinterpret_cast<uintptr_t>(a.out_f32) & 3) != 0) return SI_E_UNSUPPORTED;
    a.out_bytes = (unsigned)out_bytes;
    const long long cols = (long long)a.n * a.w_tiles;
    const size_t lds = (W * 2 * ((size_t)a.n_in_rows * a.row_len + 8) + 2 * (size_t)((KH * GPR + 1) / 2) * 2 * 32 * NCT * 8) * sizeof(half_t);
    auto kern = conv_stem_split_f32_kernel<W, NCT, KH, GPR, RB, EVEN>;
    if (si_allow_dynamic_lds(kern, lds) != hipSuccess) return SI_E_UNSUPPORTED;
    // one segment per resident wave (a workgroup that had to wait for a slot would start when the others finish); a segment no shorter than 4 row
    // blocks (the rows above its first item are re-reads)
    const long long resident = (long long)256 * si_resident_blocks(kern, W * 64, lds) * W / a.oc_tiles;
    long long segs = std::max(1LL, std::min(resident / cols, (long long)(a.row_blocks + 3) / 4));
    a.seg_len = (int)((a.row_blocks + segs - 1) / segs);
    a.segs = (a.row_blocks + a.seg_len - 1) / a.seg_len;
    const long long blocks = (cols * a.segs + W - 1) / W;
    if (blocks > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, a.oc_tiles), dim3(W * 64), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

bool si_conv_smallc_ok(const SiConv2dDesc* d);  // conv_smallc.hip: 1..3 channels, stride <= 2, no groups / dilation

// shape-only: the kernel-row layouts instantiated below
bool si_conv_stem_f16_ok(const SiConv2dDesc* d) {
    if (!si_conv_smallc_ok(d)) return false;
    const int gpr = groups_per_row(d);
    return (d->kh == 6 && gpr == 3) || (d->kh == 7 && gpr == 3) || (d->kh == 3 && gpr == 2);
}

extern "C" {

size_t si_hip_conv2d_stem_f16_weight_elems(const SiConv2dDesc* d) {
    if (!d || !si_conv_stem_f16_ok(d)) return 0;
    return (size_t)steps_of(d) * 2 * padded_oc(d) * 8;
}

// OIHW fp32 -> [step][lane half][ocp][8] fp16 (round to nearest even), zero filled
int si_hip_conv2d_stem_f16_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, void* w_packed) {
    if (!d || !w_oihw || !w_packed) return SI_E_BADARG;
    if (!si_conv_stem_f16_ok(d)) return SI_E_UNSUPPORTED;
    const int gpr = groups_per_row(d), ng = d->kh * gpr, ns = steps_of(d), ocp = padded_oc(d);
    half_t* w = static_cast<half_t*>(w_packed);
    for (int s = 0; s < ns; ++s)
        for (int h = 0; h < 2; ++h) {
            const int g = 2 * s + h;
            for (int o = 0; o < ocp; ++o)
                for (int t = 0; t < 8; ++t) {
                    float v = 0.0f;
                    const int j = 8 * (g % gpr) + t;
                    if (g < ng && o < d->oc && j < d->kw * d->ic) {
                        const int ky = g / gpr, kx = j / d->ic, ch = j % d->ic;
                        v = w_oihw[(((size_t)o * d->ic + ch) * d->kh + ky) * d->kw + kx];
                    }
                    w[(((size_t)s * 2 + h) * ocp + o) * 8 + t] = (half_t)v;
                }
        }
    return 0;
}

// the stem on the f32_split arithmetic: two images of the fragments above, hi then lo (scaled by 2^11); SI_E_UNSUPPORTED for a weight fp16 cannot hold
size_t si_hip_conv2d_stem_split3_weight_elems(const SiConv2dDesc* d) { return 2 * si_hip_conv2d_stem_f16_weight_elems(d); }

int si_hip_conv2d_stem_split3_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, void* w_packed) {
    if (!d || !w_oihw || !w_packed) return SI_E_BADARG;
    if (!si_conv_stem_f16_ok(d)) return SI_E_UNSUPPORTED;
    const int gpr = groups_per_row(d), ng = d->kh * gpr, ns = steps_of(d), ocp = padded_oc(d);
    half_t* const hi = static_cast<half_t*>(w_packed);
    half_t* const lo = hi + (size_t)ns * 2 * ocp * 8;
    for (int s = 0; s < ns; ++s)
        for (int h = 0; h < 2; ++h) {
            const int g = 2 * s + h;
            for (int o = 0; o < ocp; ++o)
                for (int t = 0; t < 8; ++t) {
                    float v = 0.0f;
                    const int j = 8 * (g % gpr) + t;
                    if (g < ng && o < d->oc && j < d->kw * d->ic) {
                        const int ky = g / gpr, kx = j / d->ic, ch = j % d->ic;
                        v = w_oihw[(((size_t)o * d->ic + ch) * d->kh + ky) * d->kw + kx];
                    }
                    const half_t hv = (half_t)v;
                    if (!(__builtin_fabsf((float)hv) <= 65504.0f)) return SI_E_UNSUPPORTED;
                    const size_t idx = (((size_t)s * 2 + h) * ocp + o) * 8 + t;
                    hi[idx] = hv;
                    lo[idx] = (half_t)((v - (float)hv) * 2048.0f);
                }
        }
    return 0;
}

static int stem_launch(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, void* out, si_stream_t stream, bool split);

int si_hip_conv2d_stem_split3_f32(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, float* out, si_stream_t stream) {
    return stem_launch(d, in, w_packed, bias, out, stream, true);
}

int si_hip_conv2d_stem_f16(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, void* out,
                           si_stream_t stream) {
    return stem_launch(d, in, w_packed, bias, out, stream, false);
}

static int stem_launch(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, void* out, si_stream_t stream, bool split) {
    if (!d || !in || !w_packed || !out) return SI_E_BADARG;
    if (!si_conv_stem_f16_ok(d) || d->has_residual) return SI_E_UNSUPPORTED;
    StemArgs a;
    a.w_lo = static_cast<const half_t*>(w_packed) + (split ? si_hip_conv2d_stem_f16_weight_elems(d) : 0);
    a.out_f32 = static_cast<float*>(out);
    a.out_bytes = 0;
    a.range_flag = split ? d->range_flag : nullptr;
    a.segs = a.seg_len = 0;
#ifdef SI_EXPERIMENT
    a.exp = SI_ENV_INT("SI_STEM_EXP", 0);
#endif
    a.in = in; a.w = static_cast<const half_t*>(w_packed); a.bias = d->has_bias ? bias : nullptr; a.out = static_cast<half_t*>(out);
    a.n = d->n; a.ih = d->ih; a.iw = d->iw; a.c = d->ic; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.oc = d->oc; a.ocp = padded_oc(d); a.out_ld = d->out_ld;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.pl = d->pl;
    a.row_len = a.n_in_rows = a.w_tiles = a.oc_tiles = a.row_blocks = a.items = 0;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    a.pair_store = (d->oc % 2 == 0 && d->out_ld % 2 == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) ? 1 : 0;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->ih * d->iw * d->in_ld * 4ull;
    if (in_bytes >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;  // > 4 GB input image tensor
    a.in_bytes = (unsigned)in_bytes;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool even = (d->sw * d->ic) % 2 == 0 && (d->pl * d->ic) % 2 == 0;
    const bool wide = d->oc > 32;
    const int gpr = groups_per_row(d);
    if (split) {
        // (the forms the three reference stems take, on the f32_split arithmetic.  More than 32 channels: one 32-channel tile per workgroup row
        // (blockIdx.y), each staging the image rows for itself -- a wave computing two tiles from the rows it staged (NCT = 2) needs more registers
        // than four waves per SIMD leave once the fragments start on odd half indices, and a 64-channel stem's image is small)
        if (d->kh == 6 && gpr == 3) return even ? launch_stem_split<8, 1, 6, 3, 2, true>(a, s) : launch_stem_split<8, 1, 6, 3, 2, false>(a, s);
        if (d->kh == 7 && gpr == 3) return launch_stem_split<8, 1, 7, 3, 2, false>(a, s);
        if (d->kh == 3 && gpr == 2) return launch_stem_split<8, 1, 3, 2, 2, false>(a, s);
        return SI_E_UNSUPPORTED;
    }
    if (d->kh == 6 && gpr == 3) {  // 6x6x3 (YOLOv5)
        if (wide) return launch_stem<4, 2, 6, 3, 2, false>(a, s);
        if (!even) return launch_stem<4, 1, 6, 3, 2, false>(a, s);
        if (d->ow % 160 == 0 || d->ow > 128) return launch_stem<5, 1, 6, 3, 2, true>(a, s);
        return launch_stem<4, 1, 6, 3, 2, true>(a, s);
    }
    if (d->kh == 7 && gpr == 3) {  // 7x7x3 (ResNet)
        if (wide) return launch_stem<4, 2, 7, 3, 2, false>(a, s);
        return launch_stem<4, 1, 7, 3, 2, false>(a, s);
    }
    if (d->kh == 3 && gpr == 2) {  // 3x3x3 (MobileNet)
        if (wide) return launch_stem<4, 2, 3, 2, 2, false>(a, s);
        return launch_stem<4, 1, 3, 2, 2, false>(a, s);
    }
    return SI_E_UNSUPPORTED;
}

}  // extern "C"
