// conv_stem_s2c32_f16.hip -- YOLOv5's first two convolutions in ONE persistent kernel (fp16 storage path, round 4).
//
// conv_0 (6x6 s2 p2, 3 -> 32, SiLU; reference src/layer/conv_2d.cpp:207-283 on the 3-channel image) writes 210 MB of fp16
// activations at batch 32 that conv_1 (3x3 s2 p1, 32 -> 64, SiLU) reads straight back: 118 + 70 us for 157 MB in and 105 MB out.
// Here a workgroup owns 4 x 16 pixels of conv_1's OUTPUT per item and never lets the intermediate leave the CU:
//   stage 0  the 22 x 70-pixel window of the fp32 image under the tile is requested one item ahead (global -> registers) and
//            committed to LDS as fp16 -- conv_stem_f16.hip's row staging;
//   phase A  the 9 x 33 stem pixels conv_1's tile needs are computed with conv_stem_f16.hip's contraction (kernel rows cut into
//            groups of 8 values, two groups per v_mfma_f32_32x32x16_f16, B fragments in LDS; ten 32-pixel blocks: one per patch
//            row plus one for the 33rd column), bias + SiLU, rounded to fp16 and written into conv_s2c32_f16_kernel's patch image
//            (odd / even column planes, 80-byte pixels); stem pixels outside the stem's output are conv_1's zero padding;
//   phase B  conv_s2c32_f16_kernel's nine taps from the patch, weights resident in registers, bias + SiLU, fp16 out.
// The stem is recomputed on the one-pixel seam between tiles (9 x 33 for 8 x 32: 1.16x).  Same MFMA steps in the same order and
// the same epilogue expressions as the two kernels it replaces, the same fp16 rounding of the intermediate: the same bits
// (tests/test_gpu_f16.py::test_stem_s2c32_fused_same_bits).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "si_hip.h"
#include "si_hip_internal.h"

#pragma clang fp contract(off)

typedef _Float16 half_t;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct FusedArgs {
    const float* in;      // [n][ih][iw][3] fp32
    const half_t* ws;     // stem B fragments [9 steps][lane half][32][8] (si_hip_conv2d_stem_f16_pack_weight_host)
    const float* bs;      // stem bias or null
    const half_t* wl;     // conv_1 weights in MFMA lane order: [oc / 32][18 k-steps][64 lanes][8]
    const float* bc;      // conv_1 bias or null
    half_t* out;
    int ih, iw, soh, sow; // image, stem output
    int oh, ow, oc, out_ld, nb;
    int tiles_x, tiles_y, items;
    unsigned in_bytes;
    // PW form (round 6): a 1x1 conv over conv_1's 64 channels behind it -- YOLOv5's first C3's cv1 | cv2 as ONE 64 -> 64 conv with a split
    // destination -- computed from the tile while it is still in the CU: conv_1's output is never written
    const half_t* wp;     // its weights in MFMA lane order: [2 column blocks][4 k-steps][64 lanes][8]
    const float* bp;      // its bias (64) or null
    half_t* out_b;        // destination of output channels [32, 64) (`out` takes [0, 32)); strides in elements
    int out_b_ld;
    unsigned out_bytes, out_b_bytes;   // extents of the two destinations (buffer stores: an out-of-range pixel is an out-of-range OFFSET)
};

// SI_FUSED_ABL (diagnostic builds only, tools/stem_fused_ablate.sh): bit 0 no SiLU (bias add only), 1 no stem MFMAs, 2 no phase B,
// 3 no global stores, 4 no image prefetch after the first item -- wrong results, timing only.  0 in the product build.
#ifndef SI_FUSED_ABL
#define SI_FUSED_ABL 0
#endif
__device__ __forceinline__ float silu_f(float t) { return t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)); }
// SiLU of (x0 + b, x1 + b) on the packed fp32 instructions (conv_igemm.hip epilogue_lean: each component is rounded exactly like the
// scalar form -- __expf(-t) is v_exp_f32(t * -log2(e)) -- so the bits are silu_f's), then fp16 bit patterns
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void silu2_bits(float x0, float x1, float b, unsigned& h0, unsigned& h1) {
    const f32x2 v = f32x2{x0, x1} + f32x2{b, b};
    const f32x2 x = v * f32x2{-1.44269504088896340736f, -1.44269504088896340736f};
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])} + f32x2{1.0f, 1.0f};
    const f32x2 o = (SI_FUSED_ABL & 1) ? v : v * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    h0 = __builtin_bit_cast(unsigned short, si_store_cast<half_t>(o[0]));
    h1 = __builtin_bit_cast(unsigned short, si_store_cast<half_t>(o[1]));
}

constexpr unsigned OOB = 0xFFFFFF00u;
constexpr int PITCH = 80, ROWP = 2688, EOFF = 17 * PITCH;     // patch image, bytes (conv_s2c32_f16_kernel)
constexpr int PR = 9;
constexpr int RL = 216;                                       // halves per staged image row: 70 pixels x 3 + the last group's over-read
constexpr int IR = 22;                                        // image rows under a tile
constexpr int VPRW = RL / 4;                                  // 16-byte fp32 vectors per staged row (54)
constexpr int NVEC = IR * VPRW, N_IT = (NVEC + 255) / 256;
constexpr int PATCH_BYTES = PR * ROWP, STAGE_BYTES = IR * RL * 2, WS_BYTES = 9 * 2 * 32 * 8 * 2;

#ifndef SI_FUSED_MINW
#define SI_FUSED_MINW 2
#endif
constexpr int TP = 72;   // PW form: halves per pixel of the tile image [64 pixels][64 channels + 8] (144-byte pitch: conflict-free ds_read_b128)
// PW: phase C behind phase B -- the tile (4 x 16 pixels x 64 channels, SiLU'd and rounded to fp16 exactly as it would have been stored) goes
// to LDS as [pixel][channel]; after one barrier every wave multiplies ITS 32 pixels by ITS 32 output columns of the 1x1 conv (four 16-deep MFMA
// steps over the 64 channels, ascending: the generic fp16 tiles' k order, so the bits are si_hip_conv2d_split_f16's), bias + SiLU, and writes
// output channels [0, 32) to `out`, [32, 64) to `out_b` (the C3's cv1 result and cv2's slice of the concat buffer).
// BST (round 6, the form the product launches): the output stores are raw buffer stores whose out-of-range case is an out-of-range offset -- no
// branch around a store, so the compiler can COUNT the stores between the next window's loads and commit(): the wait there is vmcnt(<stores> + n)
// instead of vmcnt(n), which drained the item's own stores before the next window could be staged (conv_stem_f16.hip's split form, LAB_NOTEBOOK
// R6.10).  Same values to the same addresses.  BST = false (the pointer stores under `if`) exists in experiment builds only, for the A/B.
template <bool PW, bool BST = true>
__global__ __launch_bounds__(256, SI_FUSED_MINW) void conv_stem_s2c32_f16_kernel(const FusedArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char patch[PATCH_BYTES];
    __shared__ __attribute__((aligned(16))) half_t stage[IR * RL];
    __shared__ __attribute__((aligned(16))) half_t wsl[9 * 2 * 32 * 8];
    __shared__ __attribute__((aligned(16))) half_t tile[PW ? 64 * TP : 8];

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const int row_floats = a.iw * 3;

    // staging vectors of this thread: vector v of staged row r is image floats [192 tx - 12 + 4 v, + 4) of image row 16 ty - 4 + r
    int v_r[N_IT], v_v[N_IT];
#pragma unroll
    for (int i = 0; i < N_IT; ++i) {
        const int c = tid + 256 * i;
        v_r[i] = c < NVEC ? c / VPRW : -1;
        v_v[i] = c - (c / VPRW) * VPRW;
    }
    f32x4 pre[N_IT];
    auto prefetch = [&](int item) {
        const int tx = item % a.tiles_x, t2 = item / a.tiles_x;
        const int ty = t2 % a.tiles_y, img = t2 / a.tiles_y;
        const int iy0 = 16 * ty - 4, e0 = 192 * tx - 12;
#pragma unroll
        for (int i = 0; i < N_IT; ++i) {
            const int y = iy0 + v_r[i], e = e0 + 4 * v_v[i];
            const bool ok = v_r[i] >= 0 && (unsigned)y < (unsigned)a.ih && e >= 0 && e + 3 < row_floats;
            const unsigned off = ((unsigned)((img * a.ih + y) * row_floats + e)) * 4u;
            pre[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off : OOB, 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < N_IT; ++i)
            if (v_r[i] >= 0) {
                f16x4 h;
#pragma unroll
                for (int t = 0; t < 4; ++t) h[t] = (half_t)pre[i][t];
                *reinterpret_cast<f16x4*>(stage + v_r[i] * RL + 4 * v_v[i]) = h;
            }
    };

    int item = blockIdx.x;
    if (item >= a.items) return;
    prefetch(item);
    for (int i = tid; i < 9 * 2 * 32; i += 256) *reinterpret_cast<f16x8*>(wsl + i * 8) = *reinterpret_cast<const f16x8*>(a.ws + (size_t)i * 8);
    // conv_1: this wave's column block of the weights, 18 k-steps (tap-major, two 16-channel halves per tap), resident in registers
    const int wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.wl), 0, (unsigned)a.nb * 18u * 1024u, 0x00020000);
    f16x8 wf[18];
    {
        const unsigned vo = wn < a.nb ? (unsigned)wn * 18u * 1024u + (unsigned)lane * 16u : 0x80000000u;
#pragma unroll
        for (int s = 0; s < 18; ++s) wf[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wl, vo, (unsigned)(s * 1024), 0));
    }
    const float bsv = a.bs ? a.bs[l31] : 0.0f;
    const int o = wn * 32 + l31;
    const float bcv = (a.bc && o < a.oc) ? a.bc[o] : 0.0f;
    // PW: this wave's column block of the 1x1 conv's weights (four k-steps), resident as well, and its bias
    f16x8 wpf[PW ? 4 : 1];
    float bpv = 0.0f;
    if (PW) {
        const __amdgpu_buffer_rsrc_t rs_wp = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.wp), 0, 2u * 4u * 1024u, 0x00020000);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            wpf[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wp, (unsigned)wn * 4096u + (unsigned)lane * 16u, (unsigned)(s * 1024), 0));
        bpv = a.bp ? a.bp[o] : 0.0f;
    }
    // the stem's B fragments of this lane, resident too (SI_FUSED_WS_REGS=0: re-read from LDS per MFMA, as conv_stem_f16.hip does)
#ifndef SI_FUSED_WS_REGS
#define SI_FUSED_WS_REGS 1
#endif
    const half_t* const wfrag = wsl + (lh * 32 + l31) * 8;
    f16x8 wsf[9];
    if (SI_FUSED_WS_REGS) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 9; ++s) wsf[s] = *reinterpret_cast<const f16x8*>(wfrag + s * (2 * 32 * 8));
    }
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out_b = __builtin_amdgcn_make_buffer_rsrc(PW ? a.out_b : a.out, 0, PW ? a.out_b_bytes : a.out_bytes, 0x00020000);
    const uint32_t* const stage_w = reinterpret_cast<const uint32_t*>(stage);
    const int a_base = (2 * (2 * wm + (l31 >> 4))) * ROWP + (l31 & 15) * PITCH + lh * 16;
    // conv_1's weights and the biases have LANDED before the loop: otherwise every MFMA of phase B carries a vmcnt(N) wait for "its"
    // weight register, N counting down to 0, i.e. for the next item's window loads and this item's output stores as well
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

    for (; item < a.items; item += gridDim.x) {
        const int tx = item % a.tiles_x, t2 = item / a.tiles_x;
        const int ty = t2 % a.tiles_y, img = t2 / a.tiles_y;
        commit();          // (waits for this item's window; the stores of the previous item are younger than those loads)
        __syncthreads();   // the window is staged; every wave is done with the previous item's patch
        const int next = item + gridDim.x;
        if (next < a.items && !(SI_FUSED_ABL & 16)) prefetch(next);

        // ---- phase A: stem pixels of the patch.  Block b < 9: patch row b, columns 0..31; block 9: column 32 of rows 0..8.
        // Everything about an element's place is either uniform (the row, its validity) or a per-lane base plus a compile-time
        // offset (column r = (e & 3) + 8 (e >> 2) + 4 lh: its plane is the parity of the compile-time part).
        const int sx_l = 32 * tx - 1 + 4 * lh;                               // stem column of this lane's row r = 4 lh
        unsigned char* const prow_l = patch + (2 * lh) * PITCH + l31 * 2;     // ... and its place in a patch row
        for (int b = wave; b < 10; b += 4) {
            const int pi = b < 9 ? b : min(l31, 8), pj = b < 9 ? l31 : 32;
            const int base_h = (2 * pi) * RL + 6 * pj;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            // all nine A fragments of the block are requested before its first MFMA (left interleaved, hipcc keeps ONE step of
            // reads ahead of a chain of dependent MFMAs: an LDS round trip per step, ~1200 cycles per block for 288 of matrix work)
            u32x4 fa[9];
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                // group 2s for the low lanes, 2s + 1 for the high ones (conv_stem_f16.hip): kernel row G / 3, values 8 (G % 3) ..
                const int G0 = 2 * s, G1 = 2 * s + 1;
                const int off0 = (G0 / 3) * RL + 8 * (G0 % 3), off1 = (G1 / 3) * RL + 8 * (G1 % 3);
                const int h0 = base_h + (lh ? off1 : off0);
                const uint32_t* p = stage_w + (h0 >> 1);
                fa[s][0] = p[0]; fa[s][1] = p[1]; fa[s][2] = p[2]; fa[s][3] = p[3];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < ((SI_FUSED_ABL & 2) ? 0 : 9); ++s) {
                const f16x8 fb = SI_FUSED_WS_REGS ? wsf[s] : *reinterpret_cast<const f16x8*>(wfrag + s * (2 * 32 * 8));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[s]), fb, acc, 0, 0, 0);
            }
            // C/D map: col = lane & 31 (stem channel), row = (e & 3) + 8 (e >> 2) + 4 lh (block pixel)
            if (b < 9) {
                const bool row_ok = (unsigned)(8 * ty - 1 + b) < (unsigned)a.soh;   // uniform
                unsigned char* const pr = prow_l + b * ROWP;
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int c = (e & 3) + 8 * (e >> 2);                        // compile-time part of the column (c, c + 1)
                    unsigned h0, h1;
                    silu2_bits(acc[e], acc[e + 1], bsv, h0, h1);
                    // (masks, not branches: a pixel outside the stem's output is conv_1's zero padding)
                    h0 &= (row_ok && (unsigned)(sx_l + c) < (unsigned)a.sow) ? 0xFFFFu : 0u;
                    h1 &= (row_ok && (unsigned)(sx_l + c + 1) < (unsigned)a.sow) ? 0xFFFFu : 0u;
                    *reinterpret_cast<unsigned short*>(pr + ((c & 1) ? EOFF : 0) + (c >> 1) * PITCH) = (unsigned short)h0;
                    *reinterpret_cast<unsigned short*>(pr + (((c + 1) & 1) ? EOFF : 0) + ((c + 1) >> 1) * PITCH) = (unsigned short)h1;
                }
            } else {
                const bool col_ok = (unsigned)(32 * tx + 31) < (unsigned)a.sow;  // uniform
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                    unsigned h0, h1;
                    silu2_bits(acc[e], acc[e + 1], bsv, h0, h1);
                    h0 &= (col_ok && (unsigned)(8 * ty - 1 + r) < (unsigned)a.soh) ? 0xFFFFu : 0u;
                    h1 &= (col_ok && (unsigned)(8 * ty + r) < (unsigned)a.soh) ? 0xFFFFu : 0u;
                    if (r < PR) *reinterpret_cast<unsigned short*>(patch + r * ROWP + 16 * PITCH + l31 * 2) = (unsigned short)h0;
                    if (r + 1 < PR) *reinterpret_cast<unsigned short*>(patch + (r + 1) * ROWP + 16 * PITCH + l31 * 2) = (unsigned short)h1;
                }
            }
        }
        __syncthreads();

        // ---- phase B: conv_1 from the patch (conv_s2c32_f16_kernel)
        if (!(SI_FUSED_ABL & 4)) {
            const unsigned char* const P = patch + a_base;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            // A fragments in groups of three, the next group requested before this one multiplies (see phase A)
            auto frag = [&](int s) {
                const int tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky;
                const int off = ky * ROWP + (kx == 1 ? EOFF : (kx == 2 ? PITCH : 0)) + (s & 1) * 32;
                return *reinterpret_cast<const f16x8*>(P + off);
            };
            f16x8 fb[2][3];
#pragma unroll
            for (int i = 0; i < 3; ++i) fb[0][i] = frag(i);
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                if (g + 1 < 6) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) fb[(g + 1) & 1][i] = frag(3 * (g + 1) + i);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 3; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[g & 1][i], wf[3 * g + i], acc, 0, 0, 0);
            }
            const int oy0 = ty * 4 + 2 * wm, ox0 = tx * 16;
            if constexpr (PW) {
                // the tile into LDS as [pixel 32 wm + r][channel o] (conv_1's own epilogue expressions and rounding: silu2_bits)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                    unsigned h0, h1;
                    silu2_bits(acc[e], acc[e + 1], bcv, h0, h1);
                    *reinterpret_cast<unsigned short*>(tile + (32 * wm + r) * TP + o) = (unsigned short)h0;
                    *reinterpret_cast<unsigned short*>(tile + (32 * wm + r + 1) * TP + o) = (unsigned short)h1;
                }
                __syncthreads();   // both column-block waves of a pixel block have written their channels
                f32x16 acc2;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[e] = 0.0f;
                const half_t* const T = tile + (32 * wm + l31) * TP + 8 * lh;
                f16x8 ta[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) ta[s] = *reinterpret_cast<const f16x8*>(T + 16 * s);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta[s], wpf[s], acc2, 0, 0, 0);
                half_t* const ob = wn == 0 ? a.out + l31 : a.out_b + l31;
                const int old = wn == 0 ? a.out_ld : a.out_b_ld;
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
                    unsigned h0, h1;
                    silu2_bits(acc2[e], acc2[e + 1], bpv, h0, h1);
                    if constexpr (BST) {
                        const int ow_live = oy < a.oh ? a.ow : 0;   // (uniform: r >> 4 is a compile-time 0 / 1)
                        const unsigned off = ((unsigned)((img * a.oh + oy) * a.ow + ox) * (unsigned)old + (unsigned)l31) * 2u;
                        const __amdgpu_buffer_rsrc_t rs = wn == 0 ? rs_out : rs_out_b;
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)h0, rs, ox < ow_live ? off : OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)h1, rs, ox + 1 < ow_live ? off : OOB, old * 2, 0);
                    } else {
                    unsigned short* const op = reinterpret_cast<unsigned short*>(ob + (size_t)((img * a.oh + oy) * a.ow + ox) * old);
                    if (oy < a.oh && ox < a.ow) op[0] = (unsigned short)h0;
                    if (oy < a.oh && ox + 1 < a.ow) op[old] = (unsigned short)h1;
                    }
                }
                continue;   // (the next item's commit() writes `stage`, its barrier orders this item's tile reads before the next tile writes)
            }
            half_t* const ob = a.out + o;
            if constexpr (BST) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;   // rows r, r + 1: the same output row (r & 15 is even)
                    const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
                    unsigned h0, h1;
                    silu2_bits(acc[e], acc[e + 1], bcv, h0, h1);
                    const int ow_live = (oy < a.oh && o < a.oc && !(SI_FUSED_ABL & 8)) ? a.ow : 0;
                    const unsigned off = ((unsigned)((img * a.oh + oy) * a.ow + ox) * (unsigned)a.out_ld + (unsigned)o) * 2u;
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)h0, rs_out, ox < ow_live ? off : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)h1, rs_out, ox + 1 < ow_live ? off : OOB, a.out_ld * 2, 0);
                }
            } else if (o < a.oc) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;   // rows r, r + 1: the same output row (r & 15 is even)
                    const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
                    unsigned h0, h1;
                    silu2_bits(acc[e], acc[e + 1], bcv, h0, h1);
                    unsigned short* const op = reinterpret_cast<unsigned short*>(ob + (size_t)((img * a.oh + oy) * a.ow + ox) * a.out_ld);
                    if (SI_FUSED_ABL & 8) { asm volatile("" ::"v"(h0), "v"(h1)); continue; }
                    if (oy < a.oh && ox < a.ow) op[0] = (unsigned short)h0;
                    if (oy < a.oh && ox + 1 < a.ow) op[a.out_ld] = (unsigned short)h1;
                }
            }
        }
    }
}

bool stem_ok(const SiConv2dDesc* d) {
    return d->groups == 1 && d->ic == 3 && d->in_ld == 3 && d->oc == 32 && d->kh == 6 && d->kw == 6 && d->sh == 2 && d->sw == 2 &&
           d->dh == 1 && d->dw == 1 && d->pt == 2 && d->pl == 2 && !d->has_residual && d->act1 == SI_ACT_SILU && d->act2 == SI_ACT_NONE &&
           d->iw % 4 == 0 && d->oh == (d->ih + 4 - 6) / 2 + 1 && d->ow == (d->iw + 4 - 6) / 2 + 1 && d->oh > 0 && d->ow > 0;
}
bool conv_ok(const SiConv2dDesc* d) {
    return d->groups == 1 && d->ic == 32 && (d->oc == 32 || d->oc == 64) && d->kh == 3 && d->kw == 3 && d->sh == 2 && d->sw == 2 &&
           d->dh == 1 && d->dw == 1 && d->pt == 1 && d->pl == 1 && !d->has_residual && d->act1 == SI_ACT_SILU && d->act2 == SI_ACT_NONE &&
           d->oh == (d->ih + 2 - 3) / 2 + 1 && d->ow == (d->iw + 2 - 3) / 2 + 1 && d->oh > 0 && d->ow > 0;
}

// the 1x1 conv of the PW form: 64 -> 64 over conv_1's output, bias optional, SiLU, no shortcut (YOLOv5's first C3: cv1 | cv2 fused)
bool pw_ok(const SiConv2dDesc* d) {
    return d->groups == 1 && d->ic == 64 && d->oc == 64 && d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->dh == 1 && d->dw == 1 &&
           d->pt == 0 && d->pl == 0 && !d->has_residual && d->act1 == SI_ACT_SILU && d->act2 == SI_ACT_NONE && d->oh == d->ih && d->ow == d->iw;
}

int fused_launch(const SiConv2dDesc* stem, const SiConv2dDesc* conv, const SiConv2dDesc* pw, const float* in, const void* stem_w_packed,
                 const float* stem_bias, const void* conv_w_packed, const float* conv_bias, const void* pw_w_packed, const float* pw_bias,
                 void* out, int split_oc, void* out2, int out2_ld, si_stream_t stream) {
    const unsigned long long in_bytes = (unsigned long long)stem->n * stem->ih * stem->iw * 3ull * 4ull;
    if (in_bytes >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;
    FusedArgs a;
    a.in = in;
    a.ws = static_cast<const half_t*>(stem_w_packed);
    a.bs = stem->has_bias ? stem_bias : nullptr;
    // the lane-order image sits behind the row-major one in si_hip_conv2d_f16_pack_weight_host's buffer ([oc][9 * 32] rows first)
    a.wl = static_cast<const half_t*>(conv_w_packed) + (size_t)conv->oc * 9 * 32;
    a.bc = conv->has_bias ? conv_bias : nullptr;
    a.out = static_cast<half_t*>(out);
    a.ih = stem->ih; a.iw = stem->iw; a.soh = stem->oh; a.sow = stem->ow;
    a.oh = conv->oh; a.ow = conv->ow; a.oc = conv->oc; a.out_ld = conv->out_ld; a.nb = conv->oc / 32;
    a.tiles_x = (conv->ow + 15) / 16; a.tiles_y = (conv->oh + 3) / 4;
    const long long items = (long long)conv->n * a.tiles_x * a.tiles_y;
    if (items > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    a.items = (int)items;
    a.in_bytes = (unsigned)in_bytes;
    a.wp = nullptr; a.bp = nullptr; a.out_b = nullptr; a.out_b_ld = 0;
    const unsigned long long px = (unsigned long long)conv->n * conv->oh * conv->ow;
    unsigned long long ob0 = ((px - 1) * (unsigned long long)conv->out_ld + conv->oc) * 2ull, ob1 = 0;
    if (pw) {
        ob0 = ((px - 1) * (unsigned long long)pw->out_ld + 32) * 2ull;
        ob1 = ((px - 1) * (unsigned long long)(split_oc ? out2_ld : pw->out_ld) + 32) * 2ull;
    }
    if (ob0 >= 0xFFFFFF00ull || ob1 >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;   // (a destination past 4 GB: 1300 images of 640 x 640)
    a.out_bytes = (unsigned)ob0; a.out_b_bytes = (unsigned)ob1;
    if (pw) {
        a.wp = static_cast<const half_t*>(pw_w_packed) + (size_t)64 * 64;   // (row-major image first, as above)
        a.bp = pw->has_bias ? pw_bias : nullptr;
        a.out_ld = pw->out_ld;
        a.out_b = split_oc ? static_cast<half_t*>(out2) : static_cast<half_t*>(out) + 32;
        a.out_b_ld = split_oc ? out2_ld : pw->out_ld;
    }
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    auto go = [&](auto kern) {
        const int per_cu = si_resident_blocks(kern, 256, 0);
        long long grid = (long long)cus * per_cu;
        if (grid > items) grid = items;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
        return (int)hipGetLastError();
    };
#ifdef SI_EXPERIMENT
    if (SI_ENV_INT("SI_FUSED_BST", 1) == 0) return pw ? go(conv_stem_s2c32_f16_kernel<true, false>) : go(conv_stem_s2c32_f16_kernel<false, false>);
#endif
    return pw ? go(conv_stem_s2c32_f16_kernel<true, true>) : go(conv_stem_s2c32_f16_kernel<false, true>);
}

}  // namespace

extern "C" {

int si_hip_conv2d_stem_s2c32_f16_supported(const SiConv2dDesc* stem, const SiConv2dDesc* conv) {
    if (!stem || !conv) return 0;
    return (stem_ok(stem) && conv_ok(conv) && stem->n == conv->n && stem->oh == conv->ih && stem->ow == conv->iw) ? 1 : 0;
}

int si_hip_conv2d_stem_s2c32_f16(const SiConv2dDesc* stem, const SiConv2dDesc* conv, const float* in, const void* stem_w_packed,
                                 const float* stem_bias, const void* conv_w_packed, const float* conv_bias, void* out,
                                 si_stream_t stream) {
    if (!stem || !conv || !in || !stem_w_packed || !conv_w_packed || !out) return SI_E_BADARG;
    if (stem->has_bias && !stem_bias) return SI_E_BADARG;
    if (conv->has_bias && !conv_bias) return SI_E_BADARG;
    if (!si_hip_conv2d_stem_s2c32_f16_supported(stem, conv)) return SI_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(in) & 15) != 0 || (reinterpret_cast<uintptr_t>(conv_w_packed) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(stem_w_packed) & 15) != 0)
        return SI_E_UNSUPPORTED;
    return fused_launch(stem, conv, nullptr, in, stem_w_packed, stem_bias, conv_w_packed, conv_bias, nullptr, nullptr, out, 0, nullptr, 0, stream);
}

int si_hip_conv2d_stem_s2c32_pw_f16_supported(const SiConv2dDesc* stem, const SiConv2dDesc* conv, const SiConv2dDesc* pw, int split_oc) {
    if (!pw || !si_hip_conv2d_stem_s2c32_f16_supported(stem, conv)) return 0;
    return (conv->oc == 64 && pw_ok(pw) && pw->n == conv->n && pw->ih == conv->oh && pw->iw == conv->ow && (split_oc == 0 || split_oc == 32)) ? 1 : 0;
}

int si_hip_conv2d_stem_s2c32_pw_f16(const SiConv2dDesc* stem, const SiConv2dDesc* conv, const SiConv2dDesc* pw, const float* in,
                                    const void* stem_w_packed, const float* stem_bias, const void* conv_w_packed, const float* conv_bias,
                                    const void* pw_w_packed, const float* pw_bias, void* out, int split_oc, void* out2, int out2_ld,
                                    si_stream_t stream) {
    if (!stem || !conv || !pw || !in || !stem_w_packed || !conv_w_packed || !pw_w_packed || !out || (split_oc && !out2)) return SI_E_BADARG;
    if ((stem->has_bias && !stem_bias) || (conv->has_bias && !conv_bias) || (pw->has_bias && !pw_bias)) return SI_E_BADARG;
    if (!si_hip_conv2d_stem_s2c32_pw_f16_supported(stem, conv, pw, split_oc)) return SI_E_UNSUPPORTED;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(conv_w_packed) | reinterpret_cast<uintptr_t>(stem_w_packed) |
          reinterpret_cast<uintptr_t>(pw_w_packed)) & 15) != 0)
        return SI_E_UNSUPPORTED;
    return fused_launch(stem, conv, pw, in, stem_w_packed, stem_bias, conv_w_packed, conv_bias, pw_w_packed, pw_bias, out, split_oc, out2, out2_ld, stream);
}

}  // extern "C"
