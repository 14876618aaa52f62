// conv_pw_patch_f16.hip -- the 64- / 32-channel C3 bottleneck pair, 1x1 + SiLU -> 3x3 + SiLU (+ shortcut), as ONE persistent launch (round 5).
//
// YOLOv5s at batch 32 runs three such pairs over 80x80x64 maps (reference: models/common.py Bottleneck through
// src/layer/conv_2d.cpp:207-283 twice).  As two launches the 26 MB intermediate is written and read back and the 1x1 is a launch of
// its own (17 + 32 us).  Here the 3x3 keeps the form of conv_c32_patch_f16_kernel<1, 2, .., 64> (conv_igemm_f16.hip: persistent
// workgroups over 4 x 16-pixel output tiles, the 3x3's 36 weight fragments resident in registers, the next item's input in flight
// while this one multiplies) -- but what is fetched is x, the PAIR's input, over the tile's 6 x 18-pixel patch, and the patch itself
// is computed in front of every item's matrix loop: y = SiLU(W1 x + b1) on the same MFMA with the weights as its A operand (a lane
// then ends with four consecutive channels of one pixel: 8-byte LDS writes straight into the patch layout), rounded to fp16 exactly
// as the 1x1 launch would have stored it, ZERO where the patch pixel lies outside the image (the 3x3 pads its input, not the 1x1's).
// Halo recompute 108 / 64 pixels; the 1x1 is 8 MFMAs per wave and item beside the 3x3's 36.  The 160x160x32 pair (YOLOv5s' first C3)
// takes the same kernel over the 32-channel geometry (8 x 16-pixel tiles, 180-pixel patches in six pixel blocks over four waves).
// Same k order, same 16-deep steps, the same epilogue expressions as the two launches: the same bits (tests/test_gpu_f16.py).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

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

constexpr unsigned OOB = 0xFFFFFF00u;   // beyond any resource below: the load returns zeros

struct PwPatchArgs {
    const half_t* x;            // the pair's input [n][ih][iw][x_ld]
    const half_t* wlA;          // 1x1: lane-order weights, 2 column blocks x 4 k-steps x 1 KB
    const float* biasA;
    const half_t* wl;           // 3x3: lane-order weights, 2 column blocks x 36 k-steps x 1 KB
    const float* bias;
    const half_t* res;
    half_t* out;
    int ih, iw, x_ld, out_ld, res_ld;
    unsigned x_bytes;
    unsigned out_bytes;         // extent of `out`: the output stores are buffer stores (countable: see conv_stem_s2c32_f16.hip BST); 0 = pointer stores
    int tiles_x, tiles_y, items;
    // CV3 form (round 6): the C3's closing 1x1 conv over cat(y, z) -- y this pair's output, z the C3's other branch -- computed from the tile
    const half_t* z;            // [n][ih][iw][z_ld], 64 channels
    int z_ld;
    unsigned z_bytes;
    const half_t* w3;           // lane-order weights of the 128 -> 128 conv: 4 column blocks x 8 k-steps x 1 KB
    const float* bias3;
    half_t* out3;
    int out3_ld;
};

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// CB = 64: the geometry of conv_c32_patch_f16_kernel<1, 2, .., 64> -- 4 x 16-pixel tiles, waves 2 (row pairs) x 2 (column blocks), 144-byte
// pixels, 2816-byte patch rows; CB = 32: of <1, 1, .., 32> -- 8 x 16-pixel tiles, four row-pair waves, 80-byte pixels, 1536-byte rows.
// CV3 (CB = 64 only; round 6): the pair is the LAST bottleneck of a YOLOv5 C3 and the C3's closing 1x1 conv (cv3: 128 -> 128 over cat(y, z), SiLU)
// follows in the same launch.  The pair's output tile y (4 x 16 pixels x 64 channels, rounded to fp16 exactly as it would have been stored) goes
// to LDS as [pixel][channel 0..63] beside z's 64 channels of the same pixels (fetched under the 3x3's MFMAs) -- the image lies over the dead x
// buffer --; after a barrier every wave multiplies its 32 pixels by 64 of cv3's 128 output columns (its weights: 32 KB of LDS, lane order; eight
// 16-deep steps over y's block then z's: the generic tiles' k order on the concat buffer, hence their bits), bias + SiLU, and the C3's OUTPUT
// leaves.  Neither y nor the concat buffer is written; one launch and two tensor round trips less per C3.
template <bool HAS_RES, int CB, bool CV3 = false>
__global__ __launch_bounds__(256, CB == 64 ? 2 : 3) void conv_pw_patch_f16_kernel(const PwPatchArgs a) {
    static_assert(CB == 64 || CB == 32, "instantiated forms");
    static_assert(!CV3 || CB == 64, "the cv3 form exists for the 64-channel pair");
    constexpr int TP3 = 272;   // CV3: bytes per pixel of the [64 pixels][128 channels + 8] image (conflict-free ds_read_b128)
    constexpr int PITCH = CB * 2 + 16, ROWP = CB == 64 ? 2816 : 1536, NBW = CB / 32, TR = 2 * (4 / NBW), PR = TR + 2, PC = 18, QS = CB / 16, KS = 9 * QS;
    constexpr int CH8 = CB / 8, NPIX = PR * PC, NCH = NPIX * CH8, N_IT = (NCH + 255) / 256, MT0 = (NPIX + 31) / 32;
    __shared__ __attribute__((aligned(16))) unsigned char patch[PR * ROWP];
    __shared__ __attribute__((aligned(16))) unsigned char xbuf[MT0 * 32 * PITCH];   // x over the patch, pixel-major; rows from NPIX on are never written
    // the 1x1's weights (lane order: column blocks x k-steps x 64 lanes x 16 B) and bias: registers are what this kernel is short of
    __shared__ __attribute__((aligned(16))) unsigned char wa_s[NBW * QS * 1024];
    __shared__ __attribute__((aligned(16))) float bias_a[CB];
    __shared__ __attribute__((aligned(16))) unsigned char w3_s[CV3 ? 4 * 8 * 1024 : 16];
    static_assert(!CV3 || MT0 * 32 * PITCH >= 64 * TP3, "the cv3 tile image fits the x buffer it overlays");

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / NBW, wn = wave - wm * NBW;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.wl), 0, (unsigned)NBW * KS * 1024u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wa = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.wlA), 0, (unsigned)NBW * QS * 1024u, 0x00020000);

    // staging chunks of this thread: chunk c -> (patch pixel, 16-byte channel chunk); item-invariant
    int l_off[N_IT], c_pos[N_IT];   // (c_pos: patch row << 8 | patch column)
#pragma unroll
    for (int i = 0; i < N_IT; ++i) {
        const int c = tid + 256 * i;
        const int q = c / CH8, pr = q / PC;
        c_pos[i] = (pr << 8) | (q - pr * PC);
        l_off[i] = c < NCH ? q * PITCH + (c % CH8) * 16 : -1;
    }
    u32x4 rp[N_IT];
    auto prefetch = [&](int item) {
        const int tx = item % a.tiles_x, t2 = item / a.tiles_x;
        const int ty = t2 % a.tiles_y, img = t2 / a.tiles_y;
        const int y0 = ty * TR - 1, x0 = tx * 16 - 1;
#pragma unroll
        for (int i = 0; i < N_IT; ++i) {
            const int gy = y0 + (c_pos[i] >> 8), gx = x0 + (c_pos[i] & 255);
            const bool ok = l_off[i] >= 0 && (unsigned)gy < (unsigned)a.ih && (unsigned)gx < (unsigned)a.iw;
            const unsigned off = (unsigned)((img * a.ih + gy) * a.iw + gx) * (unsigned)(a.x_ld * 2) + (unsigned)((tid % CH8) * 16);
            rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? off : OOB, 0, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < N_IT; ++i)
            if (l_off[i] >= 0) *reinterpret_cast<u32x4*>(xbuf + l_off[i]) = rp[i];
    };

    int item = blockIdx.x;
    if (item >= a.items) return;
    prefetch(item);
    // resident in registers: this wave's column block of the 3x3 (36 k-steps, tap-major, four 16-channel steps per tap)
    f16x8 wf[KS];
    {
        const unsigned vo = (unsigned)wn * (unsigned)KS * 1024u + (unsigned)lane * 16u;
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wl, vo, (unsigned)(s * 1024), 0));
        // (the 1x1's 8 KB / 2 KB: 16 bytes per thread and pass)
#pragma unroll
        for (int i = 0; i < (NBW * QS * 64 + 255) / 256; ++i)
            if (tid + 256 * i < NBW * QS * 64)
                *reinterpret_cast<u32x4*>(wa_s + (tid + 256 * i) * 16) = __builtin_amdgcn_raw_buffer_load_b128(rs_wa, (unsigned)(tid + 256 * i) * 16u, 0, 0);
    }
    if (CV3) {
        const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.w3), 0, 4u * 8u * 1024u, 0x00020000);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<u32x4*>(w3_s + (tid + 256 * i) * 16) = __builtin_amdgcn_raw_buffer_load_b128(rs_w3, (unsigned)(tid + 256 * i) * 16u, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(CV3 ? a.z : a.x), 0, CV3 ? a.z_bytes : 0u, 0x00020000);
    const unsigned char* const WA = wa_s + wn * QS * 1024 + lane * 16;
    const int o = wn * 32 + l31;
    const float bv = a.bias ? a.bias[o] : 0.0f;
    if (tid < CB) bias_a[tid] = a.biasA ? a.biasA[tid] : 0.0f;
    // 1x1 bias of this lane's channels (weights = A operand: rows (e & 3) + 8 (e >> 2) + 4 lh of the 32-channel block), read per item
    const float* const ba = bias_a + wn * 32 + 4 * lh;
    // the patch pixels this lane produces in phase 0 -- pixel blocks 2 wm, 2 wm + 1 of four (64 channels: this wave's column block of
    // them), blocks wave, wave + 4 of six (32 channels) --: patch offset of its channels, position
    const int tq0 = CB == 64 ? 2 * wm : wave, tqs = CB == 64 ? 1 : 4;
    int p_off[2], p_pos[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int q = (tq0 + tqs * t) * 32 + l31, pr = q / PC, px = q - pr * PC;
        p_pos[t] = (pr << 8) | px;
        p_off[t] = q < NPIX ? pr * ROWP + px * PITCH + wn * 64 + lh * 8 : -1;
    }
    commit();
    // (everything above has LANDED before the loop: see conv_c32_patch_f16_kernel on why the weights must not be waited for inside it)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();

    const unsigned char* const X = xbuf + (tq0 * 32 + l31) * PITCH + lh * 16;
    const bool two = CB == 64 || wave + 4 < MT0;   // (wave-uniform: this wave has a second pixel block)
    const int x1 = two ? tqs * 32 * PITCH : 0;
    const unsigned char* const P = patch + (2 * wm + (l31 >> 4)) * ROWP + (l31 & 15) * PITCH + lh * 16;
    for (; item < a.items; item += gridDim.x) {
        const int next = item + gridDim.x;
        if (next < a.items) prefetch(next);
        const int tx = item % a.tiles_x, t2 = item / a.tiles_x;
        const int ty = t2 % a.tiles_y, img = t2 / a.tiles_y;
        const int oy0 = ty * TR + 2 * wm, ox0 = tx * 16;
        // ---- phase 0: the 1x1 conv + SiLU over the patch's pixels, into the patch
        {
            const int y0 = ty * TR - 1, x0 = tx * 16 - 1;
            auto finish = [&](const f32x16& acc0, int t) {
                if (p_off[t] >= 0) {
                    const bool inside = (unsigned)(y0 + (p_pos[t] >> 8)) < (unsigned)a.ih && (unsigned)(x0 + (p_pos[t] & 255)) < (unsigned)a.iw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 bj = *reinterpret_cast<const f32x4*>(ba + 8 * j);
                        f16x4 hv;
#pragma unroll
                        for (int i = 0; i < 4; ++i) hv[i] = si_store_cast<half_t>(silu(acc0[4 * j + i] + bj[i]));
                        if (!inside) hv = f16x4{(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
                        *reinterpret_cast<f16x4*>(patch + p_off[t] + j * 16) = hv;
                    }
                }
            };
            if constexpr (CB == 64) {
                // both pixel blocks side by side (every weight fragment read once)
                f32x16 acc0[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc0[t][e] = 0.0f;
                f16x8 xf[2][2], wa[2];
                wa[0] = *reinterpret_cast<const f16x8*>(WA);
#pragma unroll
                for (int t = 0; t < 2; ++t) xf[0][t] = *reinterpret_cast<const f16x8*>(X + t * x1);
#pragma unroll
                for (int s = 0; s < QS; ++s) {
                    if (s + 1 < QS) {
                        wa[(s + 1) & 1] = *reinterpret_cast<const f16x8*>(WA + (s + 1) * 1024);
#pragma unroll
                        for (int t = 0; t < 2; ++t) xf[(s + 1) & 1][t] = *reinterpret_cast<const f16x8*>(X + t * x1 + (s + 1) * 32);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[s & 1], xf[s & 1][t], acc0[t], 0, 0, 0);
                }
                finish(acc0[0], 0);
                finish(acc0[1], 1);
            } else {
                // one pixel block after the other (one accumulator set: three workgroups per CU leave 168 registers)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (t == 1 && !two) break;
                    f32x16 acc0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc0[e] = 0.0f;
                    f16x8 xf[QS], wa[QS];
#pragma unroll
                    for (int s = 0; s < QS; ++s) {
                        wa[s] = *reinterpret_cast<const f16x8*>(WA + s * 1024);
                        xf[s] = *reinterpret_cast<const f16x8*>(X + t * x1 + s * 32);
                    }
#pragma unroll
                    for (int s = 0; s < QS; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[s], xf[s], acc0, 0, 0, 0);
                    finish(acc0, t);
                }
            }
        }
        __syncthreads();   // the patch is complete (and x of this item is dead)
        // CV3: z's 64 channels of the tile's 64 pixels, requested here, written beside y behind the 3x3's MFMAs (512 16-byte chunks: two per thread)
        u32x4 zr[CV3 ? 2 : 1];
        if (CV3) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + 256 * i, px = c >> 3, oy = ty * TR + (px >> 4), ox = tx * 16 + (px & 15);
                const unsigned off = (unsigned)((img * a.ih + oy) * a.iw + ox) * (unsigned)(a.z_ld * 2) + (unsigned)((c & 7) * 16);
                zr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, (oy < a.ih && ox < a.iw) ? off : OOB, 0, 0);
            }
        }
        // ---- the 3x3 conv from the patch: conv_c32_patch_f16_kernel's loop and epilogue
        half_t rv[HAS_RES ? 16 : 1];
        auto load_res = [&]() {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
                const bool in = oy < a.ih && ox < a.iw;
                rv[e] = a.res[in ? (size_t)((img * a.ih + oy) * a.iw + ox) * a.res_ld + o : (size_t)0];
            }
        };
        // (CV3: the kernel has no registers left for sixteen values held across the matrix loop -- the shortcut is fetched behind it)
        if (HAS_RES && !CV3) load_res();
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        auto frag = [&](int s) {
            const int tap = s / QS, ky = tap / 3, kx = tap - 3 * ky;
            return *reinterpret_cast<const f16x8*>(P + ky * ROWP + kx * PITCH + (s % QS) * 32);
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
        if constexpr (CV3) {
            // y into the tile image [pixel 32 wm + r][channel o] (over the dead x buffer), z beside it; one barrier; cv3
            if (HAS_RES) load_res();
            unsigned char* const T = xbuf;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                float v = silu(acc[e] + bv);
                if (HAS_RES) v += (float)rv[e];
                *reinterpret_cast<half_t*>(T + (32 * wm + r) * TP3 + o * 2) = si_store_cast<half_t>(v);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + 256 * i;
                *reinterpret_cast<u32x4*>(T + (c >> 3) * TP3 + 128 + (c & 7) * 16) = zr[i];
            }
            __syncthreads();
            const unsigned char* const TA = T + (32 * wm + l31) * TP3 + lh * 16;
#pragma unroll
            for (int cbk = 0; cbk < 2; ++cbk) {
                const int nb3 = 2 * wn + cbk;   // this wave's column blocks of cv3: 2 wn, 2 wn + 1
                const unsigned char* const W3 = w3_s + nb3 * 8 * 1024 + lane * 16;
                f32x16 acc3;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc3[e] = 0.0f;
                // (fragments in two groups of four k-steps: y's block, then z's -- the kernel has no registers to spare)
#pragma unroll
                for (int g3 = 0; g3 < 2; ++g3) {
                    f16x8 ta[4], tb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ta[i] = *reinterpret_cast<const f16x8*>(TA + (4 * g3 + i) * 32);
                        tb[i] = *reinterpret_cast<const f16x8*>(W3 + (4 * g3 + i) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta[i], tb[i], acc3, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int o3 = nb3 * 32 + l31;
                const float b3 = a.bias3 ? a.bias3[o3] : 0.0f;
                half_t* const ob = a.out3 + o3;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
                    if (oy < a.ih && ox < a.iw) ob[(size_t)((img * a.ih + oy) * a.iw + ox) * a.out3_ld] = si_store_cast<half_t>(silu(acc3[e] + b3));
                }
            }
            if (next < a.items) __syncthreads();   // every wave is done with the tile image: the next item's x goes over it
        } else {
            // buffer stores, the out-of-range pixel as an out-of-range offset: no branch around a store, so the wait in front of the next item's
            // commit() counts them instead of draining them (conv_stem_s2c32_f16.hip BST; LAB_NOTEBOOK R6.10)
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
                float v = silu(acc[e] + bv);
                if (HAS_RES) v += (float)rv[e];
                const unsigned off = ((unsigned)((img * a.ih + oy) * a.iw + ox) * (unsigned)a.out_ld + (unsigned)o) * 2u;
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, si_store_cast<half_t>(v)), rs_out,
                                                      (oy < a.ih && ox < a.iw) ? off : OOB, 0, 0);
            }
        }
        if (next < a.items) {
            commit();
            __syncthreads();   // x of the next item is in place; every wave is done with this item's patch
        }
    }
}

// on when the shape allows unless the 3x3 conv's plan says 0 (SiConvPlan::f16_pw_patch)
bool pw_patch_on(const SiConv2dDesc* d) {
    const int v = (d && d->plan && d->plan->f16_pw_patch >= 0) ? d->plan->f16_pw_patch : SI_ENV_INT("SI_CONV_F16_PW_PATCH", 1);
    return v != 0;
}

int cu_count() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}

}  // namespace

// the pair's shapes: a 3x3 stride-1 pad-1 conv over c -> c channels (c = 64 on maps of whole 4 x 16-pixel tiles, c = 32 of whole 8 x 16 ones)
// with SiLU (optional shortcut), behind a 1x1 conv + SiLU over the same c channels and the same map
bool si_conv_pw_patch_f16_ok(const SiConv2dDesc* pw, const SiConv2dDesc* d) {
    if (!pw || !d || !pw_patch_on(d)) return false;
    const int c = d->ic;
    if (c != 64 && c != 32) return false;
    const bool conv = d->groups == 1 && d->oc == c && d->kh == 3 && d->kw == 3 && d->sh == 1 && d->sw == 1 && d->dh == 1 && d->dw == 1 &&
                      d->pt == 1 && d->pl == 1 && d->oh == d->ih && d->ow == d->iw && d->ow % 16 == 0 && d->oh % (c == 64 ? 4 : 8) == 0 &&
                      d->act1 == SI_ACT_SILU && d->act2 == SI_ACT_NONE && (!d->has_residual || d->res_ld % 2 == 0) && d->n > 0;
    const bool pwc = pw->groups == 1 && pw->kh == 1 && pw->kw == 1 && pw->sh == 1 && pw->sw == 1 && pw->pt == 0 && pw->pl == 0 && pw->dh == 1 && pw->dw == 1 &&
                     pw->ic == c && pw->oc == c && pw->n == d->n && pw->ih == d->ih && pw->iw == d->iw && pw->oh == d->ih && pw->ow == d->iw &&
                     pw->act1 == SI_ACT_SILU && pw->act2 == SI_ACT_NONE && !pw->has_residual && pw->in_ld % 8 == 0;
    if (!conv || !pwc) return false;
    const unsigned long long xb = (unsigned long long)pw->n * pw->ih * pw->iw * pw->in_ld * 2ull;
    const long long items = (long long)d->n * (d->ow / 16) * (d->oh / (c == 64 ? 4 : 8));
    return xb < 0xFFFFFF00ull && items <= 0x7fffffffLL;
}

// the C3's closing 1x1 conv of the CV3 form: 128 -> 128 over cat(y, z) with y the 64-channel pair's output, SiLU, no shortcut
static bool cv3_ok(const SiConv2dDesc* pw, const SiConv2dDesc* d, const SiConv2dDesc* c3) {
    return c3 && si_conv_pw_patch_f16_ok(pw, d) && d->ic == 64 && c3->groups == 1 && c3->ic == 128 && c3->oc == 128 && c3->kh == 1 && c3->kw == 1 &&
           c3->sh == 1 && c3->sw == 1 && c3->dh == 1 && c3->dw == 1 && c3->pt == 0 && c3->pl == 0 && !c3->has_residual && c3->act1 == SI_ACT_SILU &&
           c3->act2 == SI_ACT_NONE && c3->n == d->n && c3->ih == d->oh && c3->iw == d->ow && c3->oh == d->oh && c3->ow == d->ow;
}

static int pw_patch_launch(const SiConv2dDesc* pw, const SiConv2dDesc* d, const void* in, const void* wlA, const float* biasA, const void* wl,
                           const float* bias, const void* residual, void* out, hipStream_t s, const SiConv2dDesc* c3, const void* z, int z_ld,
                           const void* w3, const float* bias3);

int si_conv_pw_patch_f16_launch(const SiConv2dDesc* pw, const SiConv2dDesc* d, const void* in, const void* wlA, const float* biasA, const void* wl,
                                const float* bias, const void* residual, void* out, hipStream_t s) {
    return pw_patch_launch(pw, d, in, wlA, biasA, wl, bias, residual, out, s, nullptr, nullptr, 0, nullptr, nullptr);
}

extern "C" int si_hip_conv2d_pw_cv3_f16_supported(const SiConv2dDesc* pw, const SiConv2dDesc* conv, const SiConv2dDesc* cv3) {
    return (pw && conv && cv3_ok(pw, conv, cv3)) ? 1 : 0;
}

extern "C" int si_hip_conv2d_pw_cv3_f16(const SiConv2dDesc* pw, const SiConv2dDesc* conv, const SiConv2dDesc* cv3, const void* in, const void* pw_w_packed,
                                        const float* pw_bias, const void* w_packed, const float* bias, const void* residual, const void* z, int z_ld,
                                        const void* cv3_w_packed, const float* cv3_bias, void* out, si_stream_t stream) {
    if (!pw || !conv || !cv3 || !in || !pw_w_packed || !w_packed || !z || !cv3_w_packed || !out) return SI_E_BADARG;
    if ((pw->has_bias && !pw_bias) || (conv->has_bias && !bias) || (cv3->has_bias && !cv3_bias) || (conv->has_residual && !residual)) return SI_E_BADARG;
    if (!cv3_ok(pw, conv, cv3) || z_ld % 8 != 0 || (reinterpret_cast<uintptr_t>(z) & 15) != 0 || (reinterpret_cast<uintptr_t>(cv3_w_packed) & 15) != 0) return SI_E_UNSUPPORTED;
    // (si_hip_conv2d_f16_pack_weight_host: the row-major image, then the lane-order one)
    const size_t cc = (size_t)conv->ic * conv->ic;
    const half_t* const wA = static_cast<const half_t*>(pw_w_packed) + cc;
    const half_t* const wB = static_cast<const half_t*>(w_packed) + 9 * cc;
    const half_t* const w3 = static_cast<const half_t*>(cv3_w_packed) + (size_t)128 * 128;
    return pw_patch_launch(pw, conv, in, wA, pw_bias, wB, bias, residual, out, static_cast<hipStream_t>(stream), cv3, z, z_ld, w3, cv3_bias);
}

static int pw_patch_launch(const SiConv2dDesc* pw, const SiConv2dDesc* d, const void* in, const void* wlA, const float* biasA, const void* wl,
                           const float* bias, const void* residual, void* out, hipStream_t s, const SiConv2dDesc* c3, const void* z, int z_ld,
                           const void* w3, const float* bias3) {
    if (!si_conv_pw_patch_f16_ok(pw, d)) return SI_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(in) & 15) != 0 || (d->has_residual && (reinterpret_cast<uintptr_t>(residual) & 1) != 0)) return SI_E_UNSUPPORTED;
    PwPatchArgs a;
    a.x = static_cast<const half_t*>(in);
    a.wlA = static_cast<const half_t*>(wlA);
    a.biasA = pw->has_bias ? biasA : nullptr;
    a.wl = static_cast<const half_t*>(wl);
    a.bias = d->has_bias ? bias : nullptr;
    a.res = d->has_residual ? static_cast<const half_t*>(residual) : nullptr;
    a.out = static_cast<half_t*>(out);
    a.ih = d->ih; a.iw = d->iw; a.x_ld = pw->in_ld; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.x_bytes = (unsigned)((unsigned long long)pw->n * pw->ih * pw->iw * pw->in_ld * 2ull);
    const unsigned long long ob = (((unsigned long long)d->n * d->oh * d->ow - 1) * d->out_ld + d->oc) * 2ull;
    if (ob >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;
    a.out_bytes = (unsigned)ob;
    a.tiles_x = d->ow / 16; a.tiles_y = d->oh / (d->ic == 64 ? 4 : 8);
    a.items = d->n * a.tiles_x * a.tiles_y;
    a.z = nullptr; a.z_ld = 0; a.z_bytes = 0; a.w3 = nullptr; a.bias3 = nullptr; a.out3 = nullptr; a.out3_ld = 0;
    if (c3) {
        const unsigned long long zb = (unsigned long long)d->n * d->oh * d->ow * z_ld * 2ull;
        if (zb >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;
        a.z = static_cast<const half_t*>(z); a.z_ld = z_ld; a.z_bytes = (unsigned)zb;
        a.w3 = static_cast<const half_t*>(w3); a.bias3 = c3->has_bias ? bias3 : nullptr;
        a.out3 = static_cast<half_t*>(out); a.out3_ld = c3->out_ld;
        a.out = nullptr;
    }
    auto go = [&](auto kern) {
        const int per_cu = si_resident_blocks(kern, 256, 0);
        long long grid = (long long)cu_count() * per_cu;
        if (grid > a.items) grid = a.items;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, s, a);
        return (int)hipGetLastError();
    };
    if (c3) return d->has_residual ? go(conv_pw_patch_f16_kernel<true, 64, true>) : go(conv_pw_patch_f16_kernel<false, 64, true>);
    if (d->ic == 32) return d->has_residual ? go(conv_pw_patch_f16_kernel<true, 32>) : go(conv_pw_patch_f16_kernel<false, 32>);
    return d->has_residual ? go(conv_pw_patch_f16_kernel<true, 64>) : go(conv_pw_patch_f16_kernel<false, 64>);
}
