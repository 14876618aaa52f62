// conv_stem_roll.hip -- the fp32 first-layer ("stem") convolution as a ROLLING row window: RGB input, stride 2, large kernel
// (YOLOv5s 640x640x3 -> 320x320x32 6x6 s2 p2 -- the reference's "Test Conv2d layer 2" shape, test/test_layer/test_conv_2d.cpp:
// 276-416 -- ResNet18 7x7 s2 p3 -> 64, MobileNet 3x3 s2 p1 -> 16).  Replaces Conv2d::ForwardIm2Col (src/layer/conv_2d.cpp:
// 207-283) + AddBiasNHWC + the activation pass for those shapes.
//
// What was wrong with the first stem kernel (conv_smallc.hip), measured (profiles/traffic.json, r01_pmc_lds_stem.txt):
//   * every item staged the KH input rows of ONE output row, so each input row was fetched KH / stride = 3 times
//     (487 MB for a 157 MB image);
//   * the A operand of v_mfma_f32_32x32x2_f32 was read with a lane stride of stride * C = 6 floats: lanes l and l + 16 of a
//     32-lane group hit the same bank (2-way conflict on every read, 30 % of the LDS cycles).
// Here a persistent workgroup owns (image, column tile, a run of RC output rows) and walks DOWN the run: the KH + 2 input rows
// live in an LDS ring; per output row only the 2 NEW input rows are fetched (one 16-byte load per thread per row, issued before
// the MFMAs of the current row, committed after them into the two slots the current row does not read -- ONE barrier per row).
// The contraction runs on v_mfma_f32_16x16x4_f32 (same FLOP/cycle as 32x32x2): its A operand is 16 pixels x 4 k, so a 32-lane
// group reads 16 pixels (6-float stride: 16 distinct even/odd banks) x 2 consecutive k -- conflict free -- and the flattened
// K = KH * KW * C axis (108 for the YOLOv5 stem) is cut into steps of 4 with no per-row padding (the old kernel padded every
// kernel row to an even length).  B: the weight image [k][oc] in LDS; lanes with odd k read the two 16-channel halves of a
// 32-channel tile in the opposite order (k and k + 1 then sit on different banks) and swap them back in registers.
// Four waves per workgroup, each owning PB blocks of 16 pixels: every SIMD of the CU carries the same MFMA load (the first
// version gave 5 waves 32 pixels each -- one SIMD did twice the work of the others between two barriers: 0.248 ms on the
// YOLOv5s stem at batch 32 against 0.259 ms for the kernel it replaced; balanced: see DESIGN.md).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "si_hip.h"
#include "si_hip_internal.h"

// one rounding per operation, whatever instantiation a pixel goes through (bit-exact batch sharding)
#pragma clang fp contract(off)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

// development ablations (tools/build_exp.sh): 1 no output stores, 2 no activation, 4 no row fetch in the loop
#ifndef SI_STEM_ABLATE
#define SI_STEM_ABLATE 0
#endif
// (development switch: two waves per SIMD promised to the compiler, accumulators in the unified VGPR file, no v_accvgpr_read per
// value in the epilogue: 0.2100 -> 0.2080 ms, ResNet18 stem 0.1448 -> 0.1431; not worth a non-default register contract)
#if defined(SI_STEM_UNIFIED_REGS)
#define SI_STEM_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define SI_STEM_WAVES_ATTR
#endif
#ifndef SI_STEM_EXP   // development: 1 compile-time epilogue, 2 padded weight pitch (no operand swap), 4 literal-zero first step,
#define SI_STEM_EXP 15   // 8 buffer stores with scalar tile offsets
#endif

namespace {

struct StemArgs {
    const float* in;
    const float* w;      // [STEPS * 4][32 * NT], zero rows behind k = KH*KW*3
    const float* bias;
    const float* res;
    void* out;
    int n, ih, iw, in_ld;
    int oh, ow, oc, out_ld, res_ld;
    int pt, pl;
    int row_len;         // floats staged per input row (multiple of 4)
    int rc;              // output rows per task
    int w_tiles, row_chunks, tasks;
    unsigned in_bytes;
    unsigned out_bytes;  // extent of `out` when it fits a buffer resource with 16-byte aligned rows, else 0
    int act1, act2;
    float act_param;
};

__device__ __forceinline__ float stem_act(int act, float v, float p) {
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

// NW waves, each PB blocks of 16 output pixels along W; NT 32-wide output-channel tiles per wave; KH x KW kernel, 3 channels,
// stride 2.
// EP: the epilogue known at compile time -- 1: bias + SiLU, 2: bias + ReLU (no residual, no second activation: the YOLOv5 / ResNet
// stems; straight-line code), 0: anything (runtime switches per tile).
template <int NW, int PB, int NT, int KH, int KW, typename OutT, bool VEC, int EP = 0>
__global__ __launch_bounds__(NW * 64) SI_STEM_WAVES_ATTR void conv_stem_roll_kernel(const StemArgs a) {
    constexpr int C = 3, S = 2, PXS = S * C;
    constexpr int KWC = KW * C, K = KH * KWC, STEPS = (K + 3) / 4;
    constexpr int NR = KH + S;               // ring slots: the KH rows of the current output row + the S incoming ones
    constexpr int WLD = 32 * NT;
    // LDS pitch of a weight row.  Two 32-channel tiles: 16 floats of padding put rows k and k + 1 sixteen banks apart, so the two
    // k of a 32-lane group read their 16 channels from disjoint banks with no operand swap (ResNet stem 0.178 -> 0.163 ms).  One
    // tile: the padded image costs the wide instantiation its second resident workgroup (0.237 -> 0.297 ms), so there lanes with
    // odd k read the two 16-channel halves in the opposite order and swap them back in registers.
    constexpr bool PADW = (SI_STEM_EXP & 2) && NT == 2;
    constexpr int WLP = WLD + (PADW ? 16 : 0);
    constexpr int NTHR = NW * 64, TOW = 16 * PB * NW;
    constexpr int NOC = 2 * NT;              // 16-channel MFMA column tiles per wave
    constexpr int VPT = ((3 + (TOW - 1) * PXS + KWC + 3) / 4 + NTHR - 1) / NTHR;   // 16-byte vectors per thread per staged row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const ring = smem;
    float* const wl = smem + NR * a.row_len;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const int row_floats = a.iw * C;

    // ---- once per persistent workgroup: weight image and bias
    {
        const int nvec = STEPS * 4 * WLD / 4;
        const float4* src = reinterpret_cast<const float4*>(a.w);
        for (int i = tid; i < nvec; i += NTHR) {
            const int row = i / (WLD / 4), c4 = i - row * (WLD / 4);
            *reinterpret_cast<float4*>(wl + row * WLP + c4 * 4) = src[i];
        }
    }
    // the MFMA runs with the weights as its A operand: a lane's 4 accumulator registers are 4 CONSECUTIVE CHANNELS
    // (u * 16 + 4 * kq + e) of ONE pixel (l15), so the epilogue stores 16 bytes per lane and tile
    f32x4 bv[NOC];
#pragma unroll
    for (int u = 0; u < NOC; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = u * 16 + 4 * kq + q;
            bv[u][q] = (a.bias && o < a.oc) ? a.bias[o] : 0.0f;
        }
    // 4 floats [e, e + 4) of image row y (zeros outside the image, or when this thread has no vector `live`)
    auto load_row = [&](int img, int y, int e, bool live) -> f32x4 {
        const bool yok = live && (unsigned)y < (unsigned)a.ih;
        if (VEC) {
            // the lane's part of the address is its float index in the row (or the out-of-range offset: the hardware returns 0),
            // the row's part a scalar offset (the range check does not see it; valid rows keep it inside the tensor)
            const bool ok = yok && e >= 0 && e + 3 < row_floats;
            const unsigned off = ok ? (unsigned)e * 4u : 0xFFFFFF00u;
            const unsigned row = (unsigned)y < (unsigned)a.ih ? (unsigned)((img * a.ih + y) * row_floats) * 4u : 0u;   // wave-uniform
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, row, 0));
        }
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ee = e + t;
            const int px = ee / C, ch = ee - px * C;
            unsigned off = ((unsigned)(((img * a.ih + y) * a.iw + px) * a.in_ld + ch)) * 4u;
            if (!(yok && ee >= 0 && ee < row_floats)) off = 0xFFFFFF00u;
            v[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, off, 0, 0));
        }
        return v;
    };

    for (int task = blockIdx.x; task < a.tasks; task += gridDim.x) {
        int tt = task;
        const int wt = tt % a.w_tiles; tt /= a.w_tiles;
        const int rck = tt % a.row_chunks;
        const int img = tt / a.row_chunks;
        const int oy_a = rck * a.rc;
        const int rows = min(a.rc, a.oh - oy_a);
        const int ox0 = wt * TOW;
        const int x0 = (ox0 * S - a.pl) * C;          // float index of the window start in an image row (may be negative)
        const int e0 = x0 & ~3;                       // staged window start: thread t holds floats [e0 + 4 (t + v NTHR), +4), v < VPT
        const int shift = x0 & 3;
        const int iy_a = oy_a * S - a.pt;

        __syncthreads();                               // the previous task's last row has been read by every wave
        // the first KH rows of the run
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int f = 4 * (tid + v * NTHR);
            const bool live = f < a.row_len;
            f32x4 first[KH];
#pragma unroll
            for (int r = 0; r < KH; ++r) first[r] = load_row(img, iy_a + r, e0 + f, live);
            if (live) {
#pragma unroll
                for (int r = 0; r < KH; ++r) *reinterpret_cast<f32x4*>(ring + r * a.row_len + f) = first[r];
            }
        }
        __syncthreads();

        // lane's position inside a staged row: pixel (16 PB wave + l15) of the tile, first element of its tap window
        const int px_base = shift + (wave * 16 * PB + l15) * PXS;
        int sb = 0;                                    // ring slot of the current output row's first input row

        // One tile (16 pixels x 16 channels) of a finished row: bias, activation, residual, 16-byte store.
        // 16x16 C/D map with the weights as A: row (channel) = 4 * (lane >> 4) + e, column (pixel) = lane & 15.
        const bool simple = a.res == nullptr && a.act2 == SI_ACT_NONE;
        const bool vst = (a.out_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && sizeof(OutT) == 4;
        // The compile-time epilogues store through a buffer resource: the lane's part of the address (its pixel inside a 16-pixel
        // block, its 4 channels) is one VGPR computed once per task, the tile's part (image row, block, channel tile) a scalar
        // offset -- the 64-bit pixel index and pointer arithmetic cost ~15 vector instructions per tile, a third of what a row
        // issues beside its MFMAs.  A pixel block past the row's end, or a channel group past oc, stores to the out-of-range offset.
        constexpr bool bst = EP != 0 && (SI_STEM_EXP & 8) && sizeof(OutT) == 4;   // (the launcher picks EP != 0 only with out_bytes != 0)
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
        const unsigned st_lane = (unsigned)(((wave * PB * 16 + l15) * a.out_ld + 4 * kq) * 4);
        unsigned st_mask = 0;                         // bit h * NOC + u: this lane's (pixel block h, channel tile u) lies inside the tensor
#pragma unroll
        for (int h = 0; h < PB; ++h)
#pragma unroll
            for (int u = 0; u < NOC; ++u)
                if (ox0 + (wave * PB + h) * 16 + l15 < a.ow && u * 16 + 4 * kq + 3 < a.oc) st_mask |= 1u << (h * NOC + u);
        auto epilogue_tile = [&](const f32x4& accv, int h, int u, int oy) {
            if constexpr (bst) {
                f32x4 v = accv + bv[u];
                if (EP == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[q]));
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.0f);
                }
                const unsigned row = (unsigned)((img * a.oh + oy) * a.ow + ox0) * (unsigned)(a.out_ld * 4);   // wave-uniform
                const unsigned soff = row + (unsigned)(h * 16 * a.out_ld * 4 + u * 64);
                // (the tile's offset is ADDED to the lane's, not passed as the instruction's scalar offset: a 16-byte store with an SGPR
                // offset reads its data registers one cycle late, and with the next tile's arithmetic right behind it the last
                // four lanes of every 16 stored the next tile's first value -- 1216 wrong elements in 3.3 M on image 31 of 32)
                const unsigned voff = ((st_mask >> (h * NOC + u)) & 1u) ? st_lane + soff : 0xFFFFFF00u;
                if (!(SI_STEM_ABLATE & 1)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rs_out, voff, 0, 0);
            } else {
            const int ox = ox0 + (wave * PB + h) * 16 + l15;
            const int o = u * 16 + 4 * kq;
            if (ox >= a.ow || o >= a.oc) return;
            const size_t pix = (size_t)(img * a.oh + oy) * a.ow + ox;
            OutT* const opix = static_cast<OutT*>(a.out) + pix * a.out_ld;
            f32x4 v = accv + bv[u];
            if (SI_STEM_ABLATE & 2) {
            } else if (EP == 1 && (SI_STEM_EXP & 1)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = v[q] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[q]));
            } else if (EP == 2 && (SI_STEM_EXP & 1)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.0f);
            } else if (simple && a.act1 == SI_ACT_SILU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = v[q] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[q]));
            } else if (simple && a.act1 == SI_ACT_RELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.0f);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t2 = stem_act(a.act1, v[q], a.act_param);
                    if (a.res && o + q < a.oc) t2 += a.res[pix * a.res_ld + o + q];
                    v[q] = stem_act(a.act2, t2, a.act_param);
                }
            }
            if (SI_STEM_ABLATE & 1) {
                if (v[0] == 1234.5678f) opix[0] = si_store_cast<OutT>(v[0]);   // keeps the values live, (almost) never stores
            } else if (vst && o + 3 < a.oc) {
                *reinterpret_cast<f32x4*>(opix + o) = v;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (o + q < a.oc) opix[o + q] = si_store_cast<OutT>(v[q]);
            }
            }
        };

        // One pipeline stage: the MFMAs of output row t into `cur`, with the epilogue of row t - 1 (`prev`) cut into its tiles and
        // placed between the MFMA groups -- the epilogue's VALU / store instructions issue in the shadow of the matrix pipe
        // instead of after it (a wave is in order: appended to the row, the epilogue cost 0.06 of 0.24 ms on the YOLOv5s stem,
        // and the co-resident workgroup runs in lock-step, so nothing else covered it).
        // PIPE (the 5-block instantiation): as described.  Otherwise the epilogue of row t follows its MFMAs directly (on the
        // narrower instantiations the interleaved form measured 13-15 % slower: more live registers, same issue pressure).
        constexpr bool PIPE = PB >= 4;
        auto stage = [&](f32x4 (&cur)[PB][NOC], f32x4 (&prev)[PB][NOC], int t) {
            const bool compute = t < rows, drain = PIPE && t > 0;
            f32x4 nxt[S][VPT];
            const bool more = t + 1 < rows;
            if (more) {
#pragma unroll
                for (int i = 0; i < S; ++i)
#pragma unroll
                    for (int v = 0; v < VPT; ++v) {
                        const int f = 4 * (tid + v * NTHR);
                        nxt[i][v] = (SI_STEM_ABLATE & 4) ? f32x4{0.f, 0.f, 0.f, 0.f} : load_row(img, iy_a + KH + t * S + i, e0 + f, f < a.row_len);
                    }
            }
            // ring row offsets of this output row's KH input rows (wave-uniform)
            int rowoff[KH];
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                int sl = sb + ky;
                if (sl >= NR) sl -= NR;
                rowoff[ky] = sl * a.row_len;
            }
            if (compute && !(SI_STEM_EXP & 4)) {
#pragma unroll
                for (int h = 0; h < PB; ++h)
#pragma unroll
                    for (int u = 0; u < NOC; ++u) cur[h][u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            constexpr int TILES = PB * NOC;
            constexpr int EVERY = (STEPS - 2) / TILES > 0 ? (STEPS - 2) / TILES : 1;   // one epilogue tile every EVERY steps
#pragma clang loop unroll(full)
            for (int s = 0; s < STEPS; ++s) {
                if (compute) {
                    const int k0 = 4 * s;
                    const int ky0 = k0 / KWC, j0 = k0 - ky0 * KWC;      // compile time after unrolling
                    int off;
                    if (j0 + 3 < KWC && k0 + 3 < K) {
                        off = rowoff[ky0] + j0 + kq;                    // the four k of this step lie in one kernel row
                    } else {
                        // the step straddles a kernel-row boundary (or runs past K, where the weights are zero: any finite
                        // element will do -- the ring holds image data or zeros)
                        const int ky1 = ky0 + 1 < KH ? ky0 + 1 : KH - 1;
                        const int jl = j0 + kq;
                        off = jl < KWC ? rowoff[ky0] + jl : rowoff[ky1] + (ky0 + 1 < KH ? jl - KWC : 0);
                    }
                    float av[PB];
#pragma unroll
                    for (int h = 0; h < PB; ++h) av[h] = ring[px_base + h * 16 * PXS + off];
                    float b[NOC];
                    const float* const wrow = wl + (k0 + kq) * WLP + l15;
                    if (PADW) {
#pragma unroll
                        for (int u = 0; u < NOC; ++u) b[u] = wrow[u * 16];
                    } else {
                        const bool odd = kq & 1;
#pragma unroll
                        for (int u = 0; u < NOC; u += 2) {
                            const float r0 = wrow[u * 16 + (odd ? 16 : 0)];
                            const float r1 = wrow[u * 16 + (odd ? 0 : 16)];
                            b[u] = odd ? r1 : r0;
                            b[u + 1] = odd ? r0 : r1;
                        }
                    }
                    // (the first step accumulates onto the literal zero: no register writes to clear the row's accumulators)
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < PB; ++h)
#pragma unroll
                        for (int u = 0; u < NOC; ++u) cur[h][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u], av[h], (s == 0 && (SI_STEM_EXP & 4)) ? zero : cur[h][u], 0, 0, 0);
                }
                if (drain && s >= 1 && (s - 1) % EVERY == 0 && (s - 1) / EVERY < TILES) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int tile = (s - 1) / EVERY;
                    epilogue_tile(prev[tile / NOC][tile % NOC], tile / NOC, tile % NOC, oy_a + t - 1);
                }
            }
            if (drain) {   // tiles that did not fit the schedule (more tiles than steps)
#pragma unroll
                for (int tile = (STEPS - 2) / EVERY + 1; tile < TILES; ++tile)
                    epilogue_tile(prev[tile / NOC][tile % NOC], tile / NOC, tile % NOC, oy_a + t - 1);
            }
            if (!PIPE && compute) {
#pragma unroll
                for (int h = 0; h < PB; ++h)
#pragma unroll
                    for (int u = 0; u < NOC; ++u) epilogue_tile(cur[h][u], h, u, oy_a + t);
            }
            // the S new rows go into the slots of the rows the NEXT output row no longer needs -- slots this row did not read
            // either (its rows are sb .. sb + KH - 1), so no barrier is needed before the writes, only after them
            if (more) {
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    int sl = sb + KH + i;
                    if (sl >= NR) sl -= NR;
#pragma unroll
                    for (int v = 0; v < VPT; ++v) {
                        const int f = 4 * (tid + v * NTHR);
                        if (f < a.row_len) *reinterpret_cast<f32x4*>(ring + sl * a.row_len + f) = nxt[i][v];
                    }
                }
                __syncthreads();
            }
            sb += S;
            if (sb >= NR) sb -= NR;
        };

        f32x4 acc0[PB][NOC], acc1[PB][NOC];
        if (PIPE) {
            for (int t = 0; t <= rows; t += 2) {
                stage(acc0, acc1, t);
                if (t + 1 <= rows) stage(acc1, acc0, t + 1);
            }
        } else {
            for (int t = 0; t < rows; ++t) stage(acc0, acc1, t);
        }
    }
}

struct StemShape {
    int kh, kw, nt;
};

inline bool stem_shape(const SiConv2dDesc* d, StemShape& s) {
    if (d->groups != 1 || d->dh != 1 || d->dw != 1 || d->ic != 3 || d->sh != 2 || d->sw != 2) return false;
    if (d->oc < 1 || d->oc > 64) return false;
    s.nt = d->oc > 32 ? 2 : 1;
    s.kh = d->kh; s.kw = d->kw;
    return (d->kh == 6 && d->kw == 6) || (d->kh == 7 && d->kw == 7) || (d->kh == 3 && d->kw == 3);
}

inline int stem_steps(const StemShape& s) { return (s.kh * s.kw * 3 + 3) / 4; }

template <int NW, int PB, int NT, int KH, int KW, typename OutT, int EP = 0>
int launch_roll(StemArgs a, bool vec, hipStream_t st) {
    constexpr int S = 2, C = 3, KWC = KW * C, K = KH * KWC, STEPS = (K + 3) / 4, NR = KH + S, WLD = 32 * NT + (((SI_STEM_EXP & 2) && NT == 2) ? 16 : 0), TOW = 16 * PB * NW;
    a.w_tiles = (a.ow + TOW - 1) / TOW;
    a.row_len = (3 + (TOW - 1) * S * C + KWC + 3) / 4 * 4;
    const size_t lds = ((size_t)NR * a.row_len + (size_t)STEPS * 4 * WLD) * sizeof(float);
    if (lds > 160 * 1024) return SI_E_UNSUPPORTED;
    auto kern = vec ? conv_stem_roll_kernel<NW, PB, NT, KH, KW, OutT, true, EP> : conv_stem_roll_kernel<NW, PB, NT, KH, KW, OutT, false, EP>;
    if (hipError_t e = si_allow_dynamic_lds(kern, lds); e != hipSuccess) return (int)e;
    const int per_cu = si_resident_blocks(kern, NW * 64, lds);
    const long long slots = 256LL * per_cu;
    // runs of output rows: enough tasks to fill every resident workgroup once (the run's first KH - S rows are the only input
    // rows fetched twice), runs of at least 2 rows (small batches: parallelism matters more than the re-fetch)
    const long long base = (long long)a.n * a.w_tiles;
    long long chunks = (slots + base - 1) / base;
    if (chunks < 1) chunks = 1;
    int rc = (int)((a.oh + chunks - 1) / chunks);
    if (rc < 2) rc = a.oh < 2 ? a.oh : 2;
    a.rc = rc;
    a.row_chunks = (a.oh + rc - 1) / rc;
    const long long tasks = base * a.row_chunks;
    if (tasks > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    a.tasks = (int)tasks;
    const int grid = (int)(tasks < slots ? tasks : slots);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace

// ---- the interface conv_smallc.hip dispatches through (shape-only: it also fixes the weight layout) ----------------------------
bool si_conv_stemroll_ok(const SiConv2dDesc* d) {
    StemShape s;
    return stem_shape(d, s);
}

size_t si_conv_stemroll_weight_elems(const SiConv2dDesc* d) {
    StemShape s;
    if (!stem_shape(d, s)) return 0;
    return (size_t)stem_steps(s) * 4 * (32 * s.nt);
}

// OIHW -> [k = (ky, kx, c)][oc padded to 32 NT], zero filled
void si_conv_stemroll_pack(const SiConv2dDesc* d, const float* w_oihw, float* w_packed) {
    StemShape s;
    if (!stem_shape(d, s)) return;
    const int wld = 32 * s.nt;
    const size_t total = si_conv_stemroll_weight_elems(d);
    for (size_t i = 0; i < total; ++i) w_packed[i] = 0.0f;
    for (int o = 0; o < d->oc; ++o)
        for (int c = 0; c < 3; ++c)
            for (int ky = 0; ky < d->kh; ++ky)
                for (int kx = 0; kx < d->kw; ++kx)
                    w_packed[((size_t)(ky * d->kw + kx) * 3 + c) * wld + o] = w_oihw[(((size_t)o * 3 + c) * d->kh + ky) * d->kw + kx];
}

// 320-pixel column tiles (5 blocks of 16 per wave) when that wastes fewer pixel blocks than 128-pixel tiles (2 per wave)
static bool stem_wide(const SiConv2dDesc* d) {
    const long long w320 = (d->ow + 319) / 320 * 320, w128 = (d->ow + 127) / 128 * 128;
    return d->kh == 6 && d->oc <= 32 && w320 <= w128;
}

const char* si_conv_stemroll_name(const SiConv2dDesc* d) {
    StemShape s;
    if (!stem_shape(d, s)) return "invalid";
    if (s.kh == 6) return stem_wide(d) ? "conv_stem_roll_kernel<4, 5, 1, 6, 6>"
                                       : (s.nt == 1 ? "conv_stem_roll_kernel<4, 2, 1, 6, 6>" : "conv_stem_roll_kernel<4, 2, 2, 6, 6>");
    if (s.kh == 7) return s.nt == 1 ? "conv_stem_roll_kernel<4, 2, 1, 7, 7>" : "conv_stem_roll_kernel<4, 2, 2, 7, 7>";
    return s.nt == 1 ? "conv_stem_roll_kernel<4, 2, 1, 3, 3>" : "conv_stem_roll_kernel<4, 2, 2, 3, 3>";
}

template <typename OutT>
static int stemroll_launch_t(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias, const float* residual,
                             void* out, hipStream_t st) {
    StemShape s;
    if (!stem_shape(d, s)) return SI_E_UNSUPPORTED;
    StemArgs a;
    a.in = in; a.w = w_packed; a.bias = d->has_bias ? bias : nullptr; a.res = d->has_residual ? residual : nullptr; a.out = out;
    a.n = d->n; a.ih = d->ih; a.iw = d->iw; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.oc = d->oc; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.pt = d->pt; a.pl = d->pl;
    a.row_len = a.rc = a.w_tiles = a.row_chunks = a.tasks = 0;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->ih * d->iw * d->in_ld * 4ull;
    if (in_bytes >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;
    a.in_bytes = (unsigned)in_bytes;
    // extent of the output as the kernel addresses it (the last pixel's row may be a view into a wider concat buffer: out_ld)
    const unsigned long long out_bytes = ((unsigned long long)d->n * d->oh * d->ow - 1ull) * d->out_ld * 4ull + (unsigned long long)d->oc * 4ull;
    a.out_bytes = (out_bytes < 0xFFFFFF00ull && d->out_ld % 4 == 0 && d->oc % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? (unsigned)out_bytes : 0u;
    const bool vec = d->in_ld == 3 && (d->iw * 3) % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    // the two stems of the path get their epilogue at compile time (YOLOv5: SiLU; ResNet: ReLU)
    const bool plain = !d->has_residual && d->act2 == SI_ACT_NONE && a.out_bytes != 0;
    if (s.kh == 6) {
        if (stem_wide(d)) {
            if (plain && d->act1 == SI_ACT_SILU) return launch_roll<4, 5, 1, 6, 6, OutT, 1>(a, vec, st);
            return launch_roll<4, 5, 1, 6, 6, OutT>(a, vec, st);
        }
        return s.nt == 1 ? launch_roll<4, 2, 1, 6, 6, OutT>(a, vec, st) : launch_roll<4, 2, 2, 6, 6, OutT>(a, vec, st);
    }
    if (s.kh == 7) {
        if (s.nt == 2 && plain && d->act1 == SI_ACT_RELU) return launch_roll<4, 2, 2, 7, 7, OutT, 2>(a, vec, st);
        return s.nt == 1 ? launch_roll<4, 2, 1, 7, 7, OutT>(a, vec, st) : launch_roll<4, 2, 2, 7, 7, OutT>(a, vec, st);
    }
    return s.nt == 1 ? launch_roll<4, 2, 1, 3, 3, OutT>(a, vec, st) : launch_roll<4, 2, 2, 3, 3, OutT>(a, vec, st);
}

int si_conv_stemroll_launch(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias, const float* residual,
                            float* out, hipStream_t st) {
    return stemroll_launch_t<float>(d, in, w_packed, bias, residual, out, st);
}
