// conv_s2poly.hip -- 3x3 stride-2 (pad 1) convolution by POLYPHASE minimal filtering on the fp32 matrix cores (round 4).
//
// Replaces, for the layers it is eligible for, Conv2d::ForwardIm2Col of the reference (src/layer/conv_2d.cpp:207-283) on the six
// stride-2 "downsample" convs of YOLOv5s (1.33 ms of the 4.25 ms batch-32 step on the implicit GEMM, at 0.74 of the MFMA peak).
// fp32 MFMA and fp32 VALU have the same rate on gfx950, so the only way past the implicit GEMM is fewer multiplies:
//
//   one output dimension:   y0 = g0 d0 + g1 d1 + g2 d2,   y1 = g0 d2 + g1 d3 + g2 d4        (inputs d0..d4 at 4t-1 .. 4t+3)
//   even input phase (d1, d3) meets one tap (g1): 2 products.  Odd phase (d0, d2, d4) meets the 2-tap filter (g0, g2): F(2,2),
//   3 products instead of 4:   m2 = (d0 - d2) g0,  m3 = d2 (g0 + g2),  m4 = (d2 - d4) g2;   y0 = g1 d1 + m2 + m3,  y1 = g1 d3 + m3 - m4.
//   5 products per 2 outputs instead of 6; in two dimensions 25 per 2x2 outputs instead of 36: 1.44x fewer.
//
//   V = T d T^t (5x5 input patch -> 25 planes),  U = G g G^t (3x3 filter -> 25 planes, at load),  M_p = sum_c V_p U_p,  Y = A^t M A
//   T = [0 1 0 0 0; 0 0 0 1 0; 1 0 -1 0 0; 0 0 1 0 0; 0 0 1 0 -1]   G = [0 1 0; 0 1 0; 1 0 0; 1 0 1; 0 0 1]   A^t = [1 0 1 1 0; 0 1 0 1 -1]
//   (coefficients 0 / +-1 only: 20 subtractions per 25 transformed values, against 32 adds per 16 for Winograd F(2,3)).
//
// Nothing transformed touches HBM.  A workgroup owns 32 output tiles (2x2 pixels each, flattened over the batch) x 64 output
// channels and has TWO ROLES (cdna_hip_programming.md 5.6: loader / consumer split):
//   * producer waves 4-5: thread = (tile, 4 channels).  Loads its own 5x5 patch with 25 16-byte buffer loads (image borders =
//     out-of-range offsets -> 0, the conv's zero padding), transforms it in registers and stores the 25 planes to LDS as
//     V[plane][tile][16 channels] -- for the NEXT 16-channel block, while the consumers multiply the current one;
//   * consumer waves 0-3: wave w owns ALL 25 planes of 32 tiles x 16 output channels (200 accumulator registers on
//     v_mfma_f32_16x16x4_f32), so A^t M A, bias, activation, residual and the stores happen in its own registers -- no
//     exchange.  Per plane and 16-channel block: one coalesced 16-byte-per-lane load of the filter plane from L2 (U is packed
//     in the MFMA B-operand lane order at load time), two ds_read_b128 of V (the two 16-tile halves) and eight MFMAs; no VALU
//     in the loop.  One barrier per channel block.
// Accumulation per output: channel blocks ascending, inside a block k = j, 4+j, 8+j, 12+j (j = 0..3) -- fixed, so an image's
// result does not depend on the batch.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "si_hip.h"
#include "si_hip_internal.h"

#pragma clang fp contract(off)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct PolyArgs {
    const float* in;
    const float* u;      // U2[25][ic/16][oc/16][64 lanes][4]
    const float* bias;
    const float* res;
    float* out;
    int n, ih, iw, ic, in_ld;
    int oh, ow, oc, out_ld, res_ld;
    int th, tw;          // 2x2 output tiles per image column / row
    int tiles_total;     // n * th * tw
    int spatial_blocks, oc_blocks;
    unsigned mg_chunk, mg_tw, mg_thtw;
    unsigned in_bytes, u_bytes;
    int act1, act2;
    float act_param;
};

constexpr int OCW = 64;     // output channels per workgroup (4 consumer waves x 16)
constexpr int CB = 16;      // input channels per block
constexpr int NP = 25;      // planes
constexpr unsigned OOB = 0xFFFFFF00u;

__device__ __forceinline__ int poly_div(int n, int d, unsigned mg) {
    unsigned q = __umulhi((unsigned)n, mg);
    if ((unsigned)n - q * (unsigned)d >= (unsigned)d) ++q;
    return (int)q;
}

__device__ __forceinline__ float poly_act(int act, float v, float p) {
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

// SI_POLY_ABL (diagnostic builds only, timing experiments with wrong results): 1 the producers fetch block 0 only, 2 the consumers
// load the first filter planes only, 4 no output stores
#ifndef SI_POLY_ABL
#define SI_POLY_ABL 0
#endif
// TILES: output tiles per workgroup.  32: consumers hold 200 accumulator registers, one workgroup (4 consumer + 2 producer waves,
// 100 KB of LDS) per CU.  16: 100 accumulators, one producer wave, 50 KB -- two or three workgroups per CU cover each other's
// prologue, epilogue and barriers, at twice the filter traffic per multiply.
template <int TILES>
__global__ __launch_bounds__(256 + TILES * 4) void conv_s2poly_kernel(const PolyArgs a) {
    constexpr int NH = TILES / 16;   // 16-tile halves per consumer wave
    // V[buffer][plane][tile][16 channels]: a consumer's ds_read_b128 covers 16 tiles x 64 bytes = 1 KB contiguous, a producer's
    // ds_write_b128 of one plane 32 tiles x 64 bytes: no bank conflicts either way
    __shared__ __attribute__((aligned(16))) float V[2 * NP * TILES * CB];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int per_chunk = 8 * a.oc_blocks;
    const int chunk = poly_div((int)blockIdx.x, per_chunk, a.mg_chunk);
    const int rr = blockIdx.x - chunk * per_chunk;
    const int sb = chunk * 8 + (rr & 7);     // the oc blocks of one spatial block share blockIdx % 8 (one XCD's L2): speed hint only
    const int ocb = rr >> 3;
    if (sb >= a.spatial_blocks) return;
    const int ncb = a.ic / CB;
    const int thtw = a.th * a.tw;

    if (wave >= 4) {
        // ================================= producers =================================
        const int pt = tid - 256;
        const int ti = pt >> 2, q = pt & 3;
        const int t = sb * TILES + ti;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
        unsigned off[5][5];
        {
            const bool live = t < a.tiles_total;
            const int img = poly_div(live ? t : 0, thtw, a.mg_thtw);
            const int rem = (live ? t : 0) - img * thtw;
            const int ty = poly_div(rem, a.tw, a.mg_tw);
            const int tx = rem - ty * a.tw;
            const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
            // (modulo-2^32 arithmetic: patch row / column 0 may lie before the tensor; only valid positions are used)
            const unsigned base = (unsigned)((img * a.ih + y0) * a.iw + x0) * (unsigned)(a.in_ld * 4) + (unsigned)(q * 16);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const bool ok = live && (unsigned)(y0 + i) < (unsigned)a.ih && (unsigned)(x0 + j) < (unsigned)a.iw;
                    off[i][j] = ok ? base + (unsigned)((i * a.iw + j) * a.in_ld * 4) : OOB;
                }
        }
        f32x4 d[5][5];
        auto fetch = [&](int cb) {
            if ((SI_POLY_ABL & 1) && cb > 0) return;
            const bool more = cb < ncb;
            const unsigned so = (unsigned)((more ? cb : 0) * CB * 4);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    d[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, more ? off[i][j] : OOB, so, 0));
        };
        auto transform_store = [&](int buf) {
            // rows: T d   (r0 = d1, r1 = d3, r2 = d0 - d2, r3 = d2, r4 = d2 - d4), then columns: (T d) T^t
            float* dst = V + (size_t)buf * NP * TILES * CB + ti * CB + q * 4;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                f32x4 x[5];
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    x[j] = r == 0 ? d[1][j] : (r == 1 ? d[3][j] : (r == 2 ? d[0][j] - d[2][j] : (r == 3 ? d[2][j] : d[2][j] - d[4][j])));
                const f32x4 v0 = x[1], v1 = x[3], v2 = x[0] - x[2], v3 = x[2], v4 = x[2] - x[4];
                *reinterpret_cast<f32x4*>(dst + (5 * r + 0) * TILES * CB) = v0;
                *reinterpret_cast<f32x4*>(dst + (5 * r + 1) * TILES * CB) = v1;
                *reinterpret_cast<f32x4*>(dst + (5 * r + 2) * TILES * CB) = v2;
                *reinterpret_cast<f32x4*>(dst + (5 * r + 3) * TILES * CB) = v3;
                *reinterpret_cast<f32x4*>(dst + (5 * r + 4) * TILES * CB) = v4;
            }
        };
        fetch(0);
        transform_store(0);
        fetch(1);
        __syncthreads();
        for (int cb = 0; cb < ncb; ++cb) {
            // block cb + 1 was fetched one block ago: transform it into the buffer the consumers are NOT reading, then request
            // block cb + 2 (a whole block of MFMA time to land)
            if (cb + 1 < ncb) transform_store((cb + 1) & 1);
            fetch(cb + 2);
            __syncthreads();
        }
        return;
    }

    // ================================= consumers =================================
    const int g4 = lane >> 4, r16 = lane & 15;
    const int oc0 = ocb * OCW + wave * 16;
    const int nog = a.oc / 16;
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);
    // filter plane p, channel block cb, this wave's 16-channel output group: 1 KB contiguous, lane-major
    const unsigned u_lane = (unsigned)((oc0 / 16) * 64 + lane) * 16u;
    const unsigned u_cb = (unsigned)nog * 1024u;               // bytes per (plane, channel block)
    const unsigned u_plane = (unsigned)ncb * u_cb;             // bytes per plane
    auto load_u = [&](int p, int cb) {
        const bool live = cb < ncb;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, live ? u_lane : OOB, (unsigned)p * u_plane + (unsigned)(live ? cb : 0) * u_cb, 0));
    };

    f32x4 acc[NP][NH];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[p][h] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int RING = 5;   // filter planes in flight; divides the 25 planes of a block, so a plane's slot is the same in every block
    static_assert(NP % RING == 0, "ring slots must line up across channel blocks");
    f32x4 ub[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ub[i] = load_u(i, 0);
    float bias_v = 0.0f;
    if (a.bias && oc0 + r16 < a.oc) bias_v = a.bias[oc0 + r16];
    __syncthreads();   // block 0 is in V[0]

    const float* va = V + r16 * CB + g4 * 4;
    for (int cb = 0; cb < ncb; ++cb) {
        const float* vb = va + (size_t)(cb & 1) * NP * TILES * CB;
        // software pipeline, pinned with sched_barrier (one consumer wave per SIMD: nothing else hides an exposed wait): the A
        // fragments of plane p + 1 and the filter plane p + RING are requested BEFORE the eight MFMAs of plane p are issued
        f32x4 af[NH], nf[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) af[h] = *reinterpret_cast<const f32x4*>(vb + h * 16 * CB);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const f32x4 b = ub[p % RING];
#pragma unroll
            for (int h = 0; h < NH; ++h) nf[h] = p + 1 < NP ? *reinterpret_cast<const f32x4*>(vb + (p + 1) * TILES * CB + h * 16 * CB) : af[h];
            // the filter plane RING steps ahead (this block's, or the first planes of the next block)
            if (!(SI_POLY_ABL & 2)) ub[p % RING] = p + RING < NP ? load_u(p + RING, cb) : load_u(p + RING - NP, cb + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < NH; ++h) acc[p][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[h][j], b[j], acc[p][h], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < NH; ++h) af[h] = nf[h];
        }
        __syncthreads();   // the producers have finished block cb + 1; everybody is done reading block cb
    }

    // ---- Y = A^t M A, bias / activation / residual, stores.  16x16 C/D map: column (channel) = lane & 15, row (tile) = 4 (lane >> 4) + e
    const int oc_l = oc0 + r16;
    const bool oc_ok = oc_l < a.oc;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float c0[5], c1[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const float m0 = acc[5 * r + 0][h][e], m1 = acc[5 * r + 1][h][e], m2 = acc[5 * r + 2][h][e];
                const float m3 = acc[5 * r + 3][h][e], m4 = acc[5 * r + 4][h][e];
                c0[r] = (m0 + m2) + m3;
                c1[r] = (m1 + m3) - m4;
            }
            float y[2][2];
            y[0][0] = (c0[0] + c0[2]) + c0[3];
            y[1][0] = (c0[1] + c0[3]) - c0[4];
            y[0][1] = (c1[0] + c1[2]) + c1[3];
            y[1][1] = (c1[1] + c1[3]) - c1[4];
            const int t = sb * TILES + h * 16 + 4 * g4 + e;
            if (t >= a.tiles_total || !oc_ok) continue;
            const int img = poly_div(t, thtw, a.mg_thtw);
            const int rem = t - img * thtw;
            const int ty = poly_div(rem, a.tw, a.mg_tw);
            const int tx = rem - ty * a.tw;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int oy = 2 * ty + dy, ox = 2 * tx + dx;
                    if (oy >= a.oh || ox >= a.ow) continue;
                    const size_t pix = ((size_t)img * a.oh + oy) * a.ow + ox;
                    float v = poly_act(a.act1, y[dy][dx] + bias_v, a.act_param);
                    if (a.res) v += a.res[pix * a.res_ld + oc_l];
                    v = poly_act(a.act2, v, a.act_param);
                    if (!(SI_POLY_ABL & 4) || v == 12345.678f) a.out[pix * a.out_ld + oc_l] = v;
                }
        }
}

bool poly_eligible(const SiConv2dDesc* d) {
    if (!d || d->groups != 1 || d->kh != 3 || d->kw != 3 || d->sh != 2 || d->sw != 2 || d->dh != 1 || d->dw != 1 || d->pt != 1 || d->pl != 1) return false;
    if (d->ic % 16 != 0 || d->oc % 16 != 0 || d->ic < 16) return false;
    return true;
}

}  // namespace

extern "C" {

int si_hip_conv2d_s2poly_eligible(const SiConv2dDesc* d) { return poly_eligible(d) ? 1 : 0; }

size_t si_hip_conv2d_s2poly_weight_elems(const SiConv2dDesc* d) { return poly_eligible(d) ? (size_t)NP * d->ic * d->oc : 0; }

int si_hip_conv2d_s2poly_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, float* u) {
    if (!d || !w_oihw || !u) return SI_E_BADARG;
    if (!poly_eligible(d)) return SI_E_UNSUPPORTED;
    const int ic = d->ic, oc = d->oc, ncb = ic / CB, nog = oc / 16;
    for (int o = 0; o < oc; ++o)
        for (int c = 0; c < ic; ++c) {
            const float* g = w_oihw + ((size_t)o * ic + c) * 9;   // g[kh * 3 + kw]
            float x[5][3], uu[5][5];
            for (int k = 0; k < 3; ++k) {   // G g: rows g1, g1, g0, g0 + g2, g2
                x[0][k] = g[3 + k];
                x[1][k] = g[3 + k];
                x[2][k] = g[k];
                x[3][k] = g[k] + g[6 + k];
                x[4][k] = g[6 + k];
            }
            for (int r = 0; r < 5; ++r) {   // (G g) G^t
                uu[r][0] = x[r][1];
                uu[r][1] = x[r][1];
                uu[r][2] = x[r][0];
                uu[r][3] = x[r][0] + x[r][2];
                uu[r][4] = x[r][2];
            }
            const int cb = c / CB, cl = c % CB, g4 = cl / 4, j = cl % 4;
            const int og = o / 16, o16 = o % 16;
            for (int p = 0; p < NP; ++p)
                u[(((((size_t)p * ncb + cb) * nog + og) * 64) + (g4 * 16 + o16)) * 4 + j] = uu[p / 5][p % 5];
        }
    return 0;
}

int si_hip_conv2d_s2poly_f32(const SiConv2dDesc* d, const float* in, const float* u, const float* bias, const float* residual, float* out,
                             si_stream_t stream) {
    if (!d || !in || !u || !out) return SI_E_BADARG;
    if (!poly_eligible(d)) return SI_E_UNSUPPORTED;
    if (d->has_bias && !bias) return SI_E_BADARG;
    if (d->has_residual && !residual) return SI_E_BADARG;
    if (d->n <= 0 || d->oh <= 0 || d->ow <= 0) return SI_E_BADARG;
    if (d->in_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(in) & 15) != 0) return SI_E_UNSUPPORTED;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->ih * d->iw * d->in_ld * 4ull;
    const unsigned long long u_bytes = (unsigned long long)NP * d->ic * d->oc * 4ull;
    if (in_bytes >= 0xFFFFFF00ull || u_bytes >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;
    // 32-tile workgroups where they still give every CU several rounds, 16-tile ones below that (SI_POLY_TILES forces one)
    static const int forced = [] { const char* e = getenv("SI_POLY_TILES"); return e ? atoi(e) : 0; }();
    const long long tiles_ll = (long long)d->n * ((d->oh + 1) / 2) * ((d->ow + 1) / 2);
    const int TILES = forced == 16 || forced == 32 ? forced : 32;
    (void)tiles_ll;
    PolyArgs a;
    a.in = in; a.u = u; a.bias = d->has_bias ? bias : nullptr; a.res = d->has_residual ? residual : nullptr; a.out = out;
    a.n = d->n; a.ih = d->ih; a.iw = d->iw; a.ic = d->ic; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.oc = d->oc; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.th = (d->oh + 1) / 2; a.tw = (d->ow + 1) / 2;
    const long long tiles = (long long)d->n * a.th * a.tw;
    if (tiles > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    a.tiles_total = (int)tiles;
    a.spatial_blocks = (int)((tiles + TILES - 1) / TILES);
    a.oc_blocks = (d->oc + OCW - 1) / OCW;
    const int thtw = a.th * a.tw;
    a.mg_chunk = (unsigned)(0x100000000ull / (unsigned)(8 * a.oc_blocks));
    a.mg_tw = a.tw > 1 ? (unsigned)(0x100000000ull / (unsigned)a.tw) : 0xFFFFFFFFu;
    a.mg_thtw = thtw > 1 ? (unsigned)(0x100000000ull / (unsigned)thtw) : 0xFFFFFFFFu;
    a.in_bytes = (unsigned)in_bytes; a.u_bytes = (unsigned)u_bytes;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    const int chunks = (a.spatial_blocks + 7) / 8;
    if (TILES == 16) hipLaunchKernelGGL(conv_s2poly_kernel<16>, dim3(chunks * 8 * a.oc_blocks), dim3(320), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(conv_s2poly_kernel<32>, dim3(chunks * 8 * a.oc_blocks), dim3(384), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

}  // extern "C"
