// conv_wino23.hip -- 3x3 stride-1 convolution as fused Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// The reference runs these layers (11 of YOLOv5s' convs, 13 of ResNet18's) through Winograd F(2,3) on the CPU in four
// passes with two scratch tensors: input transform -> 16 GEMMs -> output transform -> bias
// (src/layer/conv_2d.cpp:382-487, src/layer/simd/winograd_helper.cpp:40-143, :413-580, :806-874).  On the MI355X the
// fp32 MFMA rate (157 TFLOP/s) is the bound for these layers, so the 2.25x cut in multiplies is worth having -- but
// only if the transformed tensors (4x the activation size) never touch HBM.  This kernel keeps everything on chip:
//
//   * a workgroup owns 32 Winograd tiles (TBH x TBW, rows of tiles counted across the whole batch) x 32 output
//     channels.  Per 16-channel block it stages the raw input patches of its tiles in LDS (channel-major, so the MFMA
//     operand reads below are plain ds_read_b32 with immediate offsets);
//   * wave r (0..3) owns plane ROW r of the 4x4 transform domain: for its tile (lane&31) and channel (2*step + lane>>5)
//     it reads the two patch rows it needs (8 values), forms t = d[ja] +- d[jb] and V[r][0..3] in 8 VALU ops -- the
//     B^T d B formulas of winograd_helper.cpp:188-239 -- and feeds them straight into 32x32x2 MFMAs as the A operand.
//     The B operand is the pre-transformed filter U = G g G^T ([plane][ic][oc], oc contiguous: one 256-byte row pair
//     per MFMA), streamed from L2 through registers one step ahead;
//   * after the channel loop each wave applies the column half of A^T M A in registers, the four waves exchange the
//     row half through LDS (reusing the patch buffer), and the 2x2 outputs get bias / activation / residual and go
//     out as 128-byte channel rows.
//
// Tile-block shape is chosen per layer so the tile grid is covered without waste: 4x8 tiles for >= 16 tiles per row
// (feature maps >= 32 wide), 8x4 for 8..15... (see wino_pick).  Tile rows are counted over (image, tile row)
// flattened, so a block may span images; each tile row stages its own 4 input rows.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "si_hip.h"
#include "si_hip_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct WinoArgs {
    const float* in;
    const float* u;      // U2[16][ic/16][oc/32][2][32][2][4]
    const float* bias;
    const float* res;
    float* out;
    int n, ih, iw, ic, in_ld;
    int oh, ow, oc, out_ld, res_ld;
    int pad;
    int th, tw;          // tiles per image column / row
    int rows_total;      // n * th
    int col_blocks, oc_blocks, spatial_blocks;
    unsigned in_bytes;
    int act1, act2;
    float act_param;
};

__device__ __forceinline__ float wino_act(int act, float v, float p) {
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

constexpr int CB = 16;  // input channels per staged block

#ifndef SI_WINO_ABLATE   // diagnostic builds only (timing experiments, wrong results): 1 no patch prefetch after block 0,
#define SI_WINO_ABLATE 0  // 2 no filter loads after the first, 4 no commit writes after block 0, 8 no output stores
#endif
SI_STAMP_ARRAY(si_diag_stamps_wino);   // diagnostic build only (si_hip_internal.h)

// LOG_TBW: log2 of tiles per block row (the workgroup's 32 tiles form a TBH x TBW block).
template <int LOG_TBW>
__global__ __launch_bounds__(256) void conv_wino23_kernel(const WinoArgs a) {
    constexpr int TBW = 1 << LOG_TBW;
    constexpr int TBH = 32 / TBW;
    constexpr int PW = 2 * TBW + 2;          // staged pixels per slot row
    constexpr int PWP = PW;                  // even row pitch: every patch row of a tile starts 8-byte aligned (ds_read_b64)
    constexpr int SLOTS = 4 * TBH;           // slot = j * TBH + tr  (j = patch row 0..3)
    // floats per staged channel, padded to 2 (mod 8): the commit writes the four channels of a 16-byte vector to four
    // planes (4*PLANE apart = 8 banks apart), conflict free within a 32-lane group
    constexpr int PLANE = ((SLOTS * PWP + 5) / 8) * 8 + 2;
    constexpr int NVEC = SLOTS * PW * (CB / 4);
    constexpr int PFV = (NVEC + 255) / 256;  // 16-byte vectors per thread per block
    static_assert(PLANE % 2 == 0, "8-byte aligned patch rows");
    constexpr int BUF = CB * PLANE;          // one staged channel block; two alternate, so a block costs ONE barrier
    constexpr int XLS = 36;                  // exchange: floats per lane (32 used); XLS/4 odd keeps ds_*_b128 conflict free
    constexpr int XCH = 4 * 64 * XLS;
    constexpr int LDS_FLOATS = 2 * BUF > XCH ? 2 * BUF : XCH;

    __shared__ __attribute__((aligned(16))) float patch[LDS_FLOATS];
    SI_STAMP_DECL;
    SI_STAMP_RT(0);
    SI_STAMP(1);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    // block id -> (spatial block, oc block): the oc blocks of one spatial block share blockIdx % 8, i.e. one XCD and its
    // L2, because they all stage the same input patches (placement is a speed hint only)
    const int per_chunk = 8 * a.oc_blocks;
    const int chunk = blockIdx.x / per_chunk;
    const int rr = blockIdx.x - chunk * per_chunk;
    const int sb = chunk * 8 + (rr & 7);
    const int ocb = rr >> 3;
    if (sb >= a.spatial_blocks) return;
    const int by = sb / a.col_blocks;
    const int bc = sb - by * a.col_blocks;
    const int row0 = by * TBH;               // first flattened tile row of the block
    const int col0 = bc * TBW;               // first tile column
    const int oc0 = ocb * 32;
    // this lane's output channel and its bias, loaded up front (not at the head of the epilogue)
    const int o = oc0 + ((int)threadIdx.x & 31);
    const bool ocok = o < a.oc;
    const float bv = (a.bias && ocok) ? a.bias[o] : 0.0f;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);

    // the block's first tile row as (image, tile row): one wave-uniform division, reused by the staging slots and the stores
    const int img0 = row0 / a.th;
    const int ty0 = row0 - img0 * a.th;

    // ---- per-thread staging slots: byte offset of channel block 0 (the LDS destination is recomputed at commit time)
    unsigned g_off[PFV];
#pragma unroll
    for (int q = 0; q < PFV; ++q) {
        const int v = tid + q * 256;
        g_off[q] = 0xFFFFFF00u;
        if (v < NVEC) {
            const int cq = v & 3;
            const int rest = v >> 2;
            const int slot = rest / PW;
            const int px = rest - slot * PW;
            const int j = slot / TBH;
            const int tr = slot - j * TBH;
            const int R = row0 + tr;
            if (R < a.rows_total) {
                int img = img0, ty = ty0 + tr;  // tr < TBH <= 16: a couple of subtractions instead of a division
                while (ty >= a.th) {
                    ty -= a.th;
                    ++img;
                }
                const int y = 2 * ty - a.pad + j;
                const int x = 2 * col0 - a.pad + px;
                if ((unsigned)y < (unsigned)a.ih && (unsigned)x < (unsigned)a.iw)
                    g_off[q] = (unsigned)((img * a.ih + y) * a.iw + x) * (unsigned)(a.in_ld * 4) + (unsigned)(cq * 16);
            }
        }
    }

    u32x4 pre[PFV];
    auto prefetch = [&](int cb) {
#pragma unroll
        for (int q = 0; q < PFV; ++q) {
            const unsigned off = g_off[q] == 0xFFFFFF00u ? 0xFFFFFF00u : g_off[q] + (unsigned)(cb * CB * 4);
            pre[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < PFV; ++q) {
            const int v = tid + q * 256;
            if (v < NVEC) {
                const int rest = v >> 2;
                const int slot = rest / PW;
                const int dst = buf * BUF + ((v & 3) * 4) * PLANE + slot * PWP + (rest - slot * PW);
                const f32x4 f = __builtin_bit_cast(f32x4, pre[q]);
                patch[dst] = f[0];
                patch[dst + PLANE] = f[1];
                patch[dst + 2 * PLANE] = f[2];
                patch[dst + 3 * PLANE] = f[3];
            }
        }
    };

    // ---- this wave's plane row r = wave: t = d[ja] + sg * d[jb]   (winograd_helper.cpp:188-239)
    //   r=0: d0 - d2    r=1: d1 + d2    r=2: d2 - d1    r=3: d1 - d3
    const int ja = (wave == 0) ? 0 : ((wave == 2) ? 2 : 1);
    const int jb = (wave == 0 || wave == 1) ? 2 : ((wave == 2) ? 1 : 3);
    const float sg = (wave == 1) ? 1.0f : -1.0f;
    const int tr_l = l31 >> LOG_TBW, tc_l = l31 & (TBW - 1);
    const float* pa0 = patch + lh * PLANE + (ja * TBH + tr_l) * PWP + 2 * tc_l;
    const float* pb0 = patch + lh * PLANE + (jb * TBH + tr_l) * PWP + 2 * tc_l;

    // B operand: filter image U2[plane][cb][oc tile][half][oc%32][lane half][4]: for one (plane, 16-channel block, 32-wide
    // oc tile, half) the 64 lanes' float4s are 1 KB contiguous; element k of lane (o, h) is U[plane][cb*16 + (half*4+k)*2 + h][o],
    // i.e. the B value of MFMA step s = half*4 + k.  One fully coalesced dwordx4 load feeds four MFMAs.
    const int noct = a.oc / 32;
    const int ncb = a.ic / CB;
    const f32x4* ub = reinterpret_cast<const f32x4*>(a.u) + l31 * 2 + lh;
    auto b_index = [&](int q, int cb, int half) -> size_t {
        return ((((size_t)(4 * wave + q) * ncb + cb) * noct + (oc0 / 32)) * 2 + half) * 64;
    };

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.0f;

    f32x4 bcur[4], bnxt[4];
    auto load_b = [&](f32x4 (&dst)[4], int cb, int half) {
        if ((SI_WINO_ABLATE & 2) && (cb | half)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = bcur[q];
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = ub[b_index(q, cb, half)];
    };

    prefetch(0);
    SI_STAMP(2);
    commit(0);
    load_b(bcur, 0, 0);
    __syncthreads();
    SI_STAMP(3);

    // software pipeline: the patch rows of step s+1 are read from LDS before the MFMAs of step s are issued, and within a
    // block the filter loads are issued BEFORE the (slower, HBM) patch prefetch so that waiting for the filter values
    // never waits for the patch (vmcnt retires in order)
    float2 da[2], db[2];
    const float *pa = pa0, *pb = pb0;
        da[0] = qa[0]; da[1] = qa[1]; db[0] = qb[0]; db[1] = qb[1];
    };
    for (int cb = 0; cb < ncb; ++cb) {
        const int buf = cb & 1;
        pa = pa0 + buf * BUF;
        pb = pb0 + buf * BUF;
    hipLaunchKernelGGL((conv_wino23_kernel<LOG_TBW>), grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

// fraction of tile slots that hold real tiles for a block shape
inline double wino_cover(int tw, int rows_total, int tbw) {
    const int tbh = 32 / tbw;
    const double cols = (double)((tw + tbw - 1) / tbw) * tbw, rows = (double)((rows_total + tbh - 1) / tbh) * tbh;
    return ((double)tw * rows_total) / (cols * rows);
}

inline int wino_pick_log_tbw(int tw, int rows_total) {
    int best = 3;
    double bc = -1.0;
    for (int l = 3; l >= 1; --l) {  // prefer wide blocks on ties (fewer staged halo columns)
        const double c = wino_cover(tw, rows_total, 1 << l);
        if (c > bc + 1e-9) {
            bc = c;
            best = l;
        }
    }
    return best;
}

}  // namespace

// shape-only eligibility (reference: Conv2d::InitWinograd, src/layer/conv_2d.cpp:182-205, plus this kernel's channel
// granularity)
extern "C" int si_hip_conv2d_wino23_eligible(const SiConv2dDesc* d) {
    if (!d) return 0;
    if (d->kh != 3 || d->kw != 3 || d->sh != 1 || d->sw != 1 || d->dh != 1 || d->dw != 1 || d->groups != 1) return 0;
    if (d->pt != d->pl || (d->pt != 0 && d->pt != 1)) return 0;
    if (d->ic % CB != 0 || d->oc % 32 != 0) return 0;
    return 1;
}

// Measured on MI355X (YOLOv5s batch 32, in-network): the fused kernel beats the implicit-GEMM kernel from 64 input
// channels up (0.125 vs 0.144 ms at 80x80x64, 0.114 vs 0.150 ms at 40x40x128, 0.105 vs 0.157 ms at 20x20x256).  At 32
// channels there are only 2 channel blocks per workgroup; the first version of this kernel lost there (0.184 vs 0.174 ms at
// 160x160x32), the current one (double-buffered staging, vectorised exchange) wins by +0.8-1.1 % of the YOLOv5s step
// (6690-6705 vs 6634-6639 img/s, same-box A/B) -- hence the threshold of 32.
extern "C" int si_hip_conv2d_wino23_preferred(const SiConv2dDesc* d) {
    static const int min_ic = [] { const char* e = getenv("SI_WINO_MIN_IC"); return e ? atoi(e) : 32; }();  // dev override
    return si_hip_conv2d_wino23_eligible(d) && d->ic >= min_ic;
}

extern "C" size_t si_hip_conv2d_wino23_weight_elems(const SiConv2dDesc* d) {
    return d ? (size_t)16 * d->ic * d->oc : 0;
}

// U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], evaluated in the order of
// src/layer/simd/winograd_helper.cpp:86-129; plane = 4*row + col
extern "C" int si_hip_conv2d_wino23_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, float* u) {
    if (!d || !w_oihw || !u) return SI_E_BADARG;
    const int ic = d->ic, oc = d->oc;
    for (int o = 0; o < oc; ++o)
        for (int c = 0; c < ic; ++c) {
            const float* g = w_oihw + ((size_t)o * ic + c) * 9;  // g[kh*3 + kw]
            float t[16];
            const float r2 = 0.5f, r4 = 0.25f;
            {
                const float a02 = g[0] + g[2];
                t[0] = g[0]; t[1] = (a02 + g[1]) * r2; t[2] = (a02 - g[1]) * r2; t[3] = g[2];
            }
            {
                const float a063 = (g[0] + g[6]) + g[3], a285 = (g[2] + g[8]) + g[5], a174 = (g[1] + g[7]) + g[4];
                t[4] = a063 * r2; t[5] = ((a063 + a285) + a174) * r4; t[6] = ((a063 + a285) - a174) * r4; t[7] = a285 * r2;
            }
            {
                const float s063 = (g[0] + g[6]) - g[3], s285 = (g[2] + g[8]) - g[5], s174 = (g[1] + g[7]) - g[4];
                t[8] = s063 * r2; t[9] = ((s063 + s285) + s174) * r4; t[10] = ((s063 + s285) - s174) * r4; t[11] = s285 * r2;
            }
            {
                const float a68 = g[6] + g[8];
                t[12] = g[6]; t[13] = (a68 + g[7]) * r2; t[14] = (a68 - g[7]) * r2; t[15] = g[8];
            }
            // U2[plane][cb][oc tile][half][oc%32][lane half][4]   (see the kernel's B operand comment)
            const int cb = c / 16, cl = c % 16;
            const int step = cl / 2, h = cl % 2;
            const int half = step / 4, k = step % 4;
            const int ncb = ic / 16, noct = oc / 32;
            for (int q = 0; q < 16; ++q)
                u[((((((size_t)q * ncb + cb) * noct + o / 32) * 2 + half) * 32 + o % 32) * 2 + h) * 4 + k] = t[q];
        }
    return 0;
}

extern "C" int si_hip_conv2d_wino23_f32(const SiConv2dDesc* d, const float* in, const float* u, const float* bias,
                                        const float* residual, float* out, si_stream_t stream) {
    if (!d || !in || !u || !out) return SI_E_BADARG;
    if (!si_hip_conv2d_wino23_eligible(d)) return SI_E_UNSUPPORTED;
    if (d->has_bias && !bias) return SI_E_BADARG;
    if (d->has_residual && !residual) return SI_E_BADARG;
    if (d->in_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(in) & 15) != 0) return SI_E_UNSUPPORTED;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->ih * d->iw * d->in_ld * 4ull;
    if (in_bytes >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;
    if (d->oh != d->ih + 2 * d->pt - 2 || d->ow != d->iw + 2 * d->pl - 2) return SI_E_BADARG;

    WinoArgs a;
    a.in = in; a.u = u; a.bias = d->has_bias ? bias : nullptr; a.res = d->has_residual ? residual : nullptr; a.out = out;
    a.n = d->n; a.ih = d->ih; a.iw = d->iw; a.ic = d->ic; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.oc = d->oc; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.pad = d->pt;
    a.th = (d->oh + 1) / 2; a.tw = (d->ow + 1) / 2;
    a.rows_total = d->n * a.th;
    a.col_blocks = a.oc_blocks = a.spatial_blocks = 0;
    a.in_bytes = (unsigned)in_bytes;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;

    hipStream_t s = static_cast<hipStream_t>(stream);
    const int l = wino_pick_log_tbw(a.tw, a.rows_total);
    if (l == 3) return launch_wino<3>(a, s);
    if (l == 2) return launch_wino<2>(a, s);
    return launch_wino<1>(a, s);
}
