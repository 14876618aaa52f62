// conv_wino23_split.hip -- the fused Winograd F(2x2, 3x3) kernel of conv_wino23.hip with its 16 plane GEMMs on the FP16 matrix cores by
// operand splitting (round 5, late; engine option f32_split, OPT-IN -- conv_split3.hip has the scheme and its references).
//
// Everything around the channel loop is conv_wino23.hip's 32-tile form: a workgroup owns 32 Winograd tiles x 32 output channels, the
// row half of B^T d B is formed where the patch rows are staged (fp32, channel-major planes in LDS, two buffers, one barrier per
// 16-channel block), wave r owns plane row r, and the output transform / bias / activation / shortcut epilogue is the same code on the
// same 32x32 C/D layout.  What changes is the loop: a 16-channel block is ONE 16-deep step of v_mfma_f32_32x32x16_f16 per plane.  A lane
// (tile, k half) reads the four row-r pixels of its eight channels, forms V[r][0..3] for each (the column half, fp32), splits every
// value v = hi + 2^-11 lo into two fp16 halves (22 significant bits) and feeds, per plane column, three MFMAs -- hi x U_hi into one
// accumulator set, hi x U_lo and lo x U_hi into a second one (the two scales meet once, in front of the output transform).  The filter
// image U = G g G^T is split the same way at load: two lane-order fp16 images [plane row][cb][oc tile][plane column][64 lanes][8].
// Per block and wave: 12 MFMAs (384 cycles) behind ~200 vector instructions of transform + split -- the fp32 loop it replaces is 32
// MFMAs of 64 cycles.  fp32 tensors in and out; another arithmetic than conv_wino23.hip (not bit-compatible with it), measured against
// the oracle's reference pipeline and the float64 convolution at the fp32 bars (tests/test_gpu_ops.py).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "si_hip.h"
#include "si_hip_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct WinoArgs {
    const float* in;
    const _Float16* u;   // U_hi then U_lo (x 2^11): each [4 plane rows][ic/16][oc/32][4 plane columns][64 lanes][8 halves]
    const float* bias;
    const float* res;
    float* out;
    int n, ih, iw, ic, in_ld;
    int oh, ow, oc, out_ld, res_ld;
    int pad;
    int th, tw;          // tiles per image column / row
    int rows_total;      // n * th
    int col_blocks, oc_blocks, spatial_blocks;
    unsigned mg_chunk, mg_cols, mg_th;   // floor(2^32 / d) of the block decode's three divisors (wino_div)
    unsigned in_bytes, u_bytes;
    int act1, act2;
    float act_param;
    int vec_out;         // out (and res) rows are 16-byte aligned: float4 stores
    unsigned* range_flag; // SiConv2dDesc::range_flag: set to 1 when an accumulator left the matrix cores non-finite (a V value overflowed fp16)
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
// SI_WS_ABL (diagnostic builds only, tools/hip_variant.sh): bit 0 no MFMAs, 1 no transform / split (constant operands), 2 no staging stores
// after block 0, 3 no filter loads after block 0, 4 no patch fetches after block 1, 5 no LDS reads -- wrong results, timing only
#ifndef SI_WS_ABL
#define SI_WS_ABL 0
#endif

// n / d for 0 <= n < 2^32 with mg = floor(2^32 / d) (0xFFFFFFFF for d = 1): the high product is the quotient or one less
__device__ __forceinline__ int wino_div(int n, int d, unsigned mg) {
    unsigned q = __umulhi((unsigned)n, mg);
    if ((unsigned)n - q * (unsigned)d >= (unsigned)d) ++q;
    return (int)q;
}

#ifndef SI_WINO_ABLATE   // diagnostic builds only (timing experiments, wrong results): 1 no patch loads after block 0,
#define SI_WINO_ABLATE 0  // 2 no filter loads after the first, 4 no staging stores after block 0, 8 no output stores
#endif

#define SI_WINO_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifndef SI_WINO16_WAVES   // waves per SIMD the 16-tile form is compiled for (experiments: 4 needs <= 128 registers and <= 40 KB of LDS)
#define SI_WINO16_WAVES 3
#endif

// LOG_TBW: log2 of tiles per block row (the workgroup's 32 tiles form a TBH x TBW block).  OCG: 32-channel output groups per
// workgroup (1 or 2): with two, every transformed input value feeds eight MFMAs instead of four and the patches are fetched
// half as often, at the price of 128 accumulator registers (two waves per SIMD instead of three).
// MT: the MFMA tile.  32: v_mfma_f32_32x32x2_f32, a workgroup = 32 tiles x 32 * OCG output channels, 16-channel blocks of 8 steps
// (2 channels each).  16 (round 3): v_mfma_f32_16x16x4_f32, a workgroup = 16 tiles x 32 output channels (two 16-wide halves),
// 32-channel blocks of 8 steps (4 channels each) -- HALF-SIZE work units for launches whose 32-tile units do not fill the chip's
// residency slots evenly or do not fill it at all (small batches).  Same transforms, same ascending channel order
// in one fma chain per output, so the two forms produce the same bits (tests/test_gpu_ops.py).
template <int LOG_TBW, int OCG, int MT = 32>
__global__ __launch_bounds__(256, 2) void conv_wino23s_kernel(const WinoArgs a) {
    static_assert(MT == 32 && OCG == 1, "the split form exists for 32-tile units of one 32-channel output group");
    constexpr int TILES = MT;                // tiles per workgroup
    constexpr int CBX = MT == 32 ? CB : 32;  // input channels per staged block
    constexpr int LOG_CQ = MT == 32 ? 2 : 3; // log2 of the block's 4-channel vectors
    constexpr int CQ = 1 << LOG_CQ;
    constexpr int CPS = MT == 32 ? 2 : 4;    // channels per step (the MFMA's k)
    constexpr int NH = MT == 32 ? 1 : 2;     // MFMA column tiles per 32 output channels
    constexpr int TBW = 1 << LOG_TBW;
    constexpr int TBH = TILES / TBW;
    constexpr int PW = 2 * TBW + 2;          // staged pixels per tile row: 2 * TBW of its own and two of halo
    constexpr int PWP = PW;                  // even row pitch: every tile's four pixels start 8-byte aligned (ds_read_b64)
    constexpr int SLOTS = 4 * TBH;           // slot = r * TBH + tr  (r = plane row 0..3)
    // floats per staged channel, padded to 2 (mod 8): the staging stores of a 16-byte vector's four channels go to four
    // planes (4*PLANE apart = 8 banks apart), conflict free within a 32-lane group
    constexpr int PLANE = ((SLOTS * PWP + 5) / 8) * 8 + 2;
    static_assert(PLANE % 2 == 0, "8-byte aligned tile pixels");
    constexpr int BUF = CBX * PLANE;         // one staged channel block; two alternate, so a block costs ONE barrier
    constexpr int OCW = 32 * OCG;            // output channels per workgroup
    constexpr int XCH = 4 * 2 * TILES * OCW; // exchange: [plane row][output column][tile][oc]
    // filter values are requested RING steps before their step (32x32 form: 1024 MFMA cycles; the 16x16 form's two loads per step make
    // a 4-deep ring 32 registers -- with 2 it fits three waves per SIMD without spilling)
    constexpr int RING = (OCG == 1 && MT == 32) ? 4 : 2;
    constexpr int LDS_FLOATS = 2 * BUF > XCH ? 2 * BUF : XCH;
    constexpr int NHALO = TBH * 2 * CQ * 4;  // halo loads: (tile row, pixel 2*TBW or 2*TBW+1, 4 channels, patch row)
    constexpr int HR = (NHALO + 255) / 256;  // ... per thread

    __shared__ __attribute__((aligned(16))) float patch[LDS_FLOATS];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & (MT - 1), lh = lane / MT;   // operand row (tile / oc) and k slot of the lane
    // block id -> (spatial block, oc block): the oc blocks of one spatial block share blockIdx % 8, i.e. one XCD and its
    // L2, because they all stage the same input patches (placement is a speed hint only)
    const int per_chunk = 8 * a.oc_blocks;
    const int chunk = wino_div((int)blockIdx.x, per_chunk, a.mg_chunk);
    const int rr = blockIdx.x - chunk * per_chunk;
    const int sb = chunk * 8 + (rr & 7);
    const int ocb = rr >> 3;
    if (sb >= a.spatial_blocks) return;
    const int by = wino_div(sb, a.col_blocks, a.mg_cols);
    const int bc = sb - by * a.col_blocks;
    const int row0 = by * TBH;               // first flattened tile row of the block
    const int col0 = bc * TBW;               // first tile column
    const int oc0 = ocb * OCW;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const unsigned row_pitch = (unsigned)(a.iw * a.in_ld * 4);

    // ---- staging items.  Core: thread -> (tile row, one of the 2*TBW own pixels, 4 channels); halo: the first NHALO threads
    // -> (tile row, one of the two halo pixels, 4 channels).  An item is the byte offset of patch row 0 (it may lie before
    // the tensor when that row is padding; only valid rows use it) and a 4-bit mask of the rows inside the image.
    auto make_item = [&](int tr, int px, int cq, unsigned& base, unsigned& rows) {
        const int img = wino_div(row0 + tr, a.th, a.mg_th);   // (image, tile row) of the item's flattened tile row, branch free
        const int ty = row0 + tr - img * a.th;
        const int y0 = 2 * ty - a.pad, x = 2 * col0 - a.pad + px;
        const bool ok = row0 + tr < a.rows_total && (unsigned)x < (unsigned)a.iw;
        rows = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) rows |= (ok && (unsigned)(y0 + j) < (unsigned)a.ih) ? (1u << j) : 0u;
        base = (unsigned)((img * a.ih + y0) * a.iw + x) * (unsigned)(a.in_ld * 4) + (unsigned)(cq * 16);
    };
    const int c_cq = tid & (CQ - 1), c_px = (tid >> LOG_CQ) & (2 * TBW - 1), c_tr = tid >> (LOG_TBW + 1 + LOG_CQ);
    unsigned c_base, c_rows;
    make_item(c_tr, c_px, c_cq, c_base, c_rows);
    c_base += row_pitch;                     // offset of patch row 1 (see fetch)
    // LDS destination (floats) of channel 0 / plane row 0 of the item
    const int c_dst = (c_cq * 4) * PLANE + c_tr * PWP + c_px;
    // Halo: a QUAD of lanes shares one item, lane j of it fetching patch row j alone; the row transform then takes its two
    // rows from the quad by DPP (4 staging registers instead of 16 for a column only one thread in eight would own).
    const int h_j = tid & 3;
    const float h_sign = h_j == 1 ? 1.0f : -1.0f;
    unsigned h_off[HR];
    int h_dst[HR];
    bool h_live[HR];
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int hv = tid + 256 * i;
        const int h_cq = (hv >> 2) & (CQ - 1), h_px = 2 * TBW + ((hv >> (2 + LOG_CQ)) & 1), h_tr = (hv >> (3 + LOG_CQ)) & (TBH - 1);
        unsigned base, rows;
        make_item(h_tr, h_px, h_cq, base, rows);
        h_live[i] = hv < NHALO;
        h_off[i] = (h_live[i] && ((rows >> h_j) & 1u)) ? base + (unsigned)h_j * row_pitch : 0xFFFFFF00u;
        h_dst[i] = (h_cq * 4) * PLANE + (h_j * TBH + h_tr) * PWP + h_px;   // plane row j is the one this lane produces
    }

    u32x4 cpre[4], hpre[HR];
    // (the patch row and the channel block ride in the instruction's SCALAR offset, which the range check does not see: a lane
    // costs one select per load -- its item's offset, or the out-of-range one when that row is masked)
    // Per-lane load offsets, fixed for the workgroup: patch row j of the core item (row 0 is one pitch below row 1, whose offset
    // is never before the tensor: pad <= 1 -- the hardware adds the scalar offset in 64 bits, so a wrapped "negative" lane offset
    // would not come back), or the out-of-range offset where that row is padding.  The patch row (rows 1..3) and the channel block
    // ride in the SCALAR offset, which the range check does not see; a fetch past the last block goes through a descriptor of
    // zero records (`live` false), so every staging load is issued unconditionally and costs no vector instruction.
    unsigned c_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c_off[j] = ((c_rows >> j) & 1u) ? (j == 0 ? c_base - row_pitch : c_base) : 0xFFFFFF00u;
    const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, 0, 0x00020000);
    auto fetch = [&](u32x4 (&dst)[4], int cb, bool live) {
        const __amdgpu_buffer_rsrc_t rs = live ? rs_in : rs_none;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, c_off[j], (unsigned)(j == 0 ? 0 : j - 1) * row_pitch + (unsigned)(cb * CBX * 4), 0);
    };
    // the row half of B^T d B (winograd_helper.cpp:188-239):  r=0: d0 - d2   r=1: d1 + d2   r=2: d2 - d1   r=3: d1 - d3,
    // computed where it is stored (rows r_lo..r_hi-1 of one item: 4 channels to 4 planes each)
    auto store_rows = [&](const u32x4 (&p)[4], int dst, auto r_lo, auto r_hi) {
        const f32x4 d0 = __builtin_bit_cast(f32x4, p[0]), d1 = __builtin_bit_cast(f32x4, p[1]);
        const f32x4 d2 = __builtin_bit_cast(f32x4, p[2]), d3 = __builtin_bit_cast(f32x4, p[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < decltype(r_lo)::value || r >= decltype(r_hi)::value) continue;
            const f32x4 f = r == 0 ? d0 - d2 : (r == 1 ? d1 + d2 : (r == 2 ? d2 - d1 : d1 - d3));
#pragma unroll
            for (int k = 0; k < 4; ++k) patch[dst + k * PLANE + r * TBH * PWP] = f[k];
        }
    };
    auto fetch_halo = [&](int cb, bool live) {
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            hpre[i] = __builtin_amdgcn_raw_buffer_load_b128(live ? rs_in : rs_none, h_off[i], (unsigned)(cb * CBX * 4), 0);
        }
    };
    // lane j of a quad:  t_j = d[ja] +- d[jb]  with (ja, jb) = (0,2) (1,2) (2,1) (1,3): two quad permutes and one fma per value
    auto store_halo = [&](int buf_off) {
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            if (wave * 64 + 256 * i >= NHALO) continue;   // (wave-uniform: a wave without halo lanes skips the permutes too)
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int d = (int)hpre[i][k];
                const float da = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(d, 0x64, 0xF, 0xF, false));   // quad_perm [0,1,2,1]
                const float db = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(d, 0xDA, 0xF, 0xF, false));   // quad_perm [2,2,1,3]
                t[k] = da + h_sign * db;
            }
            if (h_live[i]) {
#pragma unroll
                for (int k = 0; k < 4; ++k) patch[buf_off + h_dst[i] + k * PLANE] = t[k];
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;

    // ---- this wave's plane row r = wave; its lane's tile and its eight channels of a block (8 lh .. 8 lh + 7)
    const int tr_l = l31 >> LOG_TBW, tc_l = l31 & (TBW - 1);
    const int rd0 = (8 * lh) * PLANE + (wave * TBH + tr_l) * PWP + 2 * tc_l;
    int rd1 = rd0 + 2;
    // two ds_read_b64 (2 LDS cycles each); merged into one ds_read2_b64 they would take 8
    asm volatile("" : "+v"(rd1));
    rd1 &= ~1;                               // (still 8-byte aligned, which the compiler can no longer see)

    // B operand: the split filter images, [plane row][cb][oc tile][plane column q][64 lanes][8 halves]: lane (o, h) of (row, cb, oc tile, q)
    // holds U[4 row + q][cb * 16 + 8 h .. + 7][oc tile * 32 + o] -- one coalesced 16-byte load per lane, plane column and image
    const int noct = a.oc / 32;
    const int ncb = a.ic / CBX;
    const unsigned img_bytes = (unsigned)(16 * a.ic) * (unsigned)a.oc * 2u;
    const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.u), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.u) + (size_t)16 * a.ic * a.oc, 0, img_bytes, 0x00020000);
    const unsigned u_lane = (unsigned)lane * 16u;
    const unsigned u_row = (unsigned)(wave * ncb * noct + oc0 / 32) * 4096u;   // (row, cb 0, oc tile, q 0)
    const unsigned u_cb = (unsigned)noct * 4096u;

    f32x16 acc_h[4], acc_x[4];   // per plane column: hi x hi; hi x lo + lo x hi (scaled by 2^11)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_h[q][e] = acc_x[q][e] = 0.0f;
    f16x8 bh[4], bl[4];
    auto load_bq = [&](int cb, int q) {
        const unsigned so = u_row + (unsigned)cb * u_cb + (unsigned)q * 1024u;
        bh[q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_hi, u_lane, so, 0));
        bl[q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_lo, u_lane, so, 0));
    };

    // ---- prologue: block 0 staged, block 1 in flight, the filter fragments of block 0 requested
    fetch(cpre, 0, true);
    fetch_halo(0, true);
#pragma unroll
    for (int q = 0; q < 4; ++q) load_bq(0, q);
    store_rows(cpre, c_dst, I0{}, I4{});
    store_halo(0);
    fetch(cpre, 1, ncb > 1);
    fetch_halo(1, ncb > 1);
    __syncthreads();

    // ---- channel loop: one 16-channel block = one 16-deep MFMA step per plane column.  The lane's eight channels: the four row-r
    // pixels of each, the column half of B^T d B (V[r][0..3] = t0 - t2, t1 + t2, t2 - t1, t1 - t3), the split, three MFMAs per column.
    constexpr float kLo = 2048.0f;   // 2^11
    auto block = [&](int cb, auto more_t) {
        constexpr bool more = decltype(more_t)::value;
        const int buf = cb & 1, nbuf = buf ^ 1;
        const int p0 = rd0 + buf * BUF, p1 = rd1 + buf * BUF;
        float v[4][8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float2 ta = {1.0f, 2.0f}, tb = {3.0f, 4.0f};
            if (!(SI_WS_ABL & 32)) {
                ta = *reinterpret_cast<const float2*>(patch + p0 + c * PLANE);
                tb = *reinterpret_cast<const float2*>(patch + p1 + c * PLANE);
            }
            v[0][c] = ta.x - tb.x;
            v[1][c] = ta.y + tb.x;
            v[2][c] = tb.x - ta.y;
            v[3][c] = ta.y - tb.y;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f16x8 hi, lo;
            if (SI_WS_ABL & 2) {
                hi = __builtin_bit_cast(f16x8, bh[q]);
                lo = __builtin_bit_cast(f16x8, bl[q]);
                asm volatile("" ::"v"(v[q][0]), "v"(v[q][7]));
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const half_t hv = (half_t)v[q][c];
                    hi[c] = hv;
                    lo[c] = (half_t)((v[q][c] - (float)hv) * kLo);
                }
            }
            if (!(SI_WS_ABL & 1)) {
                acc_h[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, bh[q], acc_h[q], 0, 0, 0);
                acc_x[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, bl[q], acc_x[q], 0, 0, 0);
                acc_x[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, bh[q], acc_x[q], 0, 0, 0);
            } else {
                asm volatile("" ::"v"(hi), "v"(lo));
            }
            // this column's filter fragments are free as soon as its MFMAs have issued: the next block's are requested here, the rest of
            // the block ahead of their use (requested behind the block they would wait for a whole L2 round trip in front of every block)
            if (more && !(SI_WS_ABL & 8)) load_bq(cb + 1, q);
            SI_WINO_FENCE();   // (left alone hipcc sinks these requests behind the last MFMA of the block)
        }
        if (more) {
            // block cb + 1 (in the staging registers since the last block) into the other buffer; block cb + 2 requested; the block's one
            // barrier (every wave has read this block's buffer above)
            if (!(SI_WS_ABL & 4)) {
                store_rows(cpre, nbuf * BUF + c_dst, I0{}, I4{});
                store_halo(nbuf * BUF);
            }
            const bool live = cb + 2 < ncb;
            if (!(SI_WS_ABL & 16)) {
                fetch(cpre, cb + 2, live);
                fetch_halo(cb + 2, live);
            }
            __syncthreads();
        }
    };
    for (int cb = 0; cb + 1 < ncb; ++cb) block(cb, std::true_type{});
    block(ncb - 1, std::false_type{});

    // ---- output transform.  Column half in registers (winograd_helper.cpp:582-590): Z0 = m0+m1+m2, Z1 = m1-m2-m3, parked in LDS
    // as [plane row][output column jc][tile][oc] (scalar stores of 32 consecutive floats per half wave).  After ONE barrier
    // wave w finishes output row i = w & 1 of output column jc = w >> 1: a lane takes four channels of TPI tiles per pass,
    // reads the four plane rows as float4s and applies the row half (:592-615), Y0 = Z(r0)+Z(r1)+Z(r2), Y1 = Z(r1)-Z(r2)-Z(r3),
    // then bias / activation / residual.
    constexpr int QPT = OCW / 4;              // float4s per tile
    constexpr int TPI = 64 / QPT;             // tiles per pass
    const int quad = lane & (QPT - 1);
    f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + oc0 + 4 * quad);
    __syncthreads();                          // every wave has read its last pixels: the patch buffers become the exchange
    float* xz = patch;
    {
        float* mine = xz + (wave * 2 * TILES + 4 * lh) * OCW + l31;
        // range guard (conv_split3.hip split_range_report; include/si_hip.h): a transformed input value V = B^T d B that rounds to fp16 infinity
        // leaves every accumulator it feeds Inf / NaN -- tested here, where the two scales meet, in front of the output transform and the epilogue
        bool bad = false;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            // the two scales meet; C/D map: row (tile inside the block) = (e&3) + 8*(e>>2) + 4*lh, column = oc % 32
            const int m = (e & 3) + 8 * (e >> 2);
            float mq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mq[q] = acc_h[q][e] + acc_x[q][e] * (1.0f / kLo);
                bad |= !__builtin_isfinite(mq[q]);
            }
            mine[m * OCW] = (mq[0] + mq[1]) + mq[2];
            mine[(TILES + m) * OCW] = (mq[1] - mq[2]) - mq[3];
            if ((e & 3) == 3) SI_WINO_FENCE();
        }
        if (a.range_flag && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) *reinterpret_cast<volatile unsigned*>(a.range_flag) = 1u;
    }
    __syncthreads();
    const int i_out = wave & 1, jc = wave >> 1;
    // activation / residual combination resolved once per workgroup: the loop body is straight-line code
    auto finish = [&](auto act1, auto act2, auto has_res, auto vec) {
#pragma clang fp contract(off)  // every instantiation must round alike (bit-exact batch sharding)
#pragma unroll
        for (int i4 = 0; i4 < TILES / TPI; ++i4) {
            const int t = lane / QPT + TPI * i4;
            f32x4 zr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) zr[r] = *reinterpret_cast<const f32x4*>(xz + ((r * 2 + jc) * TILES + t) * OCW + 4 * quad);
            const f32x4 y = (i_out == 0) ? (zr[0] + zr[1]) + zr[2] : (zr[1] - zr[2]) - zr[3];
            const int tr = t >> LOG_TBW, tc = t & (TBW - 1);
            const int txg = col0 + tc;
            if (row0 + tr >= a.rows_total || txg >= a.tw) continue;
            const int img = wino_div(row0 + tr, a.th, a.mg_th);
            const int ty = row0 + tr - img * a.th;
            const int oy = 2 * ty + i_out, ox = 2 * txg + jc;
            if (oy >= a.oh || ox >= a.ow) continue;
            const size_t pix = (size_t)(img * a.oh + oy) * a.ow + ox;
            f32x4 rv = {0.0f, 0.0f, 0.0f, 0.0f};
            if (decltype(has_res)::value) {
                const float* rp = a.res + pix * a.res_ld + oc0 + 4 * quad;
                if (decltype(vec)::value) rv = *reinterpret_cast<const f32x4*>(rp);
                else rv = f32x4{rp[0], rp[1], rp[2], rp[3]};
            }
            f32x4 o4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float vv = y[k] + bv[k];
                vv = decltype(act1)::value < 0 ? wino_act(a.act1, vv, a.act_param)
                                               : (decltype(act1)::value == SI_ACT_SILU ? vv * __builtin_amdgcn_rcpf(1.0f + __expf(-vv))
                                                  : (decltype(act1)::value == SI_ACT_RELU ? fmaxf(vv, 0.0f) : vv));
                if (decltype(has_res)::value) vv += rv[k];
                vv = decltype(act2)::value < 0 ? wino_act(a.act2, vv, a.act_param)
                                               : (decltype(act2)::value == SI_ACT_RELU ? fmaxf(vv, 0.0f) : vv);
                o4[k] = vv;
            }
            float* op = a.out + pix * a.out_ld + oc0 + 4 * quad;
            if ((SI_WINO_ABLATE & 8) && o4[0] != 123.456f) continue;
            if (decltype(vec)::value) {
                *reinterpret_cast<f32x4*>(op) = o4;
            } else {
                op[0] = o4[0]; op[1] = o4[1]; op[2] = o4[2]; op[3] = o4[3];
            }
        }
    };
    using SiluT = std::integral_constant<int, SI_ACT_SILU>;
    using ReluT = std::integral_constant<int, SI_ACT_RELU>;
    using NoneT = std::integral_constant<int, SI_ACT_NONE>;
    using AnyT = std::integral_constant<int, -1>;
    const bool res = a.res != nullptr;
    if (!a.vec_out) {
        if (res) finish(AnyT{}, AnyT{}, std::true_type{}, std::false_type{});
        else finish(AnyT{}, AnyT{}, std::false_type{}, std::false_type{});
    } else if (a.act1 == SI_ACT_SILU && a.act2 == SI_ACT_NONE) {
        if (res) finish(SiluT{}, NoneT{}, std::true_type{}, std::true_type{});
        else finish(SiluT{}, NoneT{}, std::false_type{}, std::true_type{});
    } else if (a.act1 == SI_ACT_RELU && a.act2 == SI_ACT_NONE && !res) {
        finish(ReluT{}, NoneT{}, std::false_type{}, std::true_type{});
    } else if (a.act1 == SI_ACT_NONE && a.act2 == SI_ACT_RELU && res) {
        finish(NoneT{}, ReluT{}, std::true_type{}, std::true_type{});
    } else if (res) {
        finish(AnyT{}, AnyT{}, std::true_type{}, std::true_type{});
    } else {
        finish(AnyT{}, AnyT{}, std::false_type{}, std::true_type{});
    }
}


template <int LOG_TBW>
int launch_wino_split(WinoArgs a, hipStream_t s) {
    constexpr int TBW = 1 << LOG_TBW;
    constexpr int TBH = 32 / TBW;
    a.col_blocks = (a.tw + TBW - 1) / TBW;
    a.oc_blocks = a.oc / 32;
    const int row_blocks = (a.rows_total + TBH - 1) / TBH;
    a.spatial_blocks = a.col_blocks * row_blocks;
    auto magic = [](int d) { return d > 1 ? (unsigned)(0x100000000ull / (unsigned)d) : 0xFFFFFFFFu; };
    a.mg_chunk = magic(8 * a.oc_blocks);
    a.mg_cols = magic(a.col_blocks);
    a.mg_th = magic(a.th);
    const long long nblocks = (long long)((a.spatial_blocks + 7) / 8) * 8 * a.oc_blocks;
    if (nblocks > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    hipLaunchKernelGGL((conv_wino23s_kernel<LOG_TBW, 1, 32>), dim3((unsigned)nblocks, 1, 1), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

// fraction of tile slots that hold real tiles for a block shape (conv_wino23.hip wino_pick_log_tbw)
inline int wino_pick_log_tbw(int tw, int rows_total) {
    int best = 3;
    double bc = -1.0;
    for (int l = 3; l >= 1; --l) {
        const int tbw = 1 << l, tbh = 32 / tbw;
        const double cols = (double)((tw + tbw - 1) / tbw) * tbw, rows = (double)((rows_total + tbh - 1) / tbh) * tbh;
        const double c = ((double)tw * rows_total) / (cols * rows);
        if (c > bc + 1e-9) {
            bc = c;
            best = l;
        }
    }
    return best;
}

bool wino_split_ok(const SiConv2dDesc* d) {
    if (!d) return false;
    if (d->kh != 3 || d->kw != 3 || d->sh != 1 || d->sw != 1 || d->dh != 1 || d->dw != 1 || d->groups != 1) return false;
    if (d->pt != d->pl || (d->pt != 0 && d->pt != 1)) return false;
    return d->ic % CB == 0 && d->oc % 32 == 0;
}

}  // namespace

extern "C" int si_hip_conv2d_wino23_split_supported(const SiConv2dDesc* d) { return wino_split_ok(d) ? 1 : 0; }

// halves: two images (hi, then lo scaled by 2^11) of U = G g G^T
extern "C" size_t si_hip_conv2d_wino23_split_weight_elems(const SiConv2dDesc* d) { return wino_split_ok(d) ? (size_t)2 * 16 * d->ic * d->oc : 0; }

// U = G g G^T as conv_wino23.hip evaluates it (src/layer/simd/winograd_helper.cpp:86-129), every value split hi + 2^-11 lo;
// [plane row][cb][oc tile][plane column][lane = 32 (c % 16 / 8) + o % 32][c % 8]
extern "C" int si_hip_conv2d_wino23_split_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, void* u_packed) {
    if (!d || !w_oihw || !u_packed) return SI_E_BADARG;
    if (!wino_split_ok(d)) return SI_E_UNSUPPORTED;
    const int ic = d->ic, oc = d->oc;
    const int ncb = ic / 16, noct = oc / 32;
    half_t* const uh = static_cast<half_t*>(u_packed);
    half_t* const ul = uh + (size_t)16 * ic * oc;
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
            const int cb = c / 16, cl = c % 16;
            const int lane = 32 * (cl / 8) + o % 32, j = cl % 8;
            for (int p = 0; p < 16; ++p) {
                const size_t idx = ((((((size_t)(p / 4) * ncb + cb) * noct + o / 32) * 4 + p % 4) * 64) + lane) * 8 + j;
                const half_t hv = (half_t)t[p];
                // a filter value that is not finite or rounds to fp16 infinity cannot be split: the layer stays on the fp32 kernels
                if (!(__builtin_fabsf((float)hv) <= 65504.0f)) return SI_E_UNSUPPORTED;
                uh[idx] = hv;
                ul[idx] = (half_t)((t[p] - (float)hv) * 2048.0f);
            }
        }
    return 0;
}

extern "C" int si_hip_conv2d_wino23_split_f32(const SiConv2dDesc* d, const float* in, const void* u_packed, const float* bias,
                                              const float* residual, float* out, si_stream_t stream) {
    if (!d || !in || !u_packed || !out) return SI_E_BADARG;
    if (!wino_split_ok(d)) return SI_E_UNSUPPORTED;
    if (d->has_bias && !bias) return SI_E_BADARG;
    if (d->has_residual && !residual) return SI_E_BADARG;
    if (d->in_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(in) & 15) != 0) return SI_E_UNSUPPORTED;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->ih * d->iw * d->in_ld * 4ull;
    if (in_bytes >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;
    if (16ull * d->ic * d->oc * 2ull >= 0x7FFFFF00ull) return SI_E_UNSUPPORTED;
    if (d->has_bias && (reinterpret_cast<uintptr_t>(bias) & 15) != 0) return SI_E_UNSUPPORTED;
    if (d->oh != d->ih + 2 * d->pt - 2 || d->ow != d->iw + 2 * d->pl - 2) return SI_E_BADARG;

    WinoArgs a;
    a.in = in; a.u = static_cast<const _Float16*>(u_packed); a.bias = d->has_bias ? bias : nullptr;
    a.res = d->has_residual ? residual : nullptr; a.out = out;
    a.n = d->n; a.ih = d->ih; a.iw = d->iw; a.ic = d->ic; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.oc = d->oc; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.pad = d->pt;
    a.th = (d->oh + 1) / 2; a.tw = (d->ow + 1) / 2;
    a.rows_total = d->n * a.th;
    a.col_blocks = a.oc_blocks = a.spatial_blocks = 0;
    a.in_bytes = (unsigned)in_bytes;
    a.u_bytes = 0;
    a.vec_out = (reinterpret_cast<uintptr_t>(out) & 15) == 0 && d->out_ld % 4 == 0 &&
                (!d->has_residual || ((reinterpret_cast<uintptr_t>(residual) & 15) == 0 && d->res_ld % 4 == 0));
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    a.range_flag = d->range_flag;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int l = wino_pick_log_tbw(a.tw, a.rows_total);
    if (l == 3) return launch_wino_split<3>(a, s);
    if (l == 2) return launch_wino_split<2>(a, s);
    return launch_wino_split<1>(a, s);
}
