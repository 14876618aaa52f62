// conv_wino23.hip -- 3x3 stride-1 convolution as fused Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// The reference runs these layers (11 of YOLOv5s' convs, 13 of ResNet18's) through Winograd F(2,3) on the CPU in four
// passes with two scratch tensors: input transform -> 16 GEMMs -> output transform -> bias
// (src/layer/conv_2d.cpp:382-487, src/layer/simd/winograd_helper.cpp:40-143, :413-580, :806-874).  On the MI355X the
// fp32 MFMA rate (157 TFLOP/s) is the bound for these layers, so the 2.25x cut in multiplies is worth having -- but
// only if the transformed tensors (4x the activation size) never touch HBM.  This kernel keeps everything on chip:
//
//   * a workgroup owns 32 Winograd tiles (TBH x TBW, rows of tiles counted across the whole batch) x 32 output
//     channels.  Per 16-channel block every thread fetches the four patch rows of one (tile row, pixel, 4 channels) item,
//     forms the ROW half of B^T d B in registers -- t_r = d[ja] +- d[jb] for the four plane rows r, the formulas of
//     winograd_helper.cpp:188-239 -- and stores the four results channel-major in LDS;
//   * wave r (0..3) owns plane ROW r of the 4x4 transform domain: for its tile (lane&31) and channel (2*step + lane>>5)
//     it reads the four pixels of row r with two ds_read_b64, forms V[r][0..3] in 4 VALU ops (the column half) and feeds them
//     straight into four 32x32x2 MFMAs as the A operand.  The B operand is the pre-transformed filter U = G g G^T
//     ([plane][ic][oc], oc contiguous), streamed from L2 through registers half a block ahead;
//   * the loop is software-pipelined by hand (round 2; the stamps of tools/conv_diag.py --algo wino showed one wave alone
//     reaching only 48 % of the MFMA rate with the LDS latency and the transform exposed after every four MFMAs): the
//     operands of step s+1 are transformed and the pixels of step s+2 requested between the MFMAs of step s; the staging
//     stores of the next block are spread over the MFMA gaps of steps 4 and 5, its global loads have a whole block to land,
//     and the one barrier per block sits behind three issued MFMAs;
//   * after the channel loop each wave applies the column half of A^T M A in registers and parks the result in LDS
//     transposed to [plane row][output column][tile][oc]; after one barrier wave w finishes output (row w&1, column w>>1)
//     of every tile: lane = (tile, 4 channels), so the row half, bias / activation / residual and the store work on float4s
//     and a wave stores eight 128-byte channel rows per instruction.
//
// Tile-block shape is chosen per layer so the tile grid is covered without waste: 4x8 tiles for >= 16 tiles per row
// (feature maps >= 32 wide), 8x4 for 8..15... (see wino_pick).  Tile rows are counted over (image, tile row)
// flattened, so a block may span images; each tile row stages its own 4 input rows.
#include <hip/hip_runtime.h>

#include <atomic>
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
    unsigned mg_chunk, mg_cols, mg_th;   // floor(2^32 / d) of the block decode's three divisors (wino_div)
    unsigned in_bytes, u_bytes;
    int act1, act2;
    float act_param;
    int vec_out;         // out (and res) rows are 16-byte aligned: float4 stores
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

// n / d for 0 <= n < 2^32 with mg = floor(2^32 / d) (0xFFFFFFFF for d = 1): the high product is the quotient or one less
__device__ __forceinline__ int wino_div(int n, int d, unsigned mg) {
    unsigned q = __umulhi((unsigned)n, mg);
    if ((unsigned)n - q * (unsigned)d >= (unsigned)d) ++q;
    return (int)q;
}

#ifndef SI_WINO_ABLATE   // diagnostic builds only (timing experiments, wrong results): 1 no patch loads after block 0,
#define SI_WINO_ABLATE 0  // 2 no filter loads after the first, 4 no staging stores after block 0, 8 no output stores
#endif
SI_STAMP_ARRAY(si_diag_stamps_wino);   // diagnostic build only (si_hip_internal.h)

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
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MT == 16 ? SI_WINO16_WAVES : 4 - OCG, MT == 16 ? SI_WINO16_WAVES : 4 - OCG))) void conv_wino23_kernel(const WinoArgs a) {
    static_assert(MT == 32 || (MT == 16 && OCG == 1), "the 16x16 form has one 32-channel output group");
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
    SI_STAMP_DECL;
    SI_STAMP_RT(0);
    SI_STAMP(1);

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
    // (the 16x16 form reads the SECOND filter image, laid out for its lanes: see si_hip_conv2d_wino23_pack_weight_host)
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u) + (MT == 32 ? 0 : (size_t)16 * a.ic * a.oc), 0, a.u_bytes, 0x00020000);

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

    // ---- this wave's plane row r = wave; its lane's tile and channel parity
    const int tr_l = l31 >> LOG_TBW, tc_l = l31 & (TBW - 1);
    const int rd0 = lh * PLANE + (wave * TBH + tr_l) * PWP + 2 * tc_l;
    int rd1 = rd0 + 2;
    // two ds_read_b64 (2 LDS cycles each); merged into one ds_read2_b64 they would take 8
    asm volatile("" : "+v"(rd1));
    rd1 &= ~1;                               // (still 8-byte aligned, which the compiler can no longer see)

    // B operand: filter image U2[plane row][cb][oc tile][step][oc%32][lane half][plane column]: for one (plane row, 16-channel
    // block, 32-wide oc tile, step) the 64 lanes' float4s are 1 KB contiguous; element q of lane (o, h) is
    // U[4*row + q][cb*16 + 2*step + h][o], i.e. the B values of the step's four MFMAs.  One fully coalesced dwordx4 load per
    // output group and step, addressed by a scalar offset and one VGPR shared by all of them.
    // 16x16 form: U3[plane row][cb (32 channels)][oc tile][step][oc half][lane][plane column], lane = 16 * (channel & 3) + oc % 16:
    // again 1 KB of contiguous float4s per load instruction, two loads (the two 16-wide oc halves) per step
    const int noct = a.oc / 32;
    const int ncb = a.ic / CBX;
    constexpr unsigned UBLK = 8192u * NH;    // bytes per (plane row, channel block, oc tile)
    const unsigned u_lane = MT == 32 ? (unsigned)(l31 * 2 + lh) * 16u : (unsigned)lane * 16u;
    const unsigned u_row = (unsigned)(wave * ncb * noct + oc0 / 32) * UBLK;      // (row, cb 0, first oc tile, step 0)
    const unsigned u_cb = (unsigned)noct * UBLK;

    typedef typename std::conditional<MT == 32, f32x16, f32x4>::type acc_t;
    constexpr int NG = OCG * NH;             // accumulator groups: 32-wide oc groups (MT 32) / 16-wide halves (MT 16)
    acc_t acc[NG][4];
    f32x4 ring[RING][NG];
    auto load_b = [&](int slot, int cb, int step) {
        if ((SI_WINO_ABLATE & 2) && (cb | step)) return;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            ring[slot][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rs_u, u_lane, u_row + (unsigned)cb * u_cb + (MT == 32 ? (unsigned)(g * 8 + step) : (unsigned)(step * 2 + g)) * 1024u, 0));
    };

    // ---- prologue: block 0 staged, block 1 in flight, the filter values of the first RING steps requested
    fetch(cpre, 0, true);
    fetch_halo(0, true);
#pragma unroll
    for (int i = 0; i < RING; ++i) load_b(i, 0, i);
    SI_STAMP(2);
    store_rows(cpre, c_dst, I0{}, I4{});
    store_halo(0);
    fetch(cpre, 1, ncb > 1);
    fetch_halo(1, ncb > 1);
    __syncthreads();
    SI_STAMP(3);

    // ---- channel loop.  tq: the four row-r pixels of the lane's tile for the step after next (in flight); va / vb: the A operands
    // of even / odd steps.
    float2 tq[2];
    float va[4], vb[4];
    auto read_t = [&](int o0, int o1, int s) {
        tq[0] = *reinterpret_cast<const float2*>(patch + o0 + (CPS * s) * PLANE);
        tq[1] = *reinterpret_cast<const float2*>(patch + o1 + (CPS * s) * PLANE);
    };
    auto col_transform = [&](float (&v)[4]) {   // the column half: V[r][0..3] from t0..t3
        const float t0 = tq[0].x, t1 = tq[0].y, t2 = tq[1].x, t3 = tq[1].y;
        v[0] = t0 - t2;
        v[1] = t1 + t2;
        v[2] = t2 - t1;
        v[3] = t1 - t3;
    };
    read_t(rd0, rd1, 0);
    col_transform(va);
    read_t(rd0, rd1, 1);

    // one 16-channel block: 8 steps of 4 * OCG MFMAs.  Compile-time flags: `more` -- another block follows, so this one also
    // stages it; `first` -- the accumulators start from the literal zero (no 64 * OCG register writes up front).
    auto block = [&](int cb, auto more_t, auto first_t) {
        constexpr bool more = decltype(more_t)::value;
        constexpr bool first = decltype(first_t)::value;
        const int buf = cb & 1, nbuf = buf ^ 1;
        const int p0 = rd0 + buf * BUF, p1 = rd1 + buf * BUF;
        const int n0 = rd0 + nbuf * BUF, n1 = rd1 + nbuf * BUF;
        const acc_t zero = {};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float(&vc)[4] = (s & 1) ? vb : va;
            float(&vn)[4] = (s & 1) ? va : vb;
            const int slot = s % RING;
            auto mfma = [&](int g, int q) {
                if constexpr (MT == 32) {
                    acc[g][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[q], ring[slot][g][q], (first && s == 0) ? zero : acc[g][q], 0, 0, 0);
                } else {   // plane column q of both 16-wide oc halves
#pragma unroll
                    for (int hf = 0; hf < NH; ++hf)
                        acc[hf][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc[q], ring[slot][hf][q], (first && s == 0) ? zero : acc[hf][q], 0, 0, 0);
                }
            };
            mfma(0, 0);
            SI_WINO_FENCE();
            // operands of the next step (step 0 of the next block after step 7), then the request for the one after it
            if (s < 7 || more) col_transform(vn);
            if (s < 6) read_t(p0, p1, s + 2);
            else if (more) read_t(n0, n1, s - 6);          // the barrier of step 5 has passed
            SI_WINO_FENCE();
            mfma(0, 1);
            SI_WINO_FENCE();
            // staging of block cb+1, spread over the MFMA gaps of steps 4 and 5
            if (more && !(SI_WINO_ABLATE & 4)) {
                if (s == 4) store_rows(cpre, nbuf * BUF + c_dst, I0{}, I2{});
                if (s == 5) store_rows(cpre, nbuf * BUF + c_dst, I2{}, I4{});
                SI_WINO_FENCE();
            }
            mfma(0, 2);
            SI_WINO_FENCE();
            if (more && !(SI_WINO_ABLATE & 4) && s == 5) {
                store_halo(nbuf * BUF);
                SI_WINO_FENCE();
            }
            mfma(0, 3);
            if (OCG == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) mfma(OCG - 1, q);
            }
            SI_WINO_FENCE();
            // the step's filter registers are free: request the values of step s + RING
            if (s + RING < 8) load_b(slot, cb, s + RING);
            else if (more) load_b(slot, cb + 1, s + RING - 8);
            if (s == 5 && more) {
                // the staging registers are free too: block cb+2 has a whole block to arrive; then the block's one barrier
                // (issued unconditionally -- past the last block, and for the halo of a wave that carries none, with every row
                // masked off: a masked load moves no bytes, and a load behind a branch would make the compiler's vmcnt for the
                // filter values wait for these loads too)
                if (!(SI_WINO_ABLATE & 1)) {
                    const bool live = cb + 2 < ncb;
                    fetch(cpre, cb + 2, live);
                    fetch_halo(cb + 2, live);
                }
                __syncthreads();
            }
            SI_WINO_FENCE();
        }
    };
    if (ncb == 1) {
        block(0, std::false_type{}, std::true_type{});
    } else {
        block(0, std::true_type{}, std::true_type{});
        for (int cb = 1; cb + 1 < ncb; ++cb) block(cb, std::true_type{}, std::false_type{});
        block(ncb - 1, std::false_type{}, std::false_type{});
    }
    SI_STAMP(4);

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
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int e = 0; e < (MT == 32 ? 16 : 4); ++e) {
                // C/D map: row (tile inside the block) = (e&3) + 8*(e>>2) + 4*lh (32x32) / e + 4*lh (16x16); column = oc % MT
                const int m = MT == 32 ? (e & 3) + 8 * (e >> 2) : e;
                mine[m * OCW + g * MT] = (acc[g][0][e] + acc[g][1][e]) + acc[g][2][e];
                mine[(TILES + m) * OCW + g * MT] = (acc[g][1][e] - acc[g][2][e]) - acc[g][3][e];
                if ((e & 3) == 3) SI_WINO_FENCE();   // (keeps the accumulator reads from being hoisted into 64 * OCG live VGPRs)
            }
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
    SI_STAMP(5);
    SI_STAMP_RT(6);
    SI_STAMP_FLUSH(si_diag_stamps_wino);
}

#ifdef SI_DIAG_STAMPS
}  // namespace
SI_STAMP_ACCESSORS(si_diag_stamps_wino, si_hip_diag_stamps_read_wino, si_hip_diag_stamps_clear_wino)
namespace {
#endif

template <int LOG_TBW, int OCG, int MT = 32>
int launch_wino(WinoArgs a, hipStream_t s) {
    constexpr int TBW = 1 << LOG_TBW;
    constexpr int TBH = MT / TBW;
    a.col_blocks = (a.tw + TBW - 1) / TBW;
    a.oc_blocks = a.oc / (32 * OCG);
    const int row_blocks = (a.rows_total + TBH - 1) / TBH;
    a.spatial_blocks = a.col_blocks * row_blocks;
    auto magic = [](int d) { return d > 1 ? (unsigned)(0x100000000ull / (unsigned)d) : 0xFFFFFFFFu; };
    a.mg_chunk = magic(8 * a.oc_blocks);
    a.mg_cols = magic(a.col_blocks);
    a.mg_th = magic(a.th);
    const long long nblocks = (long long)((a.spatial_blocks + 7) / 8) * 8 * a.oc_blocks;
    if (nblocks > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    dim3 grid((unsigned)nblocks, 1, 1);
#ifdef SI_DIAG_STAMPS   // residency experiments: extra dynamic LDS per workgroup
    static const int extra_lds = SI_ENV_INT("SI_WINO_EXTRA_LDS", 0);
    hipLaunchKernelGGL((conv_wino23_kernel<LOG_TBW, OCG, MT>), grid, dim3(256), (size_t)extra_lds, s, a);
#else
    hipLaunchKernelGGL((conv_wino23_kernel<LOG_TBW, OCG, MT>), grid, dim3(256), 0, s, a);
#endif
    return (int)hipGetLastError();
}

// fraction of tile slots that hold real tiles for a block shape
inline double wino_cover(int tw, int rows_total, int tbw, int tiles = 32) {
    const int tbh = tiles / tbw;
    const double cols = (double)((tw + tbw - 1) / tbw) * tbw, rows = (double)((rows_total + tbh - 1) / tbh) * tbh;
    return ((double)tw * rows_total) / (cols * rows);
}

inline int wino_pick_log_tbw(int tw, int rows_total, int tiles = 32) {
    int best = 3;
    double bc = -1.0;
    for (int l = 3; l >= 1; --l) {  // prefer wide blocks on ties (fewer staged halo columns)
        const double c = wino_cover(tw, rows_total, 1 << l, tiles);
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
    static const int min_ic = SI_ENV_INT("SI_WINO_MIN_IC", 32);  // (experiment build: override)
    return si_hip_conv2d_wino23_eligible(d) && d->ic >= min_ic;
}

// the filter image U = G g G^T in the 32x32-MFMA kernel's operand order, followed -- when the channel count allows the 16x16-MFMA
// form (ic a multiple of 32) -- by a second copy in that form's order: the form is chosen per launch (it follows the launch
// size), the weights are packed once
static bool wino_has_mt16(const SiConv2dDesc* d) { return d->ic % 32 == 0; }
extern "C" size_t si_hip_conv2d_wino23_weight_elems(const SiConv2dDesc* d) {
    return d ? (size_t)16 * d->ic * d->oc * (wino_has_mt16(d) ? 2 : 1) : 0;
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
            // U2[plane row][cb][oc tile][step][oc%32][lane half][plane column]   (see the kernel's B operand comment)
            const int cb = c / 16, cl = c % 16;
            const int step = cl / 2, h = cl % 2;
            const int ncb = ic / 16, noct = oc / 32;
            for (int q = 0; q < 16; ++q)
                u[((((((size_t)(q / 4) * ncb + cb) * noct + o / 32) * 8 + step) * 32 + o % 32) * 2 + h) * 4 + q % 4] = t[q];
            if (wino_has_mt16(d)) {
                // U3[plane row][cb (32 channels)][oc tile][step (4 channels)][oc half][lane = 16 * (channel & 3) + oc % 16][plane column]
                float* u3 = u + (size_t)16 * ic * oc;
                const int cb3 = c / 32, c3 = c % 32, step3 = c3 / 4, k3 = c3 % 4;
                const int ncb3 = ic / 32, ot = o / 32, hf = (o % 32) / 16, o16 = o % 16;
                for (int q = 0; q < 16; ++q)
                    u3[(((((((size_t)(q / 4) * ncb3 + cb3) * noct + ot) * 8 + step3) * 2 + hf) * 64) + (k3 * 16 + o16)) * 4 + q % 4] = t[q];
            }
        }
    return 0;
}

// the work-unit form comes with the call (SiConvPlan::wino23_form: 16 / 32 force one, 0 the policy); the form never changes a result
static int wino_forced_form(const SiConv2dDesc* d) {
    int v = (d && d->plan) ? d->plan->wino23_form : 0;
    if (v == 0) v = SI_ENV_INT("SI_WINO_MT", 0);
    return (v == 16 || v == 32) ? v : 0;
}

// When the 16-tile form is expected to be faster: a round model fitted to profiles/r03_wino16_sweep.txt.  Workgroups run in
// lock-step rounds over the residency slots (3 per CU; 2 for the 64-oc form, whose rounds are ~1.6x as long); a half-size unit
// costs ~0.55 of a full one (prologue, fill and the exchange epilogue are per workgroup).  The 16-tile form wins where the 32-tile
// grid sits just above a whole number of rounds (80x80x64 at batch 8: 800 workgroups for 768 slots) or leaves most of the chip
// empty (20x20x256 at batch 8: 100 workgroups); with one 32-channel block (ic = 32) a unit is all overhead and it never does.
static bool wino_use_mt16(const SiConv2dDesc* d, int tw, int rows_total) {
    if (d->ic < 64) return false;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    auto spatial = [&](int tiles) {
        const int tbw = 1 << wino_pick_log_tbw(tw, rows_total, tiles), tbh = tiles / tbw;
        return (long long)((tw + tbw - 1) / tbw) * ((rows_total + tbh - 1) / tbh);
    };
    const long long s32 = spatial(32), s16 = spatial(16);
    const bool two = d->oc % 64 == 0 && d->ic >= 256 && s32 * (d->oc / 64) >= 384;   // (the 32-tile form's own choice, below)
    const long long w32 = s32 * (two ? d->oc / 64 : d->oc / 32), slots32 = (long long)cus * (two ? 2 : 3);
    const double est32 = (double)((w32 + slots32 - 1) / slots32) * (two ? 1.6 : 1.0);
    const long long w16 = s16 * (d->oc / 32), slots16 = (long long)cus * 3;
    const double est16 = (double)((w16 + slots16 - 1) / slots16) * 0.55;
    return est16 < 0.95 * est32;
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
    const unsigned long long u_bytes = 16ull * d->ic * d->oc * 4ull;   // (one image: the kernel's descriptor starts at the image it reads)
    if (u_bytes * 2 >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;
    if (d->has_bias && (reinterpret_cast<uintptr_t>(bias) & 15) != 0) return SI_E_UNSUPPORTED;
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
    a.u_bytes = (unsigned)u_bytes;
    a.vec_out = (reinterpret_cast<uintptr_t>(out) & 15) == 0 && d->out_ld % 4 == 0 &&
                (!d->has_residual || ((reinterpret_cast<uintptr_t>(residual) & 15) == 0 && d->res_ld % 4 == 0));
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;

    hipStream_t s = static_cast<hipStream_t>(stream);
    // Half-size work units (16 tiles on the 16x16x4 MFMA; same bits): the plan may force either; the policy is wino_use_mt16
    const int mt_force = wino_forced_form(d);
    if (wino_has_mt16(d) && (mt_force == 16 || (mt_force == 0 && wino_use_mt16(d, a.tw, a.rows_total)))) {
        const int l16 = wino_pick_log_tbw(a.tw, a.rows_total, 16);
        if (l16 == 3) return launch_wino<3, 1, 16>(a, s);
        if (l16 == 2) return launch_wino<2, 1, 16>(a, s);
        return launch_wino<1, 1, 16>(a, s);
    }
    const int l = wino_pick_log_tbw(a.tw, a.rows_total);
    // Two output groups per workgroup pay where the channel loop is long and the grid still covers the chip (MI355X, sustained:
    // 20x20x256 batch 32 0.0848 -> 0.0736 ms, 14x14x256 batch 64 0.0750 -> 0.0706; at 128 channels and below, or under
    // ~1.5 workgroups per CU, one group with three waves per SIMD is faster: 40x40x128 0.0705 vs 0.0755, 80x80x64 0.0717 vs
    // 0.0778).  SiConvPlan::wino23_ocg = 1 / 2 forces either.
    const int ocg_force = (d->plan && (d->plan->wino23_ocg == 1 || d->plan->wino23_ocg == 2)) ? d->plan->wino23_ocg : SI_ENV_INT("SI_WINO_OCG", 0);
    const int tbw = 1 << l, tbh = 32 / tbw;
    const long long wgs2 = (long long)((a.tw + tbw - 1) / tbw) * ((a.rows_total + tbh - 1) / tbh) * (d->oc / 64);
    const bool two = d->oc % 64 == 0 && (ocg_force == 2 || (ocg_force == 0 && d->ic >= 256 && wgs2 >= 384));
    if (two) {
        if (l == 3) return launch_wino<3, 2>(a, s);
        if (l == 2) return launch_wino<2, 2>(a, s);
        return launch_wino<1, 2>(a, s);
    }
    if (l == 3) return launch_wino<3, 1>(a, s);
    if (l == 2) return launch_wino<2, 1>(a, s);
    return launch_wino<1, 1>(a, s);
}
