// conv_slab_f16.hip -- 3x3 stride-1 pad-1 fp16 convolutions over 128 / 256 input channels as ONE-SHOT ROW SLABS (round 5).
//
// The layers this is for (YOLOv5s at batch 32: 40x40x128 -> 128 and 20x20x256 -> 256, 15.1 GFLOP each; computation of the
// reference's src/layer/conv_2d.cpp:207-283) ran 25-28 us through the implicit-GEMM tiles of conv_igemm_f16.hip: 0.2 of either
// roofline.  profiles/r04_f16_ablation.txt priced why: the im2col form moves every input line through the CU's vector memory path
// nine times (once per tap) and every workgroup pays a load -> LDS -> barrier round trip per K-tile on a grid that covers the chip
// 1.5 times.  Here the layer is cut so that it covers the chip ONCE: a workgroup owns TH full-width output rows of one image (a
// "slab": 5 x 40 or 5 x 20 pixels) x 128 output channels -- 256 workgroups at batch 32.  It requests the whole (TH + 2)-row input
// patch for ALL channel blocks at once (global -> registers -> LDS: every input line is fetched (TH + 2) / TH times, not nine),
// crosses ONE barrier, and then runs the entire K loop -- channel blocks x nine taps x four 16-deep steps -- without another: the
// nine taps read their A fragments from the patch at shifted addresses and the weights come straight from L2 in MFMA lane order
// (the "bd" image of si_hip_conv2d_f16_pack_weight_host) through a 6-slot register ring.  Waves: 1 x 4, every wave ALL of the
// slab's pixels (TM 32-pixel blocks) x its own 32 output channels, so a weight fragment is fetched once per workgroup and feeds TM
// MFMAs.  A slab is a CONTIGUOUS run of output pixels, so the epilogue needs no index arithmetic.  The MFMA takes the WEIGHTS as its
// A operand and the pixels as B (the same products summed in the same k order: the same bits), so a lane ends up with four
// consecutive channels of one pixel per register quad: bias / activation / shortcut in registers, 8-byte writes into an LDS image
// [pixel][128 channels] (over the patch, which is dead by then), and the slab leaves as whole 256-byte channel rows, 16 bytes per lane.
//
// LDS image per 64-channel block: [patch row][patch column][64 + 8 halves]; pixel pitch 144 B, row pitch = ow * 144 + 512 B, which
// makes the byte address of slab pixel p congruent to p * 144 modulo 256 across row ends -- any 16 lanes of a ds_read_b128 lane
// group (MI355X_MICROARCH.md section LDS) then hit 16 distinct 16-byte bank groups.
//
// Same k order as every other fp16 tile -- channel blocks ascending, taps (ky, kx) inside, 16-deep MFMA steps inside a tap, one
// accumulator chain per output element -- and the same epilogue expressions: the same bits (tests/test_gpu_f16.py).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "si_hip.h"
#include "si_hip_internal.h"

#pragma clang fp contract(off)

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

struct SlabArgs {
    const half_t* in;
    const half_t* wl;           // lane-order weight image
    const float* bias;
    const half_t* res;
    half_t* out;
    int ih, iw, in_ld, oh, ow, out_ld, res_ld;
    int oc;
    int wl_nb, wl_ks;
    int th;                     // output rows per slab
    int slabs_per_img, n_img, n_ocg;
    int rowp, blk_bytes;        // bytes per patch row / per channel block of the patch
    int pc, n_ppix;             // patch columns (ow + 2), patch pixels (th + 2) * pc
    unsigned mg_pc, mg_ow, mg_nocg, mg_spi;   // floor(2^32 / d) of the kernel's divisions (pc, ow, n_ocg, slabs_per_img)
    int act1, act2;
    float act_param;
    unsigned in_bytes, out_bytes, res_bytes;
    // PW (a 1x1 conv + SiLU in front, computed into the patch): its lane-order weights and bias, patch rows, x pixels / bytes per block
    const half_t* wlA;
    const float* biasA;
    int wlA_nb, wlA_ks, pr, n_xpix, x_blk;
};

constexpr unsigned OOB = 0xFFFFFF00u;
constexpr unsigned OOB_W = 0x80000000u;

__device__ __forceinline__ int fdiv(int n, int d, unsigned mg) {
    unsigned q = __umulhi((unsigned)n, mg);
    if ((unsigned)n - q * (unsigned)d >= (unsigned)d) ++q;
    return (int)q;
}

template <int ACT>
__device__ __forceinline__ float act_ct(float v) {
    if (ACT == SI_ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == SI_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v;
}

#ifndef SI_SLAB_NB
#define SI_SLAB_NB 6
#endif
// SI_SLAB_ABL (diagnostic builds only, tools/hip_variant.sh): bit 0 no patch loads, 1 no K loop, 2 no output stores, 3 no weight
// loads in the loop, 4 no fragment reads in the loop -- wrong results, timing only.  0 in the product build.
#ifndef SI_SLAB_ABL
#define SI_SLAB_ABL 0
#endif

SI_STAMP_ARRAY(si_diag_stamps_slab);   // diagnostic build only (si_hip_internal.h; tools/slab_diag.py)

// TM: 32-pixel blocks per wave (the slab has at most 32 TM pixels); NBLK: 64-channel blocks (input channels / 64); N_IT: staging
// requests per thread and channel block (the patch's 16-byte chunks / 256, rounded up: exact for the two YOLOv5s forms, an upper
// bound the host checks for the others)
// PW: the layer's input is itself the output of a 1x1 conv + SiLU over the same channel count (the C3 bottleneck's first conv): that conv
// is computed HERE, for the slab's patch pixels, straight into the patch (below) -- one launch and one tensor round trip less
// W2: 512 threads -- TWO waves per SIMD, the slab's pixel blocks split between the wave pair of a channel block ([0, (TM + 1) / 2) and the
// rest).  Every vector-bound phase (the prologue's index arithmetic, the 38-issue-cycle-per-element SiLU epilogues) then issues at twice
// the rate a lone wave gets (4 cycles per instruction alone, 2 with a partner); the MFMA pipe is shared as before.
template <int TM, int NBLK, int N_IT, int ACT1, bool HAS_RES, bool PW = false, bool W2 = false>
__global__ __launch_bounds__(W2 ? 512 : 256, W2 ? 2 : 1) void conv3x3s1_slab_f16_kernel(const SlabArgs a) {
    constexpr int NT = W2 ? 512 : 256;         // threads
    static_assert(!W2 || (TM == 7 && NBLK == 2), "the 512-thread form exists (and is tested) for 7 pixel blocks over 128 channels");
    constexpr int NB = SI_SLAB_NB;             // weight fragments in flight per wave, in k-steps
    constexpr int KS_TOT = NBLK * 36;
    static_assert(36 % NB == 0, "ring depth must divide the k-steps of a channel block");
    static_assert(2 * N_IT + 1 <= 26, "the next block's requests are issued at the odd steps in front of its commit");
    static_assert(!PW || ACT1 == SI_ACT_SILU, "the fused 1x1 form is the YOLOv5 bottleneck");
    extern __shared__ __attribute__((aligned(16))) unsigned char slab_smem[];

    SI_STAMP_DECL;
    SI_STAMP_RT(0);
    SI_STAMP(1);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = (tid >> 6) & 3, wm = tid >> 8;   // the wave's 32-channel block; (W2) its half of the pixel blocks
    // block -> (slab, output-channel group): the groups of one slab sit 8 blocks apart (same XCD under round-robin placement:
    // the second one finds the patch in that L2)
    // (every argument the prologue needs is pinned into scalar registers HERE: left alone hipcc fetches the argument block in four
    // dependent rounds, a scalar-cache round trip each, in front of the first request; and the block's coordinates come from
    // reciprocal multiplies, not from two 30-instruction integer divisions)
    asm volatile("" ::"s"(a.in), "s"(a.wl), "s"(a.ih), "s"(a.iw), "s"(a.in_ld), "s"(a.oh), "s"(a.ow), "s"(a.wl_nb), "s"(a.wl_ks), "s"(a.th),
                 "s"(a.slabs_per_img), "s"(a.n_img), "s"(a.n_ocg), "s"(a.rowp), "s"(a.blk_bytes), "s"(a.pc), "s"(a.n_ppix), "s"(a.mg_pc),
                 "s"(a.mg_ow), "s"(a.mg_nocg), "s"(a.mg_spi), "s"(a.in_bytes));
    // block -> (image, slab of the image, output-channel group).  Blocks b and b + 8 share an XCD under round-robin placement (speed
    // only): an XCD takes WHOLE images -- image 8 k + (b & 7) -- so that the halo rows two neighbouring slabs both fetch, and the patch
    // the output-channel groups of a slab share, meet in one L2
    const int b8 = blockIdx.x & 7, bq = blockIdx.x >> 3;
    const int bqq = fdiv(bq, a.n_ocg, a.mg_nocg);
    const int ocg = bq - bqq * a.n_ocg;
    const int imgq = fdiv(bqq, a.slabs_per_img, a.mg_spi);
    const int img = imgq * 8 + b8, y0 = (bqq - imgq * a.slabs_per_img) * a.th;
    if (img >= a.n_img) return;
    const int rows_here = min(a.th, a.oh - y0);
    const int npix = rows_here * a.ow;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wl = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(a.wl), 0, (unsigned)a.wl_nb * (unsigned)a.wl_ks * 1024u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias ? a.bias : reinterpret_cast<const float*>(a.in)), 0, a.bias ? (unsigned)a.oc * 4u : 0u, 0x00020000);

    // this wave's 32 output channels: the weight ring
    const int nb = ocg * 4 + wave;
    const unsigned w_voff = nb < a.wl_nb ? (unsigned)nb * (unsigned)a.wl_ks * 1024u + (unsigned)lane * 16u : OOB_W;
    f16x8 rb[NB];
    u32x4 rp[N_IT];
    int l_off[N_IT];
    unsigned g_off[N_IT];
    // block 0 goes to LDS as soon as it has landed; block b + 1 is requested and committed under the MFMAs of block b (below)
    auto commit = [&](int b) {
#pragma unroll
        for (int i = 0; i < N_IT; ++i) *reinterpret_cast<u32x4*>(slab_smem + b * a.blk_bytes + l_off[i]) = rp[i];
    };

    if constexpr (!PW) {
        // ---- the patch of channel block 0.  A wave-level 16-byte load costs the CU's address path 16 cycles whatever its lanes fetch, and
        // four waves issue them: a first version that requested every channel block, a 12-deep weight ring and the bias up front spent
        // 4 000-6 500 cycles ISSUING (profiles/r05_slab_diag.txt) while the bytes of block 0 had long landed.  So only what the first MFMA
        // needs is requested here -- block 0 and a short weight ring; the other blocks, the bias and the shortcut are requested under the
        // MFMAs of the K loop.  Chunk c = tid + 256 i of the patch: patch pixel c >> 3 (row-major over (th + 2) x (ow + 2)), piece c & 7.
        const int ch = tid & 7;
        const int gbase = (img * a.ih + y0 - 1) * a.iw - 1;
        const unsigned pitch = (unsigned)(a.in_ld * 2);
#pragma unroll
        for (int i = 0; i < N_IT; ++i) {
            const int q = (tid >> 3) + (NT / 8) * i;
            const int pr = fdiv(q, a.pc, a.mg_pc), px = q - pr * a.pc;
            const int gy = y0 - 1 + pr, gx = px - 1;
            const bool live = q < a.n_ppix;
            const bool ok = live && (unsigned)gy < (unsigned)a.ih && (unsigned)gx < (unsigned)a.iw && !(SI_SLAB_ABL & 1);
            unsigned go = (unsigned)(gbase + pr * a.iw + px) * pitch + (unsigned)(ch * 16);
            int lo = pr * a.rowp + px * 144 + ch * 16;
            asm volatile("" : "+v"(go), "+v"(lo));   // (computed for every lane, then selected: no exec-mask branches around the arithmetic)
            g_off[i] = ok ? go : OOB;
            l_off[i] = live ? lo : a.rowp - 16;   // (a dead chunk lands in the unused tail of patch row 0)
            rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, g_off[i], 0, 0);
        }
#ifdef SI_SLAB_STAMP_PRO
        SI_STAMP(2);
#endif
#pragma unroll
        for (int j = 0; j < NB; ++j) rb[j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wl, w_voff, (unsigned)(j * 1024), 0));
        __builtin_amdgcn_sched_barrier(0);
#ifdef SI_SLAB_STAMP_PRO
        SI_STAMP(3);
#endif
        commit(0);
#ifdef SI_SLAB_STAMP_PRO
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        SI_STAMP(4);
        __syncthreads();
        SI_STAMP(5);
#else
        __syncthreads();
#endif
    } else {
        // ---- PW: y = SiLU(W_A x + b_A) for the (th + 2) x ow pixels of the patch that lie in the image, written into the patch as the
        // 3x3 conv's input; rows outside the image and the two pad columns are zeros (the 3x3 conv pads ITS input).  x: full-width
        // rows are contiguous pixels, so chunk c = tid + 256 i is x pixel c / (C / 8), piece c % (C / 8), no division; LDS image per
        // 64-channel block [pixel][144 B] (inside the patch region, which it does not outlive).  The same MFMA form as below: the
        // weights as the A operand, 16-deep steps in ascending k -- the same bits as the 1x1 conv's own launch.
        constexpr int C8 = NBLK * 8, LOG_C8 = NBLK == 2 ? 4 : 5;   // 16-byte pieces per pixel
        constexpr int N_ITX = 18 * 256 / NT;                       // x requests per thread (the host checks the x pixels fit)
        constexpr int TMX = TM == 7 ? 9 : 5;                       // 32-pixel blocks of x per wave
        constexpr int TNA = NBLK / 2;                              // 32-channel column blocks of y per wave
        constexpr int KSA = NBLK * 4, NBA = 8;                     // k-steps of the 1x1, weight fragments in flight
        static_assert(NBLK == 2 || NBLK == 4, "128 or 256 channels");
        const int rlo = max(0, 1 - y0), rhi = min(a.pr, a.ih - y0 + 1);   // patch rows inside the image
        const int qlo = rlo * a.ow, qhi = min(rhi * a.ow, a.n_xpix);
        const int gpix0 = (img * a.ih + y0 - 1) * a.iw;
        const unsigned pitch = (unsigned)(a.in_ld * 2);
        u32x4 rx[N_ITX];
#pragma unroll
        for (int i = 0; i < N_ITX; ++i) {
            const int c = tid + NT * i;
            const int q = c >> LOG_C8, ch = c & (C8 - 1);
            const bool ok = q >= qlo && q < qhi && !(SI_SLAB_ABL & 1);
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? (unsigned)(gpix0 + q) * pitch + (unsigned)(ch * 16) : OOB, 0, 0);
        }
        const __amdgpu_buffer_rsrc_t rs_wlA = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<half_t*>(a.wlA), 0, (unsigned)a.wlA_nb * (unsigned)a.wlA_ks * 1024u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_biasA = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.biasA ? a.biasA : reinterpret_cast<const float*>(a.in)), 0, a.biasA ? (unsigned)(NBLK * 64) * 4u : 0u, 0x00020000);
        unsigned wA_voff[TNA];
        f16x8 rbA[NBA][TNA];
#pragma unroll
        for (int u = 0; u < TNA; ++u) wA_voff[u] = (unsigned)(wave * TNA + u) * (unsigned)a.wlA_ks * 1024u + (unsigned)lane * 16u;
#pragma unroll
        for (int j = 0; j < NBA; ++j)
#pragma unroll
            for (int u = 0; u < TNA; ++u)
                rbA[j][u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wlA, wA_voff[u], (unsigned)(j * 1024), 0));
#pragma unroll
        for (int i = 0; i < N_ITX; ++i) {
            const int c = tid + NT * i;
            const int q = c >> LOG_C8, ch = c & (C8 - 1);
            // (a chunk behind the last x pixel lands behind the x images: the patch region is larger than they are)
            const int lo = q < a.n_xpix ? (ch >> 3) * a.x_blk + q * 144 + (ch & 7) * 16 : NBLK * a.x_blk;
            *reinterpret_cast<u32x4*>(slab_smem + lo) = rx[i];
        }
        __syncthreads();

        // (W2: the x pixel blocks are split between the two waves of a channel block, as the slab's are below)
        auto phase0 = [&](auto x0c, auto x1c) {
        constexpr int X0 = decltype(x0c)::value, X1 = decltype(x1c)::value;
        unsigned xbase[TMX];
#pragma unroll
        for (int t = X0; t < X1; ++t) {
            const int q = t * 32 + l31;
            xbase[t] = (unsigned)((q < a.n_xpix ? q : 0) * 144 + lh * 16);
        }
        f32x16 acc0[TMX][TNA];
#pragma unroll
        for (int t = X0; t < X1; ++t)
#pragma unroll
            for (int u = 0; u < TNA; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc0[t][u][e] = 0.0f;
        f16x8 fx[2][TMX];
#pragma unroll
        for (int t = X0; t < X1; ++t) fx[0][t] = *reinterpret_cast<const f16x8*>(slab_smem + xbase[t]);
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks) {
            if (ks + 1 < KSA) {
                const unsigned off = (unsigned)(((ks + 1) >> 2) * a.x_blk + ((ks + 1) & 3) * 32);
#pragma unroll
                for (int t = X0; t < X1; ++t) fx[(ks + 1) & 1][t] = *reinterpret_cast<const f16x8*>(slab_smem + xbase[t] + off);
            }
#pragma unroll
            for (int t = X0; t < X1; ++t)
#pragma unroll
                for (int u = 0; u < TNA; ++u) acc0[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rbA[ks % NBA][u], fx[ks & 1][t], acc0[t][u], 0, 0, 0);
            if (ks + NBA < KSA) {
#pragma unroll
                for (int u = 0; u < TNA; ++u)
                    rbA[ks % NBA][u] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wlA, wA_voff[u], (unsigned)((ks + NBA) * 1024), 0));
            }
            // (the 3x3 conv's weight ring is requested under these MFMAs)
            if (ks < NB) rb[ks] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wl, w_voff, (unsigned)(ks * 1024), 0));
#pragma unroll
            for (int t = X0; t < X1; ++t) {
                if (ks + 1 < KSA) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TNA, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, TNA + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();   // every wave is done with x: y goes over it (both halves of a wave pair cross it)

        // y: bias, SiLU, rounding (the 1x1 conv's own epilogue expressions), four channels of one pixel per 8-byte write at patch pixel
        // (row, column + 1); rows outside the image are the 3x3 conv's zero padding
#pragma unroll
        for (int u = 0; u < TNA; ++u) {
            const int chan0 = (wave * TNA + u) * 32 + 4 * lh;
            f32x4 bA[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bA[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_biasA, (unsigned)((chan0 + 8 * g) * 4), 0, 0));
#pragma unroll
            for (int t = X0; t < X1; ++t) {
                const int q = t * 32 + l31;
                const int r = fdiv(q < a.n_xpix ? q : 0, a.ow, a.mg_ow), cx = q - r * a.ow;
                const bool row_ok = (unsigned)(y0 - 1 + r) < (unsigned)a.ih;
                unsigned char* const yp = slab_smem + ((chan0 >> 6) * a.blk_bytes + r * a.rowp + (cx + 1) * 144 + (chan0 & 63) * 2);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = row_ok ? si_store_cast<half_t>(act_ct<SI_ACT_SILU>(acc0[t][u][4 * g + j] + bA[g][j])) : (half_t)0.0f;
                    if (q < a.n_xpix) *reinterpret_cast<f16x4*>(yp + g * 16) = hv;
                }
            }
        }
        };
        {
            using std::integral_constant;
            constexpr int XH = W2 ? (TMX + 1) / 2 : TMX;
            if (!W2 || wm == 0) phase0(integral_constant<int, 0>{}, integral_constant<int, XH>{});
            else phase0(integral_constant<int, XH>{}, integral_constant<int, TMX>{});
        }
        // the two pad columns of every patch row and channel block
        for (int z = tid; z < a.pr * 2 * NBLK * 8; z += NT) {
            const int ch = z & 7, side = (z >> 3) & 1, rb2 = z >> 4;
            const int blk = rb2 / a.pr, rr = rb2 - blk * a.pr;
            *reinterpret_cast<u32x4*>(slab_smem + (blk * a.blk_bytes + rr * a.rowp + (side ? (a.ow + 1) * 144 : 0) + ch * 16)) = u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
    }

    // A fragment bases: slab pixel p = 32 t + l31 -> (row, column) of the slab; a slot behind the last pixel re-reads pixel 0
    unsigned abase[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        int p = t * 32 + l31;
        p = p < npix ? p : 0;
        const int r = fdiv(p, a.ow, a.mg_ow), c = p - r * a.ow;
        abase[t] = (unsigned)(r * a.rowp + c * 144 + lh * 16);
    }
    __builtin_amdgcn_sched_barrier(0);   // (otherwise this arithmetic sinks between the first MFMAs)
#ifndef SI_SLAB_STAMP_PRO
    SI_STAMP(2);
#endif

    f32x16 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;

    // The last channel block runs in TWO passes over its 36 steps when the epilogue is SiLU (the YOLOv5 form): pixel blocks [0, TM1)
    // first, [TM1, TM) second, the weights of the block streamed twice -- so that the bias / SiLU / shortcut / rounding of the first
    // group's finished accumulators (38 issue cycles per element, 64 elements per lane) sits BETWEEN the second group's MFMAs instead
    // of behind the loop with the matrix pipe idle.  Same k order per output element: same bits.
#ifndef SI_SLAB_TM1
#define SI_SLAB_TM1 3
#endif
    constexpr int TM1 = (ACT1 == SI_ACT_SILU && !W2) ? (TM == 7 ? SI_SLAB_TM1 : TM / 2) : 0;   // (W2: one pass; the wave pair shares the epilogue)
    constexpr int STAGE_SP = 272;   // bytes per staged pixel: 128 channels + 16
    unsigned char* const stage = slab_smem + NBLK * a.blk_bytes;   // the epilogue's [pixel][128 channels] image, behind the patch

    f16x8 fa[2][TM];

    // the shortcut's values in the accumulators' layout (four channels of one pixel = 8 bytes per register quad) and the bias (C/D map
    // with the weights as the A operand: row = (e & 3) + 8 (e >> 2) + 4 lh -> output channel nb * 32 + row, col = lane & 31 -> slab pixel
    // 32 t + (lane & 31); register quad g = e >> 2 holds channels 8 g + 4 lh .. + 3: one 16-byte load) are requested under the MFMAs of the
    // LAST channel block
    const unsigned pix0 = (unsigned)((img * a.oh + y0) * a.ow);
    u32x2 rres[HAS_RES ? TM : 1][4];
    f32x4 bq4[4];
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(HAS_RES ? a.res : a.in), 0, HAS_RES ? a.res_bytes : 0u, 0x00020000);

    // one register quad of the epilogue: bias, activation, shortcut, activation, rounding (epilogue_lean_h's expressions per element),
    // four channels of one pixel as one 8-byte write into the staging image
    // elements [j0, j1) of register quad (t, g), finished and rounded
    auto quad_part = [&](auto a1c, auto a2c, int t, int g, int j0, int j1, f16x4& hv) {
        constexpr int A1 = decltype(a1c)::value, A2 = decltype(a2c)::value;
        f16x4 rv4 = {0, 0, 0, 0};
        if (HAS_RES) rv4 = __builtin_bit_cast(f16x4, rres[t][g]);
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            float v = act_ct<A1>(acc[t][4 * g + j] + bq4[g][j]);
            if (HAS_RES) v += (float)rv4[j];
            hv[j] = si_store_cast<half_t>(act_ct<A2>(v));
        }
    };
    auto quad_store = [&](f16x4 hv, int t, int g) {
        *reinterpret_cast<f16x4*>(stage + (t * 32 + l31) * STAGE_SP + (wave * 32 + 4 * lh) * 2 + g * 16) = hv;
    };
    using std::integral_constant;
    typedef integral_constant<int, SI_ACT_SILU> c_silu;
    typedef integral_constant<int, SI_ACT_NONE> c_none;
    typedef integral_constant<int, SI_ACT_RELU> c_relu;

    // The K loop, pinned (left alone hipcc re-pairs the fragment reads by address and consumes half of them right behind their
    // issue: an LDS round trip per MFMA).  One step = the fragment reads of the NEXT step, one between every two MFMAs of this
    // one (a read is consumed a whole step after its issue), then the refill of the weight-ring slot the step has just emptied and at
    // most a few other requests.  pass(b, [T0, T1), [N0, N1), MODE): the 36 steps of channel block b over pixel blocks [T0, T1); [N0, N1):
    // the pixel blocks of the pass behind it (for the fragment reads of its first step); MODE 0: an ordinary block, 1: the first pass
    // over the last block, 2: the second one (with the first group's epilogue between its MFMAs).
    auto pass = [&](auto bc, auto t0c, auto t1c, auto n0c, auto n1c, auto modec) {
        constexpr int b = decltype(bc)::value, T0 = decltype(t0c)::value, T1 = decltype(t1c)::value, N0 = decltype(n0c)::value,
                      N1 = decltype(n1c)::value, MODE = decltype(modec)::value;
        constexpr bool last_pass = b == NBLK - 1 && (MODE == 2 || TM1 == 0);
        constexpr int next_b = MODE == 1 ? b : b + 1;            // the channel block of the pass behind this one
        const unsigned blk_off = (unsigned)(b * a.blk_bytes);
        f16x4 hv_pending = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            int extra = 0;   // vector memory reads of this step besides the weight refill
            if (!PW && MODE == 0 && b + 1 < NBLK && (s & 1) && s / 2 < N_IT) {
                // (the block's offset rides in the scalar offset: it takes no part in the range check, so a dead chunk stays dead)
                rp[s / 2] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, g_off[s / 2], (unsigned)((b + 1) * 128), 0);
                extra = 1;
            }
            if (b == NBLK - 1 && MODE != 2 && (s & 1) && s / 2 < 4) {
                // (a buffer load: no bias = a zero-sized buffer = zeros, no branch)
                bq4[s / 2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, (unsigned)((nb * 32 + 8 * (s / 2) + 4 * lh) * 4), 0, 0));
                extra = 1;
            }
            if (HAS_RES && b == NBLK - 1 && MODE != 2 && !(s & 1) && s / 2 < TM) {
                const int t = s / 2, p = t * 32 + l31;
                const unsigned ro = (pix0 + (unsigned)p) * (unsigned)(a.res_ld * 2) + (unsigned)((nb * 32 + 4 * lh) * 2);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    rres[t][g] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_res, p < npix ? ro + (unsigned)(g * 16) : OOB, 0, 0));
                extra = 4;
            }
            if (!PW && MODE == 0 && b + 1 < NBLK && s == 28) commit(b + 1);
            if (!PW && MODE == 0 && b + 1 < NBLK && s == 34) __syncthreads();   // (the reads of step 35 reach into block b + 1)
            const bool more = (s + 1 < 36 || !last_pass) && !(SI_SLAB_ABL & 16);
            const int r0 = s + 1 < 36 ? T0 : N0, r1 = s + 1 < 36 ? T1 : N1;
            if (more) {
                const int s1 = (s + 1) % 36;
                const int tap = s1 / 4, q = s1 % 4, ky = tap / 3, kx = tap % 3;
                const unsigned off = (s + 1 < 36 ? blk_off : (unsigned)(next_b * a.blk_bytes)) + (unsigned)(ky * a.rowp + kx * 144 + q * 32);
#pragma unroll
                for (int t = r0; t < r1; ++t) fa[(s + 1) & 1][t] = *reinterpret_cast<const f16x8*>(slab_smem + abase[t] + off);
            }
#pragma unroll
            for (int t = T0; t < T1; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rb[s % NB], fa[s & 1][t], acc[t], 0, 0, 0);
            if (!(SI_SLAB_ABL & 8)) {
                // the k-step NB ahead in the order the passes run: this block's, then the next pass's (the same block again behind MODE 1)
                int ksn = s + NB < 36 ? b * 36 + s + NB : next_b * 36 + s + NB - 36;
                ksn = (last_pass && s + NB >= 36) ? KS_TOT - 1 : ksn;
                rb[s % NB] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wl, w_voff, (unsigned)ksn * 1024u, 0));
            }
            // the first group's epilogue, one register quad per QS steps, 4 / QS of its elements in each (every step is its own scheduling
            // region: in a region of several steps hipcc moved the fragment reads next to their use again)
            constexpr int QS = 36 / (4 * (TM1 ? TM1 : 9));   // steps per quad: 2 / 3 / 4 for four / three / two pixel blocks
            constexpr int EU = QS == 3 ? 2 : 4 / QS;          // elements per step (three steps: two, two, none)   // 2 steps per quad for four pixel blocks, 4 for two
            // (its LDS write goes behind the fragment reads of the region's LAST step: in front of them it would have to wait for every
            // value of the quad -- the compiler cannot tell that it does not alias the patch -- and drag them in front of those MFMAs)
            const bool hook = MODE == 2 && s / QS < 4 * TM1;
            if (hook && (s % QS) * EU < 4) quad_part(c_silu{}, c_none{}, (s / QS) / 4, (s / QS) % 4, (s % QS) * EU, (s % QS + 1) * EU, hv_pending);
            if (hook && (s % QS + 1) * EU == 4) quad_store(hv_pending, (s / QS) / 4, (s / QS) % 4);
            const int nr = more ? r1 - r0 : 0, nm = T1 - T0;
#pragma unroll
            for (int t = 0; t < (nr > nm ? nr : nm); ++t) {
                if (t < nr) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one ds_read
                if (t < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                if (MODE == 2) __builtin_amdgcn_sched_group_barrier(0x402, TM == 7 ? 7 : 6, 0);   // a share of the epilogue's vector instructions
            }
            if (!(SI_SLAB_ABL & 8)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // the weight load
            if (extra == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (extra == 4) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the wave's pixel blocks [LO, HI): all of them, or (W2) its half
    auto run = [&](auto loc, auto hic) {
        constexpr int LO = decltype(loc)::value, HI = decltype(hic)::value;
#pragma unroll
        for (int t = LO; t < HI; ++t) fa[0][t] = *reinterpret_cast<const f16x8*>(slab_smem + abase[t]);
        if (!(SI_SLAB_ABL & 2)) {
            typedef integral_constant<int, LO> cL;
            typedef integral_constant<int, HI> cT;
            typedef integral_constant<int, TM1> cT1;
            typedef integral_constant<int, TM1 ? TM1 : HI> cF;   // the first pass over the last block covers [LO, cF)
            typedef integral_constant<int, TM1 ? 1 : 0> cM;
            typedef integral_constant<int, 0> m0;                // MODE 0: an ordinary block
            if constexpr (NBLK == 2) {
                pass(integral_constant<int, 0>{}, cL{}, cT{}, cL{}, cF{}, m0{});
            } else if constexpr (NBLK == 4) {
                pass(integral_constant<int, 0>{}, cL{}, cT{}, cL{}, cT{}, m0{});
                pass(integral_constant<int, 1>{}, cL{}, cT{}, cL{}, cT{}, m0{});
                pass(integral_constant<int, 2>{}, cL{}, cT{}, cL{}, cF{}, m0{});
            }
            pass(integral_constant<int, NBLK - 1>{}, cL{}, cF{}, cT1{}, cT{}, cM{});
            if constexpr (TM1 != 0) pass(integral_constant<int, NBLK - 1>{}, cT1{}, cT{}, cL{}, cL{}, integral_constant<int, 2>{});
        }

        // ---- what is left of the epilogue: the second group's (or every) register quad, then the slab leaves as whole channel rows
#ifndef SI_SLAB_STAMP_PRO
        SI_STAMP(3);
#endif
        auto finish = [&](auto a1c, auto a2c) {
#pragma unroll
            for (int t = (TM1 > LO ? TM1 : LO); t < HI; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 hv;
                    quad_part(a1c, a2c, t, g, 0, 4, hv);
                    quad_store(hv, t, g);
                }
        };
        if (ACT1 == SI_ACT_SILU) finish(c_silu{}, c_none{});
        else if (a.act1 == SI_ACT_RELU) finish(c_relu{}, c_none{});
        else if (a.act2 == SI_ACT_RELU) finish(c_none{}, c_relu{});
        else finish(c_none{}, c_none{});
    };
    {
        // INVARIANT of the 512-thread form: the two halves of a wave pair run textually different instantiations of `run` (and, PW
        // form, of `phase0` above) and cross the barriers INSIDE them at different program counters.  s_barrier counts the workgroup's waves,
        // not program counters, so this is sound on gfx950 -- but only while BOTH instantiations execute the SAME NUMBER of barriers, which
        // they do by construction: `phase0` holds exactly one, unconditional; `run` makes the same sequence of passes whatever [LO, HI) is
        // (pass(b, MODE 0) for every b < NBLK - 1, then the last block) and a pass's barrier depends on template parameters (PW, MODE, b, NBLK)
        // and the step number alone -- LO / HI only choose which pixel blocks a half multiplies.  A change that makes a barrier depend on
        // LO / HI / TM1-per-half breaks this.  Held by tests/test_gpu_f16.py: the slab test runs this form AND the one-wave form
        // (SiConvPlan::f16_slab_w2 = 0) against the generic tiles on every NBLK / residual / ragged case, the bottleneck-pair tests the PW form;
        // a barrier-count mismatch hangs or corrupts there.
        constexpr int TH2 = W2 ? (TM + 1) / 2 : TM;
        if (!W2 || wm == 0) run(integral_constant<int, 0>{}, integral_constant<int, TH2>{});
        else run(integral_constant<int, TH2>{}, integral_constant<int, TM>{});
    }
    __syncthreads();
#ifndef SI_SLAB_STAMP_PRO
    SI_STAMP(4);
#endif
    {
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
        const unsigned obase = pix0 * (unsigned)(a.out_ld * 2) + (unsigned)(ocg * 256);
#pragma unroll
        for (int k = 0; k < 2 * TM * 256 / NT; ++k) {
            const int c = tid + NT * k;
            const int pixel = c >> 4, c16 = c & 15;
            const u32x4 val = *reinterpret_cast<const u32x4*>(stage + pixel * STAGE_SP + c16 * 16);
            const unsigned off = obase + (unsigned)pixel * (unsigned)(a.out_ld * 2) + (unsigned)(c16 * 16);
            if (!(SI_SLAB_ABL & 4)) __builtin_amdgcn_raw_buffer_store_b128(val, rs_out, pixel < npix ? off : OOB, 0, 0);
            else asm volatile("" ::"v"(val));
        }
    }
#ifndef SI_SLAB_STAMP_PRO
    SI_STAMP(5);
#endif
    SI_STAMP_RT(6);
    SI_STAMP_FLUSH(si_diag_stamps_slab);
}

int cu_count() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}

// on when the shape allows unless the call's plan says 0 (SiConvPlan::f16_slab)
bool slab_on(const SiConv2dDesc* d) {
    const int v = (d && d->plan && d->plan->f16_slab >= 0) ? d->plan->f16_slab : SI_ENV_INT("SI_CONV_F16_SLAB", 1);
    return v != 0;
}

struct SlabPlan {
    int tm, th, slabs_per_img, rowp, blk_bytes, lds;
};

constexpr int kLdsMax = 160 * 1024;

// the slab height: over every (TM in {7, 4}, th) whose slab fits TM 32-pixel blocks, the LDS and the staging registers, the one with
// the fewest MFMA slots on the busiest CU (rounds of the grid over the chip x TM); ties -> the larger grid; th is then evened
// out over the slabs of an image
bool slab_plan(const SiConv2dDesc* d, SlabPlan* out, int only_tm = 0) {
    const int nblk = d->ic / 64;
    const int n_ocg = (d->oc + 127) / 128;
    const int rowp = d->ow * 144 + 512;
    long long best_cost = -1, best_grid = 0;
    SlabPlan best{};
    for (int tm : {7, 4}) {
        if (only_tm && tm != only_tm) continue;
        const int n_it = tm == 7 ? 11 : 7;
        int th = tm * 32 / d->ow;
        if (th > d->oh) th = d->oh;
        for (; th >= 1; --th) {
            const int pr = th + 2;
            if ((long long)nblk * pr * rowp + tm * 32 * 272 > kLdsMax || (long long)pr * (d->ow + 2) * 8 > (long long)n_it * 256) continue;
            const int spi = (d->oh + th - 1) / th;
            const long long grid = (long long)d->n * spi * n_ocg;
            const long long cost = (grid + cu_count() - 1) / cu_count() * tm;
            if (best_cost < 0 || cost < best_cost || (cost == best_cost && grid > best_grid)) {
                best_cost = cost;
                best_grid = grid;
                const int th_even = (d->oh + spi - 1) / spi;
                best = SlabPlan{tm, th_even, spi, rowp, (th_even + 2) * rowp, nblk * (th_even + 2) * rowp};
            }
        }
    }
    if (best_cost < 0) return false;
    *out = best;
    return true;
}

bool slab_shape_ok(const SiConv2dDesc* d) {
    const bool acts = (d->act1 == SI_ACT_SILU && d->act2 == SI_ACT_NONE) || (d->act1 == SI_ACT_RELU && d->act2 == SI_ACT_NONE) ||
                      (d->act1 == SI_ACT_NONE && (d->act2 == SI_ACT_NONE || d->act2 == SI_ACT_RELU));
    return acts && d->groups == 1 && (d->ic == 128 || d->ic == 256) && d->oc % 128 == 0 && d->kh == 3 && d->kw == 3 && d->sh == 1 && d->sw == 1 &&
           d->dh == 1 && d->dw == 1 && d->pt == 1 && d->pl == 1 && d->oh == d->ih && d->ow == d->iw && d->ow <= 94 && d->in_ld % 8 == 0 &&
           d->out_ld % 8 == 0 && (!d->has_residual || d->res_ld % 4 == 0);
}

template <int TM, int NBLK, int N_IT, bool W2 = false>
int launch_slab(const SlabArgs& a, const SiConv2dDesc* d, int lds, hipStream_t s) {
    auto go = [&](auto kern) {
        const hipError_t e = si_allow_dynamic_lds(kern, (size_t)lds);
        if (e != hipSuccess) return (int)e;
        const int grid = (a.n_img + 7) / 8 * 8 * a.slabs_per_img * a.n_ocg;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(W2 ? 512 : 256), (size_t)lds, s, a);
        return (int)hipGetLastError();
    };
    const bool silu = d->act1 == SI_ACT_SILU && d->act2 == SI_ACT_NONE;
    if (d->has_residual)
        return silu ? go(conv3x3s1_slab_f16_kernel<TM, NBLK, N_IT, SI_ACT_SILU, true, false, W2>) : go(conv3x3s1_slab_f16_kernel<TM, NBLK, N_IT, SI_ACT_NONE, true, false, W2>);
    return silu ? go(conv3x3s1_slab_f16_kernel<TM, NBLK, N_IT, SI_ACT_SILU, false, false, W2>) : go(conv3x3s1_slab_f16_kernel<TM, NBLK, N_IT, SI_ACT_NONE, false, false, W2>);
}

// two waves per SIMD (512-thread workgroups): measured +4 % on the 7-block form (40x40x128: its vector-bound phases halve, 3.8 k of
// 26 k cycles, the K loop pays 2 k for the doubled weight-fragment traffic) and 1.30x instead of 1.12x on its fused bottleneck pair;
// -5 % on the 4-block form (20x20x256: two MFMAs per fragment and wave = 64 B/clk/CU through L1), which therefore keeps one wave per
// SIMD and has no 512-thread instantiation.  SiConvPlan::f16_slab_w2 = 0: one wave per SIMD everywhere (A/B runs, tests).
bool slab_w2(const SiConv2dDesc* d, int tm) {
    const int v = (d && d->plan && d->plan->f16_slab_w2 >= 0) ? d->plan->f16_slab_w2 : SI_ENV_INT("SI_CONV_F16_SLAB_W2", 1);
    return v != 0 && tm == 7;
}

// staging requests per thread of the instantiation that serves a plan: exact for the two YOLOv5s forms (10 for 5 x 40-pixel slabs over
// 128 channels, 5 for 5 x 20 over 256), the upper bound 11 / 7 otherwise
int slab_nit(const SiConv2dDesc* d, const SlabPlan& p) {
    const int need512 = ((p.th + 2) * (d->ow + 2) * 8 + 511) / 512;
    if (slab_w2(d, 7) && p.tm == 7 && d->ic == 128 && need512 <= 5) return 5;     // (512-thread forms)
    const int need = ((p.th + 2) * (d->ow + 2) * 8 + 255) / 256;
    if (p.tm == 7 && d->ic == 128 && need <= 10) return 10;
    if (p.tm == 4 && d->ic == 256 && need <= 5) return 5;
    return p.tm == 7 ? 11 : 7;
}

}  // namespace

// conv_igemm_f16.hip's dispatch asks here first; 0: not this kernel's shape (or switched off)
bool si_conv_slab_f16_ok(const SiConv2dDesc* d) {
    SlabPlan p;
    return slab_on(d) && slab_shape_ok(d) && slab_plan(d, &p);
}

const char* si_conv_slab_f16_name(const SiConv2dDesc* d) {
    SlabPlan p;
    if (!slab_shape_ok(d) || !slab_plan(d, &p)) return "";
    const bool silu = d->act1 == SI_ACT_SILU && d->act2 == SI_ACT_NONE;
    static thread_local char name[64];
    const int nit = slab_nit(d, p);
    snprintf(name, sizeof(name), "conv3x3s1_slab_f16_kernel<%d, %d, %d, %d, %s, false, %s>", p.tm, d->ic / 64, nit, silu ? SI_ACT_SILU : SI_ACT_NONE,
             d->has_residual ? "true" : "false", (p.tm == 7 && nit == 5) ? "true" : "false");
    return name;
}

namespace {

struct PwInfo {
    const half_t* wlA;
    const float* biasA;
    int wlA_nb, wlA_ks;
};

// what the fused 1x1 form needs beyond slab_shape_ok / slab_plan of the 3x3 conv: the C3 bottleneck's first conv (1x1, stride 1, same
// channel count, bias optional, SiLU), one of the two instantiated forms, x small enough for its staging requests and its image
bool pw_pair_ok(const SiConv2dDesc* pw, const SiConv2dDesc* d, const SlabPlan& p) {
    const bool pw_shape = pw->groups == 1 && pw->kh == 1 && pw->kw == 1 && pw->sh == 1 && pw->sw == 1 && pw->pt == 0 && pw->pl == 0 &&
                          pw->dh == 1 && pw->dw == 1 && pw->ic == d->ic && pw->oc == d->ic && pw->n == d->n && pw->ih == d->ih && pw->iw == d->iw &&
                          pw->oh == d->ih && pw->ow == d->iw && pw->act1 == SI_ACT_SILU && pw->act2 == SI_ACT_NONE && !pw->has_residual &&
                          pw->in_ld % 8 == 0;
    if (!pw_shape || d->act1 != SI_ACT_SILU || d->act2 != SI_ACT_NONE) return false;
    if (!((p.tm == 7 && d->ic == 128) || (p.tm == 4 && d->ic == 256))) return false;
    const int n_xpix = (p.th + 2) * d->ow, tmx = p.tm == 7 ? 9 : 5, nblk = d->ic / 64;
    if (n_xpix > tmx * 32 || n_xpix * nblk * 8 > 18 * 256) return false;
    return (long long)nblk * n_xpix * 144 + 16 <= (long long)p.lds;
}

// (the fused 1x1 form exists for 7 pixel blocks over 128 channels and 4 over 256: its plan is the best one with that block count)
int pw_tm(const SiConv2dDesc* d) { return d->ic == 128 ? 7 : 4; }

int slab_launch(const SiConv2dDesc* d, const void* in, unsigned long long in_bytes, int in_ld, const void* wl, int wl_nb, int wl_ks,
                const float* bias, const void* residual, void* out, hipStream_t s, const PwInfo* pw) {
    SlabPlan p;
    if (!slab_shape_ok(d) || !slab_plan(d, &p, pw ? pw_tm(d) : 0)) return SI_E_UNSUPPORTED;
    SlabArgs a;
    a.in = static_cast<const half_t*>(in);
    a.wl = static_cast<const half_t*>(wl);
    a.bias = d->has_bias ? bias : nullptr;
    a.res = d->has_residual ? static_cast<const half_t*>(residual) : nullptr;
    a.out = static_cast<half_t*>(out);
    a.ih = d->ih; a.iw = d->iw; a.in_ld = in_ld; a.oh = d->oh; a.ow = d->ow; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.oc = d->oc;
    a.wl_nb = wl_nb; a.wl_ks = wl_ks;
    a.th = p.th;
    a.slabs_per_img = p.slabs_per_img;
    a.n_img = d->n;
    a.n_ocg = (d->oc + 127) / 128;
    a.rowp = p.rowp; a.blk_bytes = p.blk_bytes;
    a.pc = d->ow + 2;
    a.mg_pc = (unsigned)(0x100000000ull / (unsigned)a.pc);
    a.mg_ow = d->ow > 1 ? (unsigned)(0x100000000ull / (unsigned)d->ow) : 0xFFFFFFFFu;
    a.mg_nocg = a.n_ocg > 1 ? (unsigned)(0x100000000ull / (unsigned)a.n_ocg) : 0xFFFFFFFFu;
    a.mg_spi = a.slabs_per_img > 1 ? (unsigned)(0x100000000ull / (unsigned)a.slabs_per_img) : 0xFFFFFFFFu;
    a.n_ppix = (p.th + 2) * a.pc;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    const unsigned long long out_bytes = (unsigned long long)d->n * d->oh * d->ow * d->out_ld * 2ull;
    const unsigned long long res_bytes = d->has_residual ? (unsigned long long)d->n * d->oh * d->ow * d->res_ld * 2ull : 0ull;
    if (out_bytes >= 0xFFFFFF00ull || res_bytes >= 0xFFFFFF00ull || in_bytes >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;
    a.in_bytes = (unsigned)in_bytes;
    a.out_bytes = (unsigned)out_bytes;
    a.res_bytes = (unsigned)res_bytes;
    a.wlA = nullptr; a.biasA = nullptr; a.wlA_nb = a.wlA_ks = 0;
    a.pr = p.th + 2;
    a.n_xpix = a.pr * d->ow;
    a.x_blk = a.n_xpix * 144;
    const int lds = p.lds + p.tm * 32 * 272;   // the patch, then the epilogue's [pixel][128 channels] image
    if (pw) {
        a.wlA = pw->wlA; a.biasA = pw->biasA; a.wlA_nb = pw->wlA_nb; a.wlA_ks = pw->wlA_ks;
        auto go = [&](auto kern, int threads) {
            const hipError_t e = si_allow_dynamic_lds(kern, (size_t)lds);
            if (e != hipSuccess) return (int)e;
            const int grid = (a.n_img + 7) / 8 * 8 * a.slabs_per_img * a.n_ocg;
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)threads), (size_t)lds, s, a);
            return (int)hipGetLastError();
        };
        if (slab_w2(d, pw_tm(d)))
            return d->has_residual ? go(conv3x3s1_slab_f16_kernel<7, 2, 1, SI_ACT_SILU, true, true, true>, 512) : go(conv3x3s1_slab_f16_kernel<7, 2, 1, SI_ACT_SILU, false, true, true>, 512);
        if (d->ic == 128) return d->has_residual ? go(conv3x3s1_slab_f16_kernel<7, 2, 1, SI_ACT_SILU, true, true>, 256) : go(conv3x3s1_slab_f16_kernel<7, 2, 1, SI_ACT_SILU, false, true>, 256);
        return d->has_residual ? go(conv3x3s1_slab_f16_kernel<4, 4, 1, SI_ACT_SILU, true, true>, 256) : go(conv3x3s1_slab_f16_kernel<4, 4, 1, SI_ACT_SILU, false, true>, 256);
    }
    const int nit = slab_nit(d, p);
    if (d->ic == 128) {
        if (p.tm == 7) return nit == 5 ? launch_slab<7, 2, 5, true>(a, d, lds, s) : (nit == 10 ? launch_slab<7, 2, 10>(a, d, lds, s) : launch_slab<7, 2, 11>(a, d, lds, s));
        return launch_slab<4, 2, 7>(a, d, lds, s);
    }
    if (p.tm == 7) return launch_slab<7, 4, 11>(a, d, lds, s);
    return nit == 5 ? launch_slab<4, 4, 5>(a, d, lds, s) : launch_slab<4, 4, 7>(a, d, lds, s);
}

}  // namespace

int si_conv_slab_f16_launch(const SiConv2dDesc* d, const void* in, const void* wl, int wl_nb, int wl_ks, const float* bias,
                            const void* residual, void* out, hipStream_t s) {
    return slab_launch(d, in, (unsigned long long)d->n * d->ih * d->iw * d->in_ld * 2ull, d->in_ld, wl, wl_nb, wl_ks, bias, residual, out, s, nullptr);
}

// (conv_pw_patch_f16.hip: the 64-channel pair on the persistent patch kernel)
bool si_conv_pw_patch_f16_ok(const SiConv2dDesc* pw, const SiConv2dDesc* d);
int si_conv_pw_patch_f16_launch(const SiConv2dDesc* pw, const SiConv2dDesc* d, const void* in, const void* wlA, const float* biasA, const void* wl,
                                const float* bias, const void* residual, void* out, hipStream_t s);

extern "C" int si_hip_conv2d_pw_slab_f16_supported(const SiConv2dDesc* pw, const SiConv2dDesc* conv) {
    if (si_conv_pw_patch_f16_ok(pw, conv)) return 2;
    SlabPlan p, natural;
    if (!pw || !conv || !slab_on(conv) || !slab_shape_ok(conv) || !slab_plan(conv, &p, pw_tm(conv)) || !pw_pair_ok(pw, conv, p)) return 0;
    // 2: ... and it is the plan the 3x3 conv would run under by itself on a grid that covers most of the chip (what an engine fuses on)
    const long long grid = (long long)conv->n * p.slabs_per_img * ((conv->oc + 127) / 128);
    // (... and in the form measured faster than two launches: the 7-block one on two waves per SIMD, 1.30x; the 4-block one is 0.91x)
    return (slab_plan(conv, &natural) && natural.tm == p.tm && natural.th == p.th && grid * 4 >= (long long)cu_count() * 3 && slab_w2(conv, p.tm)) ? 2 : 1;
}

extern "C" int si_hip_conv2d_pw_slab_f16(const SiConv2dDesc* pw, const SiConv2dDesc* conv, const void* in, const void* pw_w_packed,
                                         const float* pw_bias, const void* w_packed, const float* bias, const void* residual, void* out,
                                         si_stream_t stream) {
    if (!pw || !conv || !in || !pw_w_packed || !w_packed || !out) return SI_E_BADARG;
    if ((pw->has_bias && !pw_bias) || (conv->has_bias && !bias) || (conv->has_residual && !residual)) return SI_E_BADARG;
    if (si_conv_pw_patch_f16_ok(pw, conv)) {
        // (si_hip_conv2d_f16_pack_weight_host: the row-major image, then the lane-order one)
        const size_t cc = (size_t)conv->ic * conv->ic;
        const half_t* const wA = static_cast<const half_t*>(pw_w_packed) + cc;
        const half_t* const wB = static_cast<const half_t*>(w_packed) + 9 * cc;
        return si_conv_pw_patch_f16_launch(pw, conv, in, wA, pw_bias, wB, bias, residual, out, static_cast<hipStream_t>(stream));
    }
    SlabPlan p;
    if (!slab_shape_ok(conv) || !slab_plan(conv, &p, pw_tm(conv)) || !pw_pair_ok(pw, conv, p)) return SI_E_UNSUPPORTED;
    const uintptr_t al = reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | (conv->has_bias ? reinterpret_cast<uintptr_t>(bias) : 0) |
                         (pw->has_bias ? reinterpret_cast<uintptr_t>(pw_bias) : 0);
    if ((al & 15) != 0 || (conv->has_residual && (reinterpret_cast<uintptr_t>(residual) & 7) != 0)) return SI_E_UNSUPPORTED;
    const int c = conv->ic;
    // (si_hip_conv2d_f16_pack_weight_host: the row-major image, then the lane-order one)
    const half_t* const wA = static_cast<const half_t*>(pw_w_packed);
    const half_t* const wB = static_cast<const half_t*>(w_packed);
    PwInfo info{wA + (size_t)c * c, pw->has_bias ? pw_bias : nullptr, c / 32, c / 16};
    return slab_launch(conv, in, (unsigned long long)pw->n * pw->ih * pw->iw * pw->in_ld * 2ull, pw->in_ld, wB + (size_t)conv->oc * 9 * c, conv->oc / 32, 9 * c / 16, bias,
                       residual, out, static_cast<hipStream_t>(stream), &info);
}

#ifdef SI_DIAG_STAMPS
SI_STAMP_ACCESSORS(si_diag_stamps_slab, si_hip_diag_stamps_read_slab, si_hip_diag_stamps_clear_slab)
#endif


