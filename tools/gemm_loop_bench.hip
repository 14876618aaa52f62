// tools/gemm_loop_bench.hip -- K-loop structures for the fp32 implicit-GEMM conv, measured as a plain GEMM
// C[M][N] = A[M][K] * B[N][K]^T (a 1x1 convolution; the 3x3 kernels differ only in the A address), with a minimal
// epilogue (one store per accumulator register), non-persistent one-tile workgroups like the product kernel, so launch
// ramp, tail and L2 behaviour are included.  Development tool, not product code.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/glb tools/gemm_loop_bench.hip && /tmp/glb [M N K]
//
// VARIANT 0  round-1 structure: 64x64 tile, BK 32, register-staged buffer loads, ds_write_b128 into [row][36], ONE LDS stage,
//            two barriers per K-tile
// VARIANT 1  LDS-DMA (buffer_load_dwordx4 ... lds) into an unpadded, XOR-swizzled [row][32] image, TWO stages, one barrier
//            per K-tile, fragments of the K-tile read up front
// VARIANT 2  VARIANT 1 with BK 64 (two 32-wide sub-tiles per stage)
// VARIANT 3  VARIANT 1 on a 128x64 tile (wave tile 64x32: two accumulator chains, A fragments twice as many)
// VARIANT 4  VARIANT 1 with THREE stages and a counted vmcnt (the DMA of tile t+2 in flight across the barrier)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

struct Args {
    const float* A;
    const float* B;
    float* C;
    unsigned long long* stamps;  // per workgroup: cycles, realtime ticks, start cycle, hw id
    int M, N, K, m_tiles, n_tiles;
};

__device__ __forceinline__ void tile_of_block(const Args& a, int& m_tile, int& n_tile) {
    // the product kernel's map: the n-tiles of one m-panel share blockIdx % 8 (one XCD)
    const int per_chunk = 8 * a.n_tiles;
    const int chunk = blockIdx.x / per_chunk;
    const int r = blockIdx.x - chunk * per_chunk;
    m_tile = chunk * 8 + (r & 7);
    n_tile = r >> 3;
}

template <int TM>
__device__ __forceinline__ void store_acc(const Args& a, f32x16 (&acc)[TM], int m0, int n0, int wm, int wn, int l31, int lh) {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + (wm * TM + t) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (m < a.M) a.C[(size_t)m * a.N + n0 + wn * 32 + l31] = acc[t][e];
        }
}

__device__ __forceinline__ void stamp(const Args& a, unsigned long long t0, unsigned long long r0) {
    if (threadIdx.x == 0 && a.stamps) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.stamps[4 * blockIdx.x + 0] = t1 - t0;
        a.stamps[4 * blockIdx.x + 1] = r1 - r0;
        a.stamps[4 * blockIdx.x + 2] = r0;
        a.stamps[4 * blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hw;
    }
}

// ---- VARIANT 0 -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_v0(const Args a) {
    constexpr int LD = 36;
    __shared__ __attribute__((aligned(16))) float lds[128 * LD];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    int m_tile, n_tile;
    tile_of_block(a, m_tile, n_tile);
    if (m_tile >= a.m_tiles) return;
    const int m0 = m_tile * 64, n0 = n_tile * 64;
    const int tid = threadIdx.x, kv = tid & 7, r0 = tid >> 3;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (unsigned)((size_t)a.M * a.K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B), 0, (unsigned)((size_t)a.N * a.K * 4), 0x00020000);
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + r0 + 32 * i;
        aoff[i] = m < a.M ? (unsigned)m * (unsigned)(a.K * 4) + kv * 16 : 0xFFFFFF00u;
        boff[i] = (unsigned)(n0 + r0 + 32 * i) * (unsigned)(a.K * 4) + kv * 16;
    }
    u32x4 pa[2], pb[2];
    auto load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) pa[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[i] + kt * 128, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) pb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, boff[i] + kt * 128, 0, 0);
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(lds + (r0 + 32 * i) * LD + kv * 4) = pa[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(lds + 64 * LD + (r0 + 32 * i) * LD + kv * 4) = pb[i];
    };
    const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[1];
    for (int e = 0; e < 16; ++e) acc[0][e] = 0.f;
    const int nk = a.K / 32;
    load(0);
    store();
    __syncthreads();
    const float* As = lds + (wm * 32 + l31) * LD + lh * 4;
    const float* Bs = lds + 64 * LD + (wn * 32 + l31) * LD + lh * 4;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load(kt + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(As + q * 8);
            const f32x4 fb = *reinterpret_cast<const f32x4*>(Bs + q * 8);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc[0], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store();
            __syncthreads();
        }
    }
    store_acc<1>(a, acc, m0, n0, wm, wn, l31, lh);
    stamp(a, t0, rt0);
}

// ---- LDS-DMA variants ------------------------------------------------------------------------------------------------
// stage image: rows of 32 floats (128 B), unpadded; 16-byte chunk c of row r sits at chunk position c ^ ((r >> 1) & 7):
// conflict free for the ds_read_b128 fragment pattern (16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} see 16 distinct
// (row parity, chunk position) pairs).  The DMA writes lane i of a wave-instruction at base + 16 i, so lane i FETCHES the
// chunk that belongs at its position.
template <int BM, int BKT /* 32-wide sub-tiles per stage */, int NST, bool COUNTED>
__global__ __launch_bounds__(256) void gemm_dma(const Args a) {
    constexpr int TM = BM / 64;               // wave tile rows / 32
    constexpr int ROWS = BM + 64;             // A rows + B rows per sub-tile
    constexpr int SUB = ROWS * 32;            // floats per sub-tile
    constexpr int STAGE = SUB * BKT;
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    int m_tile, n_tile;
    tile_of_block(a, m_tile, n_tile);
    if (m_tile >= a.m_tiles) return;
    const int m0 = m_tile * BM, n0 = n_tile * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases and descriptors stay scalar
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (unsigned)((size_t)a.M * a.K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B), 0, (unsigned)((size_t)a.N * a.K * 4), 0x00020000);
    // DMA pieces: one wave-instruction = 8 rows x 128 B.  ROWS / 8 pieces per sub-tile, dealt round-robin to the 4 waves.
    constexpr int PIECES = ROWS / 8;          // 16 (BM 64) or 24 (BM 128)
    constexpr int PPW = PIECES / 4;           // pieces per wave per sub-tile
    unsigned goff[PPW];                       // this lane's source byte offset of piece j (k = 0)
    constexpr int APW = BM / 32;              // pieces j < APW are A rows, the rest B rows (piece = 4 j + wave)
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int piece = j * 4 + wave;
        const int row = piece * 8 + (lane >> 3);          // row inside the sub-tile image (A rows first, then B rows)
        const int c = (lane & 7) ^ ((row >> 1) & 7);      // the chunk that belongs at this lane's position
        if (j < APW) {
            const int m = m0 + row;
            goff[j] = m < a.M ? (unsigned)m * (unsigned)(a.K * 4) + c * 16 : 0xFFFFFF00u;
        } else {
            goff[j] = (unsigned)(n0 + row - BM) * (unsigned)(a.K * 4) + c * 16;
        }
    }
    auto dma = [&](int kt, int stage) {       // kt counts BKT-wide steps
#pragma unroll
        for (int s = 0; s < BKT; ++s)
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int piece = j * 4 + wave;
                float* dst = lds + stage * STAGE + s * SUB + piece * 256;
                const unsigned off = goff[j] + (unsigned)(kt * BKT + s) * 128u;
                if (j < APW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
            }
    };
    f32x16 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int nk = a.K / (32 * BKT);
    // fragment read offsets (floats) inside a sub-tile image
    int ra[TM], rb;
#pragma unroll
    for (int t = 0; t < TM; ++t) ra[t] = ((wm * TM + t) * 32 + l31) * 32;
    rb = (BM + wn * 32 + l31) * 32;
    const int swA = (l31 >> 1) & 7;           // (row >> 1) & 7 is the same for every 32-row block (32 % 16 == 0)

    dma(0, 0);
    if (NST == 3 && nk > 1) dma(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt % NST;
        if (COUNTED && NST == 3) {
            // tile kt must have landed; tile kt+1 (issued one iteration ago) may stay in flight
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * BKT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (NST == 3) { if (kt + 2 < nk) dma(kt + 2, (kt + 2) % NST); }
        else if (kt + 1 < nk) dma(kt + 1, (kt + 1) % NST);
        const float* S = lds + st * STAGE;
#pragma unroll
        for (int s = 0; s < BKT; ++s) {
            f32x4 fa[TM][4], fb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pos = ((2 * q + lh) ^ swA) * 4;
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[t][q] = *reinterpret_cast<const f32x4*>(S + s * SUB + ra[t] + pos);
                fb[q] = *reinterpret_cast<const f32x4*>(S + s * SUB + rb + pos);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < TM; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t][q][j], fb[q][j], acc[t], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    store_acc<TM>(a, acc, m0, n0, wm, wn, l31, lh);
    stamp(a, t0, rt0);
}

// ---- host --------------------------------------------------------------------------------------------------------------
struct Result { float ms; double tf; };

template <typename F>
Result time_it(F launch, double flops, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return {ms, flops / (ms * 1e-3) / 1e12};
}

void report(const char* name, const Args& a, int grid, Result r, unsigned long long* dst) {
    std::vector<unsigned long long> st((size_t)4 * grid);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc;
    double clk = 0;
    int nclk = 0;
    unsigned long long rmin = ~0ull, rmax = 0;
    std::vector<int> per_cu(8 * 64 * 4, 0);
    for (int b = 0; b < grid; ++b) {
        if (st[4 * b + 1] == 0) continue;
        cyc.push_back((double)st[4 * b]);
        clk += (double)st[4 * b] / ((double)st[4 * b + 1] * 10.0);
        ++nclk;
        rmin = std::min(rmin, st[4 * b + 2]);
        rmax = std::max(rmax, st[4 * b + 2] + st[4 * b + 1]);
        const unsigned hw = (unsigned)st[4 * b + 3], xcc = (unsigned)(st[4 * b + 3] >> 32) & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_cu[((xcc * 8 + se) * 2 + sh) * 16 + cu]++;
    }
    std::sort(cyc.begin(), cyc.end());
    int used = 0, mx = 0, mn = 1 << 30;
    for (int v : per_cu) if (v) { ++used; mx = std::max(mx, v); mn = std::min(mn, v); }
    printf("%-34s grid %6d  %8.4f ms  %6.1f TF/s  clock %.2f GHz  wg cycles med %7.0f p95 %7.0f  kernel span %6.1f us  CUs used %d (wg/CU min %d max %d)\n",
           name, grid, r.ms, r.tf, nclk ? clk / nclk : 0.0, cyc.empty() ? 0.0 : cyc[cyc.size() / 2], cyc.empty() ? 0.0 : cyc[cyc.size() * 95 / 100],
           (double)(rmax - rmin) / 100.0, used, mn, mx);
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 51200, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 1152;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int only = argc > 5 ? atoi(argv[5]) : -1;   // run one variant only (sustained-load experiments)
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s  M %d N %d K %d  (%.2f GFLOP)\n", p.name, M, N, K, 2.0 * M * N * K / 1e9);
    float *dA, *dB, *dC;
    unsigned long long* dst;
    CK(hipMalloc(&dA, (size_t)M * K * 4));
    CK(hipMalloc(&dB, (size_t)N * K * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dst, (size_t)4 * 8 * 65536));
    {
        std::vector<float> h((size_t)M * K);
        unsigned s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
        CK(hipMemcpy(dA, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        h.resize((size_t)N * K);
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
        CK(hipMemcpy(dB, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    const double flops = 2.0 * M * N * K;
    Args a{dA, dB, dC, dst, M, N, K, 0, N / 64};
    std::vector<float> ref((size_t)M * N), got((size_t)M * N);

    auto run64 = [&](const char* name, auto kernel, size_t smem, int BM) {
        Args b = a;
        b.m_tiles = (M + BM - 1) / BM;
        const int grid = (b.m_tiles + 7) / 8 * 8 * b.n_tiles;
        CK(hipMemset(dst, 0, (size_t)4 * 8 * grid));
        if (smem) CK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        CK(hipMemset(dC, 0, (size_t)M * N * 4));
        Result r = time_it([&] { hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), smem, 0, b); }, flops, reps);
        CK(hipGetLastError());
        report(name, b, grid, r, dst);
        CK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
    };
    if (only < 0 || only == 0) run64("v0 64x64 BK32 regs 1 stage", gemm_v0, 0, 64);
    ref = got;
    auto check = [&](const char* name) {
        double md = 0, mr = 0;
        for (size_t i = 0; i < ref.size(); ++i) { md = std::max(md, (double)fabsf(ref[i] - got[i])); mr = std::max(mr, (double)fabsf(ref[i])); }
        printf("    %s vs v0: max diff %.3g (scale %.3g)%s\n", name, md, mr, md > 1e-3 * mr ? "  <-- MISMATCH" : "");
    };
    if (only < 0 || only == 1) run64("v1 64x64 BK32 dma 2 stages", gemm_dma<64, 1, 2, false>, 2 * 128 * 32 * 4, 64); check("v1");
    if (only < 0 || only == 2) run64("v2 64x64 BK64 dma 2 stages", gemm_dma<64, 2, 2, false>, 2 * 2 * 128 * 32 * 4, 64); check("v2");
    if (only < 0 || only == 3) run64("v3 128x64 BK32 dma 2 stages", gemm_dma<128, 1, 2, false>, 2 * 192 * 32 * 4, 128); check("v3");
    if (only < 0 || only == 4) run64("v4 64x64 BK32 dma 3 stages counted", gemm_dma<64, 1, 3, true>, 3 * 128 * 32 * 4, 64); check("v4");
    if (only < 0 || only == 5) run64("v5 128x64 BK32 dma 3 stages counted", gemm_dma<128, 1, 3, true>, 3 * 192 * 32 * 4, 128); check("v5");
    return 0;
}
