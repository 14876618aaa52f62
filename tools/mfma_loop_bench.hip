// tools/mfma_loop_bench.hip -- what the fp32 matrix pipe sustains under the K-loop structures the implicit-GEMM conv can
// take (development tool, not product code).  One 256-thread workgroup = 4 waves = a 64x64 tile of 32x32 wave tiles, K-tile 32
// -> 16 v_mfma_f32_32x32x2_f32 per wave per iteration (1024 pipe cycles).  No global memory in the loop: the question is how
// much of the pipe the LDS reads, LDS writes and barriers cost at R resident workgroups per CU.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_loop_bench tools/mfma_loop_bench.hip && /tmp/mfma_loop_bench
//
// MODE 0  MFMA chain only (operands in registers)
// MODE 1  + the fragment reads as the round-1 kernel does them: per 4 MFMAs {2 ds_read_b128, wait}
// MODE 2  MODE 1 + barrier, 4 ds_write_b128, barrier per iteration (single LDS stage, the round-1 structure)
// MODE 3  like 2, but all 8 fragment reads of the K-tile are issued up front, then 16 MFMAs
// MODE 4  two LDS stages: fragment reads up front, 16 MFMAs, 4 ds_write_b128 to the other stage, ONE barrier
// MODE 5  like 2, fragment reads double-buffered in registers (reads of group q+1 issued before the MFMAs of group q)
// MODE 6  like 4 with the writes issued BEFORE the MFMAs (they only depend on the prefetched registers)
// MODE 7  wave tile 32x64 (two accumulator chains), 2 waves per workgroup-tile half... (TN = 2): 32 MFMAs per iteration,
//         64x128 workgroup tile, two LDS stages, one barrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

constexpr int LD = 36;

template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(float* out, unsigned long long* stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TN = MODE == 7 ? 2 : 1;
    constexpr int ROWS = 64 + 64 * TN;
    constexpr int STAGE = ROWS * LD;
    constexpr int NSTAGE = (MODE == 4 || MODE == 6 || MODE == 7) ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int kv = tid & 7, r0 = tid >> 3;
    // fill LDS with something non-trivial
    for (int i = tid; i < NSTAGE * STAGE; i += 256) lds[i] = (float)((i * 37 + blockIdx.x) % 97) * 0.01f - 0.4f;
    __syncthreads();
    f32x16 acc[TN];
    for (int u = 0; u < TN; ++u)
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;
    f32x4 st[2 + 2 * TN];
    for (int i = 0; i < 2 + 2 * TN; ++i) st[i] = f32x4{0.1f * tid, 0.2f, 0.3f, 0.4f + i};
    const float* As = lds + (wm * 32 + l31) * LD + lh * 4;
    const float* Bs = lds + 64 * LD + (wn * 32 * TN + l31) * LD + lh * 4;
    float* Ws = lds + r0 * LD + kv * 4;

    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0t = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const int cur = (MODE == 4 || MODE == 6 || MODE == 7) ? (it & 1) * STAGE : 0;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[0][j & 3], st[1][j & 3], acc[0], 0, 0, 0);
        } else if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 fa = *reinterpret_cast<const f32x4*>(As + cur + q * 8);
                f32x4 fb = *reinterpret_cast<const f32x4*>(Bs + cur + q * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc[0], 0, 0, 0);
            }
        } else if (MODE == 3 || MODE == 4 || MODE == 6) {
            f32x4 fa[4], fb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                fa[q] = *reinterpret_cast<const f32x4*>(As + cur + q * 8);
                fb[q] = *reinterpret_cast<const f32x4*>(Bs + cur + q * 8);
            }
            if (MODE == 6) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Ws + (STAGE - cur) + i * 32 * LD) = st[i];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][j], fb[q][j], acc[0], 0, 0, 0);
        } else if (MODE == 5) {
            f32x4 fa[2], fb[2];
            fa[0] = *reinterpret_cast<const f32x4*>(As);
            fb[0] = *reinterpret_cast<const f32x4*>(Bs);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < 3) {
                    fa[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(As + (q + 1) * 8);
                    fb[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(Bs + (q + 1) * 8);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][j], fb[q & 1][j], acc[0], 0, 0, 0);
            }
        } else if (MODE == 7) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 fa = *reinterpret_cast<const f32x4*>(As + cur + q * 8);
                f32x4 fb0 = *reinterpret_cast<const f32x4*>(Bs + cur + q * 8);
                f32x4 fb1 = *reinterpret_cast<const f32x4*>(Bs + cur + 32 * LD + q * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb0[j], acc[0], 0, 0, 0);
                    acc[TN - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb1[j], acc[TN - 1], 0, 0, 0);
                }
            }
        }
        if (MODE == 2 || MODE == 3 || MODE == 5) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Ws + i * 32 * LD) = st[i];
            __syncthreads();
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Ws + (STAGE - cur) + i * 32 * LD) = st[i];
            __syncthreads();
        } else if (MODE == 6) {
            __syncthreads();
        } else if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 2 + 2 * TN; ++i) *reinterpret_cast<f32x4*>(Ws + (STAGE - cur) + i * 32 * LD) = st[i];
            __syncthreads();
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1t = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int u = 0; u < TN; ++u)
        for (int e = 0; e < 16; ++e) s += acc[u][e];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1t - r0t;
    }
}

template <int MODE>
void run(int resident, int iters, int ncu, float* dout, unsigned long long* dst) {
    constexpr int TN = MODE == 7 ? 2 : 1;
    const size_t need = ((MODE == 4 || MODE == 6 || MODE == 7) ? 2 : 1) * (64 + 64 * TN) * LD * sizeof(float);
    // dynamic LDS sized so that exactly `resident` workgroups fit a CU (160 KiB)
    size_t smem = (160 * 1024 / resident) & ~size_t(255);
    if (smem < need) smem = need;
    if (smem > 160 * 1024) smem = 160 * 1024;
    CK(hipFuncSetAttribute((const void*)loop_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int grid = ncu * resident;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(grid), dim3(256), smem, 0, dout, dst, 50);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(grid), dim3(256), smem, 0, dout, dst, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * grid);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int b = 0; b < grid; ++b) { cyc += (double)st[2 * b]; real += (double)st[2 * b + 1]; }
    cyc /= grid; real /= grid;
    const double ghz = cyc / (real * 10.0);  // memrealtime ticks at 100 MHz
    const double mfma_per_wave = (double)iters * 16 * TN;
    // one wave per SIMD per workgroup -> pipe cycles needed per SIMD = resident * mfma_per_wave * 64
    const double util = resident * mfma_per_wave * 64.0 / cyc;
    const double tflops = (double)grid * 4 * mfma_per_wave * 32 * 32 * 2 * 2 / (ms * 1e-3) / 1e12;
    printf("mode %d  resident %d  grid %5d  %8.3f ms  %7.1f TF/s  pipe util %5.1f %%  clock %.2f GHz  cycles/iter/wg %.0f\n", MODE, resident,
           grid, ms, tflops, 100.0 * util, ghz, cyc / iters);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 3000;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s: %d CUs\n", p.name, ncu);
    float* dout;
    unsigned long long* dst;
    CK(hipMalloc(&dout, (size_t)ncu * 8 * 256 * 4));
    CK(hipMalloc(&dst, (size_t)ncu * 8 * 2 * 8));
    if (argc > 2) {  // sustained single configuration: mode, resident
        const int mode = atoi(argv[2]), r = argc > 3 ? atoi(argv[3]) : 8;
        if (mode == 0) run<0>(r, iters, ncu, dout, dst);
        else if (mode == 1) run<1>(r, iters, ncu, dout, dst);
        else run<2>(r, iters, ncu, dout, dst);
        return 0;
    }
    const int res[] = {1, 2, 4, 6, 8};
    for (int r : res) run<0>(r, iters, ncu, dout, dst);
    for (int r : res) run<1>(r, iters, ncu, dout, dst);
    for (int r : res) run<2>(r, iters, ncu, dout, dst);
    for (int r : res) run<3>(r, iters, ncu, dout, dst);
    for (int r : res) run<5>(r, iters, ncu, dout, dst);
    const int res2[] = {1, 2, 3, 4};
    for (int r : res2) run<4>(r, iters, ncu, dout, dst);
    for (int r : res2) run<6>(r, iters, ncu, dout, dst);
    for (int r : res2) run<7>(r, iters, ncu, dout, dst);
    return 0;
}
