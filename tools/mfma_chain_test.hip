// tools/mfma_chain_test.hip -- do the fp32 MFMA shapes of gfx950 round like ONE sequential fma chain over k?
//
// The engine's batch-invariance contract (an image's bits do not depend on the batch it rides in) lets the tile policy
// change with the batch only if every tile shape produces the same bits per output element.  Two kernels that both use
// v_mfma_f32_32x32x2_f32 and feed k in the same order trivially agree; this tool checks whether v_mfma_f32_16x16x4_f32
// (and the two-block 32x32x1 / four-block 16x16x1 forms) fed the same k order give the SAME bits, and whether all of them
// equal a scalar fmaf chain.  Development tool, not product code.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mct tools/mfma_chain_test.hip && /tmp/mct
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

constexpr int K = 96;

// A[32][K], B[32][K] (C = A * B^T), one wave
__global__ void k32x32x2(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + lh], B[l31 * K + k + lh], acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + l31] = acc[e];
}

// the four 16x16 quadrants of the same 32x32 product on v_mfma_f32_16x16x4_f32
__global__ void k16x16x4(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x, l15 = lane & 15, lq = lane >> 4;
    for (int qm = 0; qm < 2; ++qm)
        for (int qn = 0; qn < 2; ++qn) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < K; k += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(qm * 16 + l15) * K + k + lq], B[(qn * 16 + l15) * K + k + lq], acc, 0, 0, 0);
            for (int e = 0; e < 4; ++e) C[(qm * 16 + e + 4 * lq) * 32 + qn * 16 + l15] = acc[e];
        }
}

// v_mfma_f32_32x32x1_f32: two blocks, k = 1 per instruction; block b = lane >> 5 owns its own A / B column.  Used here with
// both blocks computing the SAME 32x32 product on alternating halves of K?  No: blocks are independent outputs (32 acc
// registers).  We give block 0 the product and block 1 the product with A negated, and read block 0.
typedef float f32x32 __attribute__((ext_vector_type(32)));
__global__ void k32x32x1(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    f32x32 acc;
    for (int e = 0; e < 32; ++e) acc[e] = 0.f;
    for (int k = 0; k < K; ++k) {
        const float av = A[l31 * K + k], bv = B[l31 * K + k];
        acc = __builtin_amdgcn_mfma_f32_32x32x1f32(lh ? -av : av, bv, acc, 0, 0, 0);
    }
    // D layout of the 2-block form: register e (0..15) of block 0 = rows as in 32x32x2, registers 16..31 = block 1
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + l31] = acc[e];
}

__global__ void kscalar(const float* A, const float* B, float* C) {
    const int m = threadIdx.x >> 5, n = threadIdx.x & 31;
    for (int mm = m; mm < 32; mm += blockDim.x >> 5) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = __builtin_fmaf(A[mm * K + k], B[n * K + k], acc);
        C[mm * 32 + n] = acc;
    }
}

// pairwise form: (a0*b0 + a1*b1) with one rounding each?  acc' = fma(a1, b1, fma(a0, b0, acc)) is the chain; alternatives a
// hardware could implement: acc + (a0*b0 + a1*b1) exact-then-round (a fused dot-2).  Emulated in double to tell them apart.
static void host_models(const std::vector<float>& A, const std::vector<float>& B, std::vector<float>& chain, std::vector<float>& dot2,
                        std::vector<float>& dot4) {
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            float c = 0.f;
            for (int k = 0; k < K; ++k) c = __builtin_fmaf(A[m * K + k], B[n * K + k], c);
            chain[m * 32 + n] = c;
            float d = 0.f;
            for (int k = 0; k < K; k += 2) {
                // products of two floats are exact in double; the sum of two doubles + a float acc is not always exact in
                // double, but close enough to expose a fused dot-2 against a chain on random data
                const double s = (double)A[m * K + k] * B[n * K + k] + (double)A[m * K + k + 1] * B[n * K + k + 1] + (double)d;
                d = (float)s;
            }
            dot2[m * 32 + n] = d;
            float q = 0.f;
            for (int k = 0; k < K; k += 4) {
                double s = (double)q;
                for (int j = 0; j < 4; ++j) s += (double)A[m * K + k + j] * B[n * K + k + j];
                q = (float)s;
            }
            dot4[m * 32 + n] = q;
        }
}

static int diff(const char* what, const std::vector<float>& x, const std::vector<float>& y) {
    int n = 0;
    double worst = 0;
    for (size_t i = 0; i < x.size(); ++i)
        if (memcmp(&x[i], &y[i], 4) != 0) {
            ++n;
            const double r = fabs((double)x[i] - y[i]) / (fabs((double)y[i]) + 1e-30);
            if (r > worst) worst = r;
        }
    printf("%-44s %4d / %zu elements differ (worst rel %.2e)\n", what, n, x.size(), worst);
    return n;
}

int main() {
    std::vector<float> A(32 * K), B(32 * K);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0);
    };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd() * 3.0f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 32 * 32 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> c322(1024), c164(1024), c321(1024), csc(1024), chain(1024), dot2(1024), dot4(1024);
    auto run = [&](auto kern, int threads, std::vector<float>& out) {
        CK(hipMemset(dC, 0, 4096));
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, dA, dB, dC);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dC, 4096, hipMemcpyDeviceToHost));
    };
    run(k32x32x2, 64, c322);
    run(k16x16x4, 64, c164);
    run(k32x32x1, 64, c321);
    run(kscalar, 256, csc);
    host_models(A, B, chain, dot2, dot4);
    diff("device scalar fmaf chain vs host fmaf chain", csc, chain);
    diff("mfma 32x32x2 vs fmaf chain", c322, chain);
    diff("mfma 32x32x2 vs fused dot-2 model", c322, dot2);
    diff("mfma 16x16x4 vs fmaf chain", c164, chain);
    diff("mfma 16x16x4 vs fused dot-4 model", c164, dot4);
    diff("mfma 16x16x4 vs mfma 32x32x2", c164, c322);
    diff("mfma 32x32x1 (block 0) vs fmaf chain", c321, chain);
    diff("mfma 32x32x1 (block 0) vs mfma 32x32x2", c321, c322);
    return 0;
}
