// pool_chain.hip -- three chained 5x5 stride-1 pad-2 max pools (YOLOv5's SPPF: y1 = mp(x), y2 = mp(y1), y3 = mp(y2),
// reference src/layer/max_pool_2d.cpp:77-121 applied three times) in one launch.
//
// Run one after the other the three pools are latency-bound launches over 13 MB each (0.029 ms apiece at batch 32 for
// 25 window reads per output).  Here a workgroup owns one image and a few 16-byte channel vectors, keeps the whole
// H x W map of those channels in LDS, and produces every stage with a separable max (5 horizontal + 5 vertical reads per
// output): x is read from HBM once, each y is written once, nothing is re-read.  Max is exact in any type, and a window
// position outside the map is simply skipped (= the reference's padding with the lowest value).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "si_hip.h"
#include "si_hip_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

struct ChainArgs {
    const void* in;
    void* out[3];
    int h, w, c, in_ld;
    int out_ld[3];
    int cgv;  // 16-byte channel vectors per workgroup
};

template <typename VecT, int NE>
__device__ __forceinline__ VecT vmax(VecT a, VecT b) {
#pragma unroll
    for (int k = 0; k < NE; ++k) a[k] = b[k] > a[k] ? b[k] : a[k];
    return a;
}

// VecT: 4 floats or 8 halves (NE elements, ElemT each).  LDS: two [H*W][cgv] vector planes.
template <typename VecT, typename ElemT, int NE>
__global__ __launch_bounds__(256) void maxpool5_chain3_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
    const int hw = a.h * a.w;
    const int items = hw * a.cgv;
    VecT* const buf0 = reinterpret_cast<VecT*>(chain_smem);
    VecT* const buf1 = buf0 + items;
    const int img = blockIdx.y;
    const int cv0 = blockIdx.x * a.cgv;            // first channel vector of this workgroup
    const int cv_total = a.c / NE;
    const int tid = threadIdx.x;

    const ElemT* const in = static_cast<const ElemT*>(a.in) + (size_t)img * hw * a.in_ld;
    for (int i = tid; i < items; i += 256) {
        const int pix = i / a.cgv, cv = i - pix * a.cgv;
        VecT v;
#pragma unroll
        for (int k = 0; k < NE; ++k) v[k] = (ElemT)0;
        if (cv0 + cv < cv_total) v = *reinterpret_cast<const VecT*>(in + (size_t)pix * a.in_ld + (cv0 + cv) * NE);
        buf0[i] = v;
    }
    __syncthreads();

#pragma unroll 1
    for (int stage = 0; stage < 3; ++stage) {
        // horizontal 5-max: buf0 -> buf1
        for (int i = tid; i < items; i += 256) {
            const int pix = i / a.cgv;
            const int y = pix / a.w, x = pix - y * a.w;
            VecT m = buf0[i];
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                if (dx == 0) continue;
                if ((unsigned)(x + dx) < (unsigned)a.w) m = vmax<VecT, NE>(m, buf0[i + dx * a.cgv]);
            }
            buf1[i] = m;
        }
        __syncthreads();
        // vertical 5-max: buf1 -> buf0 (the next stage's input) and out[stage]
        ElemT* const out = static_cast<ElemT*>(a.out[stage]) + (size_t)img * hw * a.out_ld[stage];
        for (int i = tid; i < items; i += 256) {
            const int pix = i / a.cgv, cv = i - pix * a.cgv;
            const int y = pix / a.w;
            VecT m = buf1[i];
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy) {
                if (dy == 0) continue;
                if ((unsigned)(y + dy) < (unsigned)a.h) m = vmax<VecT, NE>(m, buf1[i + dy * a.w * a.cgv]);
            }
            buf0[i] = m;
            if (cv0 + cv < cv_total) *reinterpret_cast<VecT*>(out + (size_t)pix * a.out_ld[stage] + (cv0 + cv) * NE) = m;
        }
        __syncthreads();
    }
}

template <typename VecT, typename ElemT, int NE>
int launch_chain(const void* in, int n, int h, int w, int c, int in_ld, void* const out[3], const int out_ld[3], hipStream_t s) {
    if (!in || n <= 0 || h <= 0 || w <= 0 || c <= 0) return SI_E_BADARG;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (c % NE != 0 || in_ld % NE != 0 || in_ld < c || !al16(in)) return SI_E_UNSUPPORTED;
    for (int k = 0; k < 3; ++k)
        if (!out[k] || out_ld[k] % NE != 0 || out_ld[k] < c || !al16(out[k])) return SI_E_UNSUPPORTED;
    // as many channel vectors per workgroup as two map planes of them fit in 64 KB of LDS (at most 4)
    int cgv = 4;
    while (cgv > 1 && (size_t)2 * h * w * cgv * 16 > 64 * 1024) cgv /= 2;
    // ... and few enough that the launch has at least two workgroups per CU
    while (cgv > 1 && (long long)((c / NE + cgv - 1) / cgv) * n < 512) cgv /= 2;
    const size_t lds = (size_t)2 * h * w * cgv * 16;
    if (lds > 64 * 1024 || n > 65535) return SI_E_UNSUPPORTED;  // a map this large: three separate pools
    ChainArgs a;
    a.in = in;
    for (int k = 0; k < 3; ++k) {
        a.out[k] = out[k];
        a.out_ld[k] = out_ld[k];
    }
    a.h = h; a.w = w; a.c = c; a.in_ld = in_ld; a.cgv = cgv;
    const int cv_total = c / NE;
    hipLaunchKernelGGL((maxpool5_chain3_kernel<VecT, ElemT, NE>), dim3((cv_total + cgv - 1) / cgv, n), dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int si_hip_maxpool5_chain3_f32(const float* in, int n, int h, int w, int c, int in_ld, float* out1, int out1_ld, float* out2,
                               int out2_ld, float* out3, int out3_ld, si_stream_t stream) {
    void* const out[3] = {out1, out2, out3};
    const int ld[3] = {out1_ld, out2_ld, out3_ld};
    return launch_chain<f32x4, float, 4>(in, n, h, w, c, in_ld, out, ld, static_cast<hipStream_t>(stream));
}

int si_hip_maxpool5_chain3_f16(const void* in, int n, int h, int w, int c, int in_ld, void* out1, int out1_ld, void* out2,
                               int out2_ld, void* out3, int out3_ld, si_stream_t stream) {
    void* const out[3] = {out1, out2, out3};
    const int ld[3] = {out1_ld, out2_ld, out3_ld};
    return launch_chain<f16x8, _Float16, 8>(in, n, h, w, c, in_ld, out, ld, static_cast<hipStream_t>(stream));
}

}  // extern "C"
