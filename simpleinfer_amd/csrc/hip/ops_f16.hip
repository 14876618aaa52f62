// ops_f16.hip -- fp16-storage versions of the HBM-bound operators that do arithmetic (BASELINE.json configs[3]).
// Pure data movement (upsample, concat slice copies, flatten of a 1x1 map) needs no fp16 kernels: the host passes
// those tensors to the fp32 copy kernels as half as many 4-byte words.  Here: activation, same-shape add / mul,
// max pool, uniform average pool and the fp32 <-> fp16 conversions at the graph boundary.  One lane owns a 16-byte
// vector of 8 channels; arithmetic is done in fp32 and rounded once on the store.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "si_hip.h"
#include "si_hip_internal.h"

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float act_f(int act, float v, float p) {
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

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool VEC>
__global__ void activation_h_kernel(int act, float ap, const half_t* __restrict__ in, size_t pixels, int c, int in_ld,
                                    half_t* __restrict__ out, int out_ld) {
    const int cv = VEC ? c / 8 : c;
    const size_t total = pixels * (size_t)cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cv;
        const int ch = (int)(i - p * cv);
        if (VEC) {
            f16x8 v = *reinterpret_cast<const f16x8*>(in + p * in_ld + ch * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = si_store_cast<half_t>(act_f(act, (float)v[k], ap));
            *reinterpret_cast<f16x8*>(out + p * out_ld + ch * 8) = v;
        } else {
            out[p * out_ld + ch] = si_store_cast<half_t>(act_f(act, (float)in[p * in_ld + ch], ap));
        }
    }
}

// UnaryOp with fp16 storage (round 5): the fp32 function of ops.hip on the widened value, one rounding on the way out
template <bool VEC>
__global__ void unary_h_kernel(int op, const half_t* __restrict__ in, size_t pixels, int c, int in_ld, half_t* __restrict__ out, int out_ld) {
    const int cv = VEC ? c / 8 : c;
    const size_t total = pixels * (size_t)cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cv;
        const int ch = (int)(i - p * cv);
        if (VEC) {
            f16x8 v = *reinterpret_cast<const f16x8*>(in + p * in_ld + ch * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = si_store_cast<half_t>(si_unary_apply(op, (float)v[k]));
            *reinterpret_cast<f16x8*>(out + p * out_ld + ch * 8) = v;
        } else {
            out[p * out_ld + ch] = si_store_cast<half_t>(si_unary_apply(op, (float)in[p * in_ld + ch]));
        }
    }
}

template <bool VEC>
__global__ void binary_same_h_kernel(int op, const half_t* __restrict__ a, int a_ld, const half_t* __restrict__ b, int b_ld,
                                     half_t* __restrict__ out, int out_ld, size_t pixels, int c) {
    const int cv = VEC ? c / 8 : c;
    const size_t total = pixels * (size_t)cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cv;
        const int ch = (int)(i - p * cv);
        if (VEC) {
            const f16x8 x = *reinterpret_cast<const f16x8*>(a + p * a_ld + ch * 8);
            const f16x8 y = *reinterpret_cast<const f16x8*>(b + p * b_ld + ch * 8);
            f16x8 r;
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = si_store_cast<half_t>((op == 0) ? (float)x[k] + (float)y[k] : (float)x[k] * (float)y[k]);
            *reinterpret_cast<f16x8*>(out + p * out_ld + ch * 8) = r;
        } else {
            const float x = (float)a[p * a_ld + ch], y = (float)b[p * b_ld + ch];
            out[p * out_ld + ch] = si_store_cast<half_t>((op == 0) ? x + y : x * y);
        }
    }
}

__global__ void binary_bcast_h_kernel(int op, const half_t* __restrict__ a, int a_ld, const half_t* __restrict__ sc, int s_ld,
                                      half_t* __restrict__ out, int out_ld, int n, size_t ppi, int c) {
    const size_t cv = (size_t)(c / 8);
    const size_t total = (size_t)n * ppi * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / cv;
        const int ch = (int)(i - p * cv) * 8;
        const size_t b = p / ppi;
        const f16x8 x = *reinterpret_cast<const f16x8*>(a + p * a_ld + ch);
        const f16x8 y = *reinterpret_cast<const f16x8*>(sc + b * s_ld + ch);
        f16x8 r;
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = si_store_cast<half_t>((op == 0) ? (float)x[k] + (float)y[k] : (float)x[k] * (float)y[k]);
        *reinterpret_cast<f16x8*>(out + p * out_ld + ch) = r;
    }
}

// window max is exact in any precision: compare the fp16 values directly
template <bool VEC>
__global__ void maxpool_h_kernel(const SiPool2dDesc d, const half_t* __restrict__ in, half_t* __restrict__ out) {
    const int cv = VEC ? d.c / 8 : d.c;
    const size_t total = (size_t)d.n * d.oh * d.ow * cv;
    const half_t lowest = (half_t)(-65504.0f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t t = i;
        const int ch = (int)(t % cv); t /= cv;
        const int x = (int)(t % d.ow); t /= d.ow;
        const int y = (int)(t % d.oh); t /= d.oh;
        const int b = (int)t;
        f16x8 m = {lowest, lowest, lowest, lowest, lowest, lowest, lowest, lowest};
        for (int ky = 0; ky < d.kh; ++ky) {
            const int yy = y * d.sh - d.pt + ky * d.dh;
            if ((unsigned)yy >= (unsigned)d.ih) continue;
            for (int kx = 0; kx < d.kw; ++kx) {
                const int xx = x * d.sw - d.pl + kx * d.dw;
                if ((unsigned)xx >= (unsigned)d.iw) continue;
                const half_t* p = in + ((size_t)(b * d.ih + yy) * d.iw + xx) * d.in_ld;
                if (VEC) {
                    const f16x8 v = *reinterpret_cast<const f16x8*>(p + ch * 8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
                } else {
                    m[0] = p[ch] > m[0] ? p[ch] : m[0];
                }
            }
        }
        half_t* o = out + ((size_t)(b * d.oh + y) * d.ow + x) * d.out_ld;
        if (VEC)
            *reinterpret_cast<f16x8*>(o + ch * 8) = m;
        else
            o[ch] = m[0];
    }
}

__global__ void avgpool_h_kernel(const half_t* __restrict__ in, int n, int ih, int iw, int c, int in_ld,
                                 half_t* __restrict__ out, int oh, int ow, int out_ld, int kh, int kw) {
    const size_t total = (size_t)n * oh * ow * c;
    const float denom = (float)(kh * kw);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t t = i;
        const int ch = (int)(t % c); t /= c;
        const int x = (int)(t % ow); t /= ow;
        const int y = (int)(t % oh); t /= oh;
        const int b = (int)t;
        float s = 0.0f;
        for (int ky = 0; ky < kh; ++ky)
            for (int kx = 0; kx < kw; ++kx)
                s += (float)in[((size_t)(b * ih + y * kh + ky) * iw + x * kw + kx) * in_ld + ch];
        out[((size_t)(b * oh + y) * ow + x) * out_ld + ch] = (half_t)(s / denom);
    }
}

// global average (output 1x1): see ops.hip global_avgpool_kernel
__global__ __launch_bounds__(256) void global_avgpool_h_kernel(const half_t* __restrict__ in, int pixels, int c, int in_ld,
                                                               half_t* __restrict__ out, int out_ld, int cw) {
    __shared__ float part[256];
    const int b = blockIdx.y;
    const int ch = blockIdx.x * cw + (threadIdx.x % cw);
    const int pg = threadIdx.x / cw, groups = 256 / cw;
    float s = 0.0f;
    if (ch < c) {
        const half_t* p = in + (size_t)b * pixels * in_ld + ch;
        for (int i = pg; i < pixels; i += groups) s += (float)p[(size_t)i * in_ld];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (pg == 0 && ch < c) {
        float t = 0.0f;
        for (int g = 0; g < groups; ++g) t += part[g * cw + (threadIdx.x % cw)];
        out[(size_t)b * out_ld + ch] = (half_t)(t / (float)pixels);
    }
}

__global__ void cvt_f32_f16_kernel(const float* __restrict__ in, size_t pixels, int c, int in_ld, half_t* __restrict__ out,
                                   int out_ld) {
    const size_t total = pixels * (size_t)c;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / c;
        const int ch = (int)(i - p * c);
        out[p * out_ld + ch] = (half_t)in[p * in_ld + ch];
    }
}

__global__ void cvt_f16_f32_kernel(const half_t* __restrict__ in, size_t pixels, int c, int in_ld, float* __restrict__ out,
                                   int out_ld) {
    const size_t total = pixels * (size_t)c;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / c;
        const int ch = (int)(i - p * c);
        out[p * out_ld + ch] = (float)in[p * in_ld + ch];
    }
}

}  // namespace

extern "C" {

int si_hip_activation_f16(int act, float act_param, const void* in, size_t pixels, int c, int in_ld, void* out, int out_ld,
                          si_stream_t stream) {
    if (!in || !out || c <= 0 || in_ld < c || out_ld < c) return SI_E_BADARG;
    if (act < SI_ACT_NONE || act > SI_ACT_LEAKYRELU) return SI_E_UNSUPPORTED;
    if (pixels == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const half_t* i = static_cast<const half_t*>(in);
    half_t* o = static_cast<half_t*>(out);
    const bool vec = (c % 8 == 0) && (in_ld % 8 == 0) && (out_ld % 8 == 0) && al16(in) && al16(out);
    if (vec)
        hipLaunchKernelGGL(activation_h_kernel<true>, dim3(si_grid_for(pixels * (size_t)(c / 8))), dim3(256), 0, s, act,
                           act_param, i, pixels, c, in_ld, o, out_ld);
    else
        hipLaunchKernelGGL(activation_h_kernel<false>, dim3(si_grid_for(pixels * (size_t)c)), dim3(256), 0, s, act, act_param,
                           i, pixels, c, in_ld, o, out_ld);
    return (int)hipGetLastError();
}

int si_hip_unary_f16(int op, const void* in, size_t pixels, int c, int in_ld, void* out, int out_ld, si_stream_t stream) {
    if (!in || !out || c <= 0 || in_ld < c || out_ld < c) return SI_E_BADARG;
    if (op < 0 || op > 17) return SI_E_UNSUPPORTED;
    if (pixels == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const half_t* i = static_cast<const half_t*>(in);
    half_t* o = static_cast<half_t*>(out);
    const bool vec = (c % 8 == 0) && (in_ld % 8 == 0) && (out_ld % 8 == 0) && al16(in) && al16(out);
    if (vec)
        hipLaunchKernelGGL(unary_h_kernel<true>, dim3(si_grid_for(pixels * (size_t)(c / 8))), dim3(256), 0, s, op, i, pixels, c, in_ld, o, out_ld);
    else
        hipLaunchKernelGGL(unary_h_kernel<false>, dim3(si_grid_for(pixels * (size_t)c)), dim3(256), 0, s, op, i, pixels, c, in_ld, o, out_ld);
    return (int)hipGetLastError();
}

int si_hip_binary_same_f16(int op, const void* a, int a_ld, const void* b, int b_ld, void* out, int out_ld, size_t pixels,
                           int c, si_stream_t stream) {
    if (!a || !b || !out || c <= 0 || a_ld < c || b_ld < c || out_ld < c) return SI_E_BADARG;
    if (op != 0 && op != 2) return SI_E_UNSUPPORTED;
    if (pixels == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (c % 8 == 0) && (a_ld % 8 == 0) && (b_ld % 8 == 0) && (out_ld % 8 == 0) && al16(a) && al16(b) && al16(out);
    if (vec)
        hipLaunchKernelGGL(binary_same_h_kernel<true>, dim3(si_grid_for(pixels * (size_t)(c / 8))), dim3(256), 0, s, op,
                           static_cast<const half_t*>(a), a_ld, static_cast<const half_t*>(b), b_ld, static_cast<half_t*>(out),
                           out_ld, pixels, c);
    else
        hipLaunchKernelGGL(binary_same_h_kernel<false>, dim3(si_grid_for(pixels * (size_t)c)), dim3(256), 0, s, op,
                           static_cast<const half_t*>(a), a_ld, static_cast<const half_t*>(b), b_ld, static_cast<half_t*>(out),
                           out_ld, pixels, c);
    return (int)hipGetLastError();
}

// out[b][p][c] = a[b][p][c] (op) s[b][c]: the squeeze-excite scale (BinaryOp with broadcast factors H, W on the second operand,
// reference src/layer/binary_op.cpp:52-94) with fp16 storage; computed in fp32, rounded once
int si_hip_binary_bcast_f16(int op, const void* a, int a_ld, const void* s, int s_ld, void* out, int out_ld, int n,
                            size_t pixels_per_image, int c, si_stream_t stream) {
    if (!a || !s || !out || c <= 0 || n < 0 || a_ld < c || s_ld < c || out_ld < c) return SI_E_BADARG;
    if (op != 0 && op != 2) return SI_E_UNSUPPORTED;
    if (n == 0 || pixels_per_image == 0) return 0;
    if (c % 8 != 0 || a_ld % 8 != 0 || s_ld % 8 != 0 || out_ld % 8 != 0 || !al16(a) || !al16(s) || !al16(out)) return SI_E_UNSUPPORTED;
    hipLaunchKernelGGL(binary_bcast_h_kernel, dim3(si_grid_for((size_t)n * pixels_per_image * (size_t)(c / 8))), dim3(256), 0, (hipStream_t)stream, op,
                       static_cast<const half_t*>(a), a_ld, static_cast<const half_t*>(s), s_ld, static_cast<half_t*>(out), out_ld, n,
                       pixels_per_image, c);
    return (int)hipGetLastError();
}

int si_hip_maxpool2d_f16(const SiPool2dDesc* d, const void* in, void* out, si_stream_t stream) {
    if (!d || !in || !out || d->c <= 0 || d->in_ld < d->c || d->out_ld < d->c) return SI_E_BADARG;
    if (d->kh <= 0 || d->kw <= 0 || d->sh <= 0 || d->sw <= 0 || d->dh <= 0 || d->dw <= 0) return SI_E_BADARG;
    const size_t total = (size_t)d->n * d->oh * d->ow;
    if (total == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (d->c % 8 == 0) && (d->in_ld % 8 == 0) && (d->out_ld % 8 == 0) && al16(in) && al16(out);
    if (vec)
        hipLaunchKernelGGL(maxpool_h_kernel<true>, dim3(si_grid_for(total * (size_t)(d->c / 8))), dim3(256), 0, s, *d,
                           static_cast<const half_t*>(in), static_cast<half_t*>(out));
    else
        hipLaunchKernelGGL(maxpool_h_kernel<false>, dim3(si_grid_for(total * (size_t)d->c)), dim3(256), 0, s, *d,
                           static_cast<const half_t*>(in), static_cast<half_t*>(out));
    return (int)hipGetLastError();
}

int si_hip_adaptive_avgpool2d_f16(const void* in, int n, int ih, int iw, int c, int in_ld, void* out, int oh, int ow,
                                  int out_ld, si_stream_t stream) {
    if (!in || !out || n <= 0 || c <= 0 || oh <= 0 || ow <= 0) return SI_E_BADARG;
    if (ih % oh != 0 || iw % ow != 0) return SI_E_UNSUPPORTED;  // uniform windows only, as the fp32 kernel
    if (oh == 1 && ow == 1) {
        int cw = 8;  // as the fp32 kernel: 8 channels x 32 pixel groups (latency, not bandwidth)
        while (cw > 1 && cw / 2 >= c) cw /= 2;
        hipLaunchKernelGGL(global_avgpool_h_kernel, dim3((c + cw - 1) / cw, n), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const half_t*>(in), ih * iw, c, in_ld, static_cast<half_t*>(out), out_ld, cw);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(avgpool_h_kernel, dim3(si_grid_for((size_t)n * oh * ow * c)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const half_t*>(in), n, ih, iw, c, in_ld, static_cast<half_t*>(out), oh, ow, out_ld, ih / oh,
                       iw / ow);
    return (int)hipGetLastError();
}

int si_hip_convert_f32_f16(const float* in, size_t pixels, int c, int in_ld, void* out, int out_ld, si_stream_t stream) {
    if (!in || !out || c <= 0 || in_ld < c || out_ld < c) return SI_E_BADARG;
    if (pixels == 0) return 0;
    hipLaunchKernelGGL(cvt_f32_f16_kernel, dim3(si_grid_for(pixels * (size_t)c)), dim3(256), 0, (hipStream_t)stream, in, pixels,
                       c, in_ld, static_cast<half_t*>(out), out_ld);
    return (int)hipGetLastError();
}

int si_hip_convert_f16_f32(const void* in, size_t pixels, int c, int in_ld, float* out, int out_ld, si_stream_t stream) {
    if (!in || !out || c <= 0 || in_ld < c || out_ld < c) return SI_E_BADARG;
    if (pixels == 0) return 0;
    hipLaunchKernelGGL(cvt_f16_f32_kernel, dim3(si_grid_for(pixels * (size_t)c)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const half_t*>(in), pixels, c, in_ld, out, out_ld);
    return (int)hipGetLastError();
}

}  // extern "C"
