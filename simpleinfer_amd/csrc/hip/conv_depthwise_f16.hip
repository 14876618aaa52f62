// conv_depthwise_f16.hip -- depthwise convolution (groups == in_channels == out_channels) with fp16 activations: the fp16-storage
// twin of conv_depthwise.hip (reference: Conv2d::ForwardIm2ColWithGroup, src/layer/conv_2d.cpp:285-380, one Eigen expression per
// channel; fp32 only -- the yardstick of this path is the fp32 oracle on fp16-rounded operands).  2*kh*kw FLOPs per output element
// and no reuse across channels: HBM-bound, no MFMA.  One lane owns a 16-byte vector of EIGHT channels and TW consecutive output
// columns; weights and bias stay fp32 (the fp32 depthwise layout [kh*kw][c], a few KB), products and the tap sum are fp32 fma
// chains in tap order, bias / activation / residual are fused, and every output is rounded to fp16 once (si_store_cast).
// Algorithmic bytes: 2 * (N*H*W*C + N*OH*OW*C).
#include <hip/hip_runtime.h>

#include "si_hip.h"
#include "si_hip_internal.h"

#pragma clang fp contract(off)

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct DwArgsH {
    const half_t* in;
    const float* w;     // [kh*kw][c] fp32 (si_hip_conv2d_pack_weight_host of the depthwise descriptor)
    const float* bias;
    const half_t* res;
    half_t* out;
    int n, ih, iw, c, in_ld;
    int oh, ow, out_ld, res_ld;
    int kh, kw, sh, sw, dh, dw, pt, pl;
    int act1, act2;
    float act_param;
};

__device__ __forceinline__ float dwh_act(int act, float v, float p) {
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

template <int TW>
__global__ __launch_bounds__(256) void conv_depthwise_f16_kernel(const DwArgsH a) {
    const int cvn = a.c / 8;
    const int wt = (a.ow + TW - 1) / TW;
    const size_t total = (size_t)a.n * a.oh * wt * cvn;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t t = i;
        const int cv = (int)(t % cvn); t /= cvn;
        const int xt = (int)(t % wt); t /= wt;
        const int oy = (int)(t % a.oh); t /= a.oh;
        const int b = (int)t;
        const int ch = cv * 8;
        const int ox0 = xt * TW;

        float acc[TW][8];
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[j][k] = 0.0f;

        for (int ky = 0; ky < a.kh; ++ky) {
            const int y = oy * a.sh - a.pt + ky * a.dh;
            if ((unsigned)y >= (unsigned)a.ih) continue;
            const half_t* row = a.in + (size_t)(b * a.ih + y) * a.iw * a.in_ld + ch;
            for (int kx = 0; kx < a.kw; ++kx) {
                const float* wp = a.w + (size_t)(ky * a.kw + kx) * a.c + ch;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int x = (ox0 + j) * a.sw - a.pl + kx * a.dw;
                    if ((unsigned)x < (unsigned)a.iw && ox0 + j < a.ow) {
                        const f16x8 v = *reinterpret_cast<const f16x8*>(row + (size_t)x * a.in_ld);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            acc[j][k] = __builtin_fmaf((float)v[k], w0[k], acc[j][k]);
                            acc[j][4 + k] = __builtin_fmaf((float)v[4 + k], w1[k], acc[j][4 + k]);
                        }
                    }
                }
            }
        }

        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + ch), b1 = *reinterpret_cast<const f32x4*>(a.bias + ch + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { bv[k] = b0[k]; bv[4 + k] = b1[k]; }
        }
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int ox = ox0 + j;
            if (ox >= a.ow) break;
            const size_t pix = (size_t)(b * a.oh + oy) * a.ow + ox;
            f16x8 r = {};
            if (a.res) r = *reinterpret_cast<const f16x8*>(a.res + pix * a.res_ld + ch);
            f16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float e = dwh_act(a.act1, acc[j][k] + bv[k], a.act_param);
                if (a.res) e += (float)r[k];
                o[k] = si_store_cast<half_t>(dwh_act(a.act2, e, a.act_param));
            }
            *reinterpret_cast<f16x8*>(a.out + pix * a.out_ld + ch) = o;
        }
    }
}

}  // namespace

bool si_conv_depthwise_f16_ok(const SiConv2dDesc* d) {
    return d && d->groups > 1 && d->groups == d->ic && d->ic == d->oc && d->kh > 0 && d->kw > 0 && d->ic % 8 == 0;
}

extern "C" int si_hip_conv2d_depthwise_f16(const SiConv2dDesc* d, const void* in, const float* w_packed, const float* bias,
                                           const void* residual, void* out, si_stream_t stream) {
    if (!d || !in || !w_packed || !out) return SI_E_BADARG;
    if (!si_conv_depthwise_f16_ok(d)) return SI_E_UNSUPPORTED;
    if (d->n <= 0 || d->oh <= 0 || d->ow <= 0) return SI_E_BADARG;
    if (d->has_bias && !bias) return SI_E_BADARG;
    if (d->has_residual && !residual) return SI_E_BADARG;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (d->in_ld % 8 != 0 || d->out_ld % 8 != 0 || !al16(in) || !al16(out) || !al16(w_packed) || (d->has_bias && !al16(bias)) ||
        (d->has_residual && (d->res_ld % 8 != 0 || !al16(residual))))
        return SI_E_UNSUPPORTED;
    DwArgsH a;
    a.in = static_cast<const half_t*>(in); a.w = w_packed; a.bias = d->has_bias ? bias : nullptr;
    a.res = d->has_residual ? static_cast<const half_t*>(residual) : nullptr; a.out = static_cast<half_t*>(out);
    a.n = d->n; a.ih = d->ih; a.iw = d->iw; a.c = d->ic; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.dh = d->dh; a.dw = d->dw; a.pt = d->pt; a.pl = d->pl;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    constexpr int TW = 2;
    const size_t items = (size_t)d->n * d->oh * ((d->ow + TW - 1) / TW) * (d->ic / 8);
    hipLaunchKernelGGL((conv_depthwise_f16_kernel<TW>), dim3(si_grid_for(items)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}
