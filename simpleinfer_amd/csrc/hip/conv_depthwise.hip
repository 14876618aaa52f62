// conv_depthwise.hip -- depthwise convolution (groups == in_channels == out_channels), NHWC fp32.
//
// The reference runs grouped convolutions through Conv2d::ForwardIm2ColWithGroup (src/layer/conv_2d.cpp:285-380): one
// Eigen patch-extraction + contraction expression PER GROUP, i.e. per channel for MobileNet's depthwise layers -- its
// slowest path.  On the GPU a depthwise layer has 2*kh*kw FLOPs per output element and no reuse across channels, so it
// is bound by HBM, not by the matrix cores: no MFMA, no GEMM reshaping.  One lane owns a 16-byte vector of 4 channels
// (consecutive lanes -> consecutive channel vectors -> coalesced 16-byte accesses) and TW consecutive output columns,
// so the kh*kw filter vectors live in registers and overlapping input columns are loaded once.  Bias / activation /
// residual are fused as in the implicit-GEMM kernel.  Algorithmic bytes: 4 * (N*H*W*C + N*OH*OW*C).
#include <hip/hip_runtime.h>

#include "si_hip.h"
#include "si_hip_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct DwArgs {
    const float* in;
    const float* w;     // [kh*kw][c] (channels contiguous)
    const float* bias;
    const float* res;
    float* out;
    int n, ih, iw, c, in_ld;
    int oh, ow, out_ld, res_ld;
    int kh, kw, sh, sw, dh, dw, pt, pl;
    int act1, act2;
    float act_param;
};

__device__ __forceinline__ float dw_act(int act, float v, float p) {
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

// Specialisation for dilation 1 and compile-time kernel width / column stride (3x3 and 5x5, stride 1 and 2: every
// MobileNet depthwise layer): the (TW-1)*SW + KW distinct input columns of a kernel row are loaded ONCE into registers
// and every tap reads them from there -- 6 loads instead of 12 per row for 3x3 s1.
template <int TW, int KW, int SW>
__global__ __launch_bounds__(256) void conv_depthwise_cols_kernel(const DwArgs a) {
    constexpr int NCOL = (TW - 1) * SW + KW;
    const int cvn = a.c / 4;
    const int wt = (a.ow + TW - 1) / TW;
    const size_t total = (size_t)a.n * a.oh * wt * cvn;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t t = i;
        const int cv = (int)(t % cvn); t /= cvn;
        const int xt = (int)(t % wt); t /= wt;
        const int oy = (int)(t % a.oh); t /= a.oh;
        const int b = (int)t;
        const int ch = cv * 4;
        const int ox0 = xt * TW;
        const int x0 = ox0 * SW - a.pl;

        f32x4 acc[TW];
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < a.kh; ++ky) {
            const int y = oy * a.sh - a.pt + ky;
            if ((unsigned)y >= (unsigned)a.ih) continue;
            const float* row = a.in + (size_t)(b * a.ih + y) * a.iw * a.in_ld + ch;
            f32x4 col[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                const int x = x0 + q;
                col[q] = (unsigned)x < (unsigned)a.iw ? *reinterpret_cast<const f32x4*>(row + (size_t)x * a.in_ld)
                                                      : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const float* wp = a.w + (size_t)(ky * KW) * a.c + ch;
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + (size_t)kx * a.c);
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const f32x4 v = col[j * SW + kx];
                    acc[j].x = fmaf(v.x, wv.x, acc[j].x);
                    acc[j].y = fmaf(v.y, wv.y, acc[j].y);
                    acc[j].z = fmaf(v.z, wv.z, acc[j].z);
                    acc[j].w = fmaf(v.w, wv.w, acc[j].w);
                }
            }
        }
        const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int ox = ox0 + j;
            if (ox >= a.ow) break;
            const size_t pix = (size_t)(b * a.oh + oy) * a.ow + ox;
            f32x4 v = acc[j] + bv;
            const f32x4 r = a.res ? *reinterpret_cast<const f32x4*>(a.res + pix * a.res_ld + ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float e = dw_act(a.act1, v[k], a.act_param);
                if (a.res) e += r[k];
                v[k] = dw_act(a.act2, e, a.act_param);
            }
            *reinterpret_cast<f32x4*>(a.out + pix * a.out_ld + ch) = v;
        }
    }
}

// Generic path (any kernel size / stride / dilation).
// VEC: 4 channels per lane (c, strides multiples of 4, 16-byte aligned bases); otherwise 1 channel per lane.
// TW output columns per lane.  Accumulation order: taps row-major, fmaf chain (the oracle's order).
template <bool VEC, int TW>
__global__ __launch_bounds__(256) void conv_depthwise_kernel(const DwArgs a) {
    constexpr int CV = VEC ? 4 : 1;
    const int cvn = a.c / CV;
    const int wt = (a.ow + TW - 1) / TW;
    const size_t total = (size_t)a.n * a.oh * wt * cvn;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t t = i;
        const int cv = (int)(t % cvn); t /= cvn;
        const int xt = (int)(t % wt); t /= wt;
        const int oy = (int)(t % a.oh); t /= a.oh;
        const int b = (int)t;
        const int ch = cv * CV;
        const int ox0 = xt * TW;

        f32x4 acc[TW];
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int ky = 0; ky < a.kh; ++ky) {
            const int y = oy * a.sh - a.pt + ky * a.dh;
            if ((unsigned)y >= (unsigned)a.ih) continue;
            const float* row = a.in + (size_t)(b * a.ih + y) * a.iw * a.in_ld + ch;
            for (int kx = 0; kx < a.kw; ++kx) {
                f32x4 wv;
                const float* wp = a.w + (size_t)(ky * a.kw + kx) * a.c + ch;
                if (VEC) wv = *reinterpret_cast<const f32x4*>(wp);
                else wv = (f32x4){wp[0], 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int x = (ox0 + j) * a.sw - a.pl + kx * a.dw;
                    if ((unsigned)x < (unsigned)a.iw && ox0 + j < a.ow) {
                        const float* p = row + (size_t)x * a.in_ld;
                        if (VEC) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(p);
                            acc[j].x = fmaf(v.x, wv.x, acc[j].x);
                            acc[j].y = fmaf(v.y, wv.y, acc[j].y);
                            acc[j].z = fmaf(v.z, wv.z, acc[j].z);
                            acc[j].w = fmaf(v.w, wv.w, acc[j].w);
                        } else {
                            acc[j].x = fmaf(p[0], wv.x, acc[j].x);
                        }
                    }
                }
            }
        }

        f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            if (VEC) bv = *reinterpret_cast<const f32x4*>(a.bias + ch);
            else bv.x = a.bias[ch];
        }
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int ox = ox0 + j;
            if (ox >= a.ow) break;
            const size_t pix = (size_t)(b * a.oh + oy) * a.ow + ox;
            f32x4 v = acc[j] + bv;
            f32x4 r = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (a.res) {
                if (VEC) r = *reinterpret_cast<const f32x4*>(a.res + pix * a.res_ld + ch);
                else r.x = a.res[pix * a.res_ld + ch];
            }
#pragma unroll
            for (int k = 0; k < CV; ++k) {
                float e = dw_act(a.act1, v[k], a.act_param);
                if (a.res) e += r[k];
                v[k] = dw_act(a.act2, e, a.act_param);
            }
            float* o = a.out + pix * a.out_ld + ch;
            if (VEC) *reinterpret_cast<f32x4*>(o) = v;
            else o[0] = v.x;
        }
    }
}

}  // namespace

// shape-only: decides the weight layout too
bool si_conv_depthwise_ok(const SiConv2dDesc* d) {
    return d && d->groups > 1 && d->groups == d->ic && d->ic == d->oc && d->kh > 0 && d->kw > 0;
}

size_t si_conv_depthwise_weight_elems(const SiConv2dDesc* d) { return (size_t)d->kh * d->kw * d->ic; }

// OIHW with I = 1: w[c][0][kh][kw] -> [kh*kw][c]
void si_conv_depthwise_pack(const SiConv2dDesc* d, const float* w_oihw, float* w_packed) {
    const int taps = d->kh * d->kw;
    for (int c = 0; c < d->ic; ++c)
        for (int t = 0; t < taps; ++t) w_packed[(size_t)t * d->ic + c] = w_oihw[(size_t)c * taps + t];
}

const char* si_conv_depthwise_name(const SiConv2dDesc* d) {
    return (d->ic % 4 == 0) ? "conv_depthwise_kernel<true, 4>" : "conv_depthwise_kernel<false, 4>";
}

int si_conv_depthwise_launch(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                             const float* residual, float* out, hipStream_t s) {
    DwArgs a;
    a.in = in; a.w = w_packed; a.bias = d->has_bias ? bias : nullptr; a.res = d->has_residual ? residual : nullptr; a.out = out;
    a.n = d->n; a.ih = d->ih; a.iw = d->iw; a.c = d->ic; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.dh = d->dh; a.dw = d->dw; a.pt = d->pt; a.pl = d->pl;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = (d->ic % 4 == 0) && (d->in_ld % 4 == 0) && (d->out_ld % 4 == 0) && al16(in) && al16(out) && al16(w_packed) &&
                     (!a.bias || al16(a.bias)) && (!a.res || (d->res_ld % 4 == 0 && al16(a.res)));
    constexpr int TW = 4;
    const size_t items = (size_t)d->n * d->oh * ((d->ow + TW - 1) / TW) * (vec ? d->ic / 4 : d->ic);
    if (items == 0) return 0;
    if (vec && d->dh == 1 && d->dw == 1 && (d->kw == 3 || d->kw == 5) && (d->sw == 1 || d->sw == 2)) {
        const dim3 g(si_grid_for(items)), blk(256);
        if (d->kw == 3 && d->sw == 1) hipLaunchKernelGGL((conv_depthwise_cols_kernel<TW, 3, 1>), g, blk, 0, s, a);
        else if (d->kw == 3) hipLaunchKernelGGL((conv_depthwise_cols_kernel<TW, 3, 2>), g, blk, 0, s, a);
        else if (d->sw == 1) hipLaunchKernelGGL((conv_depthwise_cols_kernel<TW, 5, 1>), g, blk, 0, s, a);
        else hipLaunchKernelGGL((conv_depthwise_cols_kernel<TW, 5, 2>), g, blk, 0, s, a);
        return (int)hipGetLastError();
    }
    if (vec)
        hipLaunchKernelGGL((conv_depthwise_kernel<true, TW>), dim3(si_grid_for(items)), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv_depthwise_kernel<false, TW>), dim3(si_grid_for(items)), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
