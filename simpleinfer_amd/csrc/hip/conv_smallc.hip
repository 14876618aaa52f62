// conv_smallc.hip -- first-layer ("stem") convolution: 1..3 input channels (RGB), large kernel, stride 2.
//
// YOLOv5s conv0 is 640x640x3 -> 320x320x32, 6x6 s2 p2 (reference shape: test/test_layer/test_conv_2d.cpp:279-293);
// ResNet18 conv1 is 224x224x3 -> 112x112x64, 7x7 s2 p3.  With 3 channels the im2col row of one tap is 12 bytes, so
// the implicit-GEMM kernel's 16-byte channel vectors do not apply (its scalar gather ran at 37 TFLOP/s).
//
// Here NHWC is exploited the other way round: for one output pixel and one kernel ROW, the KW taps x C channels are
// KW*C CONTIGUOUS floats of the input row, and neighbouring output pixels overlap in all but stride*C of them.
// A persistent workgroup walks over (image, output-row block, column tile) items and for each one
//   1. has the KH + (RB-1)*stride input rows it needs in LDS, staged with coalesced loads (each input byte is
//      fetched ~1.5x instead of KH*KW/stride^2 = 9x); the loads for item i+1 are issued before the MFMAs of item i
//      (register prefetch, committed to LDS after the MFMAs), and the stores of item i drain while item i+1 computes;
//   2. runs 32x32x2 MFMAs whose A operand is read straight from the staged rows,
//        A[m][k] = row[ky][(m*sw)*C + j],  j = kx*C + c,  k = (ky, j)   -- no im2col tile anywhere,
//      and whose B operand comes from a pre-transposed weight image [ky][j][oc] through the vector L1 (13.8 KB for
//      the YOLOv5 stem: every wave of the CU reads the same lines).
// One wave owns 32 consecutive output pixels of one output row and NT*32 output channels.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "si_hip.h"
#include "si_hip_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// conv_stem_roll.hip: the rolling-window kernel that serves the RGB stride-2 stems (6x6, 7x7, 3x3; <= 64 output channels);
// this file's kernel remains for every other 1..3-channel shape (other strides / kernel sizes, wider outputs)
bool si_conv_stemroll_ok(const SiConv2dDesc* d);
size_t si_conv_stemroll_weight_elems(const SiConv2dDesc* d);
void si_conv_stemroll_pack(const SiConv2dDesc* d, const float* w_oihw, float* w_packed);
const char* si_conv_stemroll_name(const SiConv2dDesc* d);
int si_conv_stemroll_launch(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                            const float* residual, float* out, hipStream_t s);

// which of the two a shape goes to is a function of the shape alone (it fixes the weight layout); SI_STEM_ROLL=0 is a
// development switch that sends everything to the old kernel (read once per process)
static bool stem_roll(const SiConv2dDesc* d) {
    static const bool enabled = SI_ENV_INT("SI_STEM_ROLL", 1) != 0;
    return enabled && si_conv_stemroll_ok(d);
}

namespace {

struct SmallCArgs {
    const float* in;
    const float* w;     // transposed image [kh][2*HP][ocp], rows j >= kw*c are zero (see si_conv_smallc_pack)
    const float* bias;
    const float* res;
    void* out;          // OutT (float, or _Float16 for the fp16 engine path)
    int n, ih, iw, c, in_ld;
    int oh, ow, oc, ocp, out_ld, res_ld;
    int kh, kw, sh, sw, pt, pl;
    int row_len;        // floats staged per input row
    int n_in_rows;
    int w_tiles, oc_tiles, row_blocks;
    int items;          // n * row_blocks * w_tiles (per channel tile)
    unsigned in_bytes;  // extent of the input tensor for the buffer resource (< 4 GB)
    int act1, act2;
    float act_param;
};

__device__ __forceinline__ float act_any(int act, float v, float p) {
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

// NW waves per workgroup (32*NW output pixels along W), NT 32-wide output-channel tiles per wave, RB output rows per
// item, HP = (kw*c rounded up to even)/2 MFMA steps per kernel row (compile time: 9 for 6x6x3, 11 for 7x7x3),
// PF = prefetch registers per thread (>= n_in_rows*row_len / (64*NW)).
template <int NW, int NT, int RB, int HP, int PF, typename OutT, bool VEC>
__global__ __launch_bounds__(NW * 64) void conv_smallc_rows_kernel(const SmallCArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NTHR = NW * 64;
    constexpr int TOW = 32 * NW;
    const int buf_len = a.n_in_rows * a.row_len;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;

    f32x4 pre[PF / 4];

    // item -> coordinates (column tile fastest, then row block, image); the channel tile is blockIdx.y, so bias and
    // everything else that is loaded from global memory besides the row prefetch is read once, before the item loop
    const int oc0 = blockIdx.y * 32 * NT;
    auto decode = [&](int item, int& img, int& oy0, int& ox0) {
        int t = item;
        const int wt = t % a.w_tiles; t /= a.w_tiles;
        const int rbk = t % a.row_blocks; t /= a.row_blocks;
        img = t; oy0 = rbk * RB; ox0 = wt * TOW;
    };

    // Issue the global loads of one item's input rows into registers (zero outside the image).  The staged window starts
    // at a multiple of 4 floats of the image row; thread t holds floats [4t, 4t+4) of every staged row.  VEC (dense image,
    // width*channels a multiple of 4, 16-byte aligned base): one 16-byte load per row -- a vector lies inside its row or
    // entirely outside, and then an offset outside the buffer makes the hardware return zeros.  Otherwise element-wise
    // loads with per-element bounds.  The values are not touched until they are committed to LDS after the MFMAs.
    constexpr int MAX_ROWS = PF / 4;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
    const int row_floats = a.iw * a.c;
    const bool mine = 4 * tid < a.row_len;
    auto prefetch = [&](int item) {
        int img, oy0, ox0;
        decode(item, img, oy0, ox0);
        const int x0 = (ox0 * a.sw - a.pl) * a.c, iy0 = oy0 * a.sh - a.pt;
        const int e = (x0 & ~3) + 4 * tid;
#pragma unroll
        for (int r = 0; r < MAX_ROWS; ++r) {
            const int y = iy0 + r;
            const bool yok = mine && r < a.n_in_rows && (unsigned)y < (unsigned)a.ih;
            if (VEC) {
                unsigned off = ((unsigned)((img * a.ih + y) * row_floats + e)) * 4u;  // modulo 2^32
                if (!(yok && e >= 0 && e + 3 < row_floats)) off = 0xFFFFFF00u;
                pre[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0));
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ee = e + t;
                    const int px = ee / a.c, ch = ee - px * a.c;
                    unsigned off = ((unsigned)(((img * a.ih + y) * a.iw + px) * a.in_ld + ch)) * 4u;
                    if (!(yok && ee >= 0 && ee < row_floats)) off = 0xFFFFFF00u;
                    pre[r][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, off, 0, 0));
                }
            }
        }
    };
    auto commit = [&](float* buf) {
        if (mine) {
#pragma unroll
            for (int r = 0; r < MAX_ROWS; ++r)
                if (r < a.n_in_rows) *reinterpret_cast<f32x4*>(buf + r * a.row_len + 4 * tid) = pre[r];
        }
    };

    int item = blockIdx.x;
    if (item >= a.items) return;
    prefetch(item);
    commit(smem);
    // the layer's weight image (all output channels), once per persistent workgroup: straight 16-byte copy
    float* wl = smem + buf_len;
    {
        const int nvec = a.kh * 2 * HP * a.ocp / 4;
        const float4* src = reinterpret_cast<const float4*>(a.w);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = tid; i < nvec; i += NTHR) dst[i] = src[i];
    }
    float bvv[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int o = oc0 + u * 32 + l31;
        bvv[u] = (a.bias && o < a.oc) ? a.bias[o] : 0.0f;
    }
    // nothing but the next item's rows is pending inside the loop (a load still in flight at the loop head makes the
    // compiler wait for ALL outstanding loads -- the prefetch included -- at its first use)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();

    const int px_off = (wave * 32 + l31) * a.sw * a.c + lh;  // lane half h reads element 2*jj + h of a kernel row
    for (; item < a.items; item += gridDim.x) {
        const int next = item + gridDim.x;
        if (next < a.items) prefetch(next);

        int img, oy0, ox0;
        decode(item, img, oy0, ox0);
        const float* rows = smem + (((ox0 * a.sw - a.pl) * a.c) & 3);  // the item's first pixel within the staged window

#pragma unroll 1
        for (int rb = 0; rb < RB; ++rb) {
            const int oy = oy0 + rb;
            if (oy >= a.oh) break;
            f32x16 acc[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[u][e] = 0.0f;

            const float* rbase = rows + rb * a.sh * a.row_len + px_off;
            const float* wbase = wl + lh * a.ocp + oc0 + l31;
            for (int ky = 0; ky < a.kh; ++ky) {
                float av[HP], bv[HP][NT];
#pragma unroll
                for (int jj = 0; jj < HP; ++jj) {
                    av[jj] = rbase[2 * jj];  // elements past kw*c belong to the next pixel: finite, and their weights are 0
#pragma unroll
                    for (int u = 0; u < NT; ++u) bv[jj][u] = wbase[(2 * jj) * a.ocp + u * 32];
                }
#pragma unroll
                for (int jj = 0; jj < HP; ++jj)
#pragma unroll
                    for (int u = 0; u < NT; ++u)
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bv[jj][u], acc[u], 0, 0, 0);
                rbase += a.row_len;
                wbase += 2 * HP * a.ocp;
            }

            // ---- epilogue: C/D map col = lane&31 (channel), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel).
            // The activation dispatch is hoisted out of the element loop (SiLU-only and ReLU-only stems are the cases).
            const bool simple = a.res == nullptr && a.act2 == SI_ACT_NONE;
            const size_t mrow = (size_t)(img * a.oh + oy) * a.ow;
            const int oxb = ox0 + wave * 32 + 4 * lh;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int o = oc0 + u * 32 + l31;
                if (o >= a.oc) continue;
                const float bvu = bvv[u];
                OutT* orow = static_cast<OutT*>(a.out) + mrow * a.out_ld + o;
                if (simple && a.act1 == SI_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ox = oxb + (e & 3) + 8 * (e >> 2);
                        const float v = acc[u][e] + bvu;
                        if (ox < a.ow) orow[(size_t)ox * a.out_ld] = si_store_cast<OutT>(v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)));
                    }
                } else if (simple && a.act1 == SI_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ox = oxb + (e & 3) + 8 * (e >> 2);
                        if (ox < a.ow) orow[(size_t)ox * a.out_ld] = si_store_cast<OutT>(fmaxf(acc[u][e] + bvu, 0.0f));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ox = oxb + (e & 3) + 8 * (e >> 2);
                        if (ox < a.ow) {
                            float v = acc[u][e] + bvu;
                            v = act_any(a.act1, v, a.act_param);
                            if (a.res) v += a.res[(mrow + ox) * a.res_ld + o];
                            v = act_any(a.act2, v, a.act_param);
                            orow[(size_t)ox * a.out_ld] = si_store_cast<OutT>(v);
                        }
                    }
                }
            }
        }

        if (next < a.items) {
            __syncthreads();  // every wave is done reading this item's rows
            commit(smem);
            __syncthreads();
        }
    }
}

template <int NW, int NT, int RB, int HP, int PF, typename OutT = float>
int launch_smallc(SmallCArgs a, hipStream_t s) {
    constexpr int TOW = 32 * NW;
    a.w_tiles = (a.ow + TOW - 1) / TOW;
    a.oc_tiles = (a.oc + 32 * NT - 1) / (32 * NT);
    a.row_blocks = (a.oh + RB - 1) / RB;
    // window shift (<= 3) + taps of the last pixel + 2 (the even-padded kernel row may read one element past), rounded to
    // whole 16-byte vectors
    a.row_len = (3 + ((TOW - 1) * a.sw + a.kw) * a.c + 2 + 3) / 4 * 4;
    a.n_in_rows = (RB - 1) * a.sh + a.kh;
    if (a.n_in_rows > PF / 4 || a.row_len > 4 * NW * 64) return SI_E_UNSUPPORTED;
    const long long items = (long long)a.n * a.row_blocks * a.w_tiles;
    if (items > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    a.items = (int)items;
    // one row buffer (padded so the weight image behind it is 16-byte aligned) + the weight image
    const size_t lds = ((size_t)a.n_in_rows * a.row_len + (size_t)a.kh * 2 * HP * a.ocp) * sizeof(float);
    if (lds > 160 * 1024) return SI_E_UNSUPPORTED;
    const bool vec = a.in_ld == a.c && (a.iw * a.c) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0;
    auto kern = vec ? conv_smallc_rows_kernel<NW, NT, RB, HP, PF, OutT, true> : conv_smallc_rows_kernel<NW, NT, RB, HP, PF, OutT, false>;
    if (hipError_t e = si_allow_dynamic_lds(kern, lds); e != hipSuccess) return (int)e;
    // persistent grid: exactly the workgroups that are resident at once (registers or LDS, whichever binds)
    const int per_cu = si_resident_blocks(kern, NW * 64, lds);
    int grid = 256 * per_cu / a.oc_tiles;
    if (grid < 1) grid = 1;
    if ((long long)grid > items) grid = (int)items;
    hipLaunchKernelGGL(kern, dim3(grid, a.oc_tiles), dim3(NW * 64), lds, s, a);
    return (int)hipGetLastError();
}

inline int smallc_hp(const SiConv2dDesc* d) { return (d->kw * d->ic + 1) / 2; }

}  // namespace

// Shape-only eligibility (it also decides the weight layout, so it must not depend on pointers or strides):
// 1..3 input channels, no groups / dilation, stride <= 2, and one of the instantiated kernel-row lengths.
bool si_conv_smallc_ok(const SiConv2dDesc* d) {
    if (d->groups != 1 || d->dh != 1 || d->dw != 1) return false;
    if (d->ic < 1 || d->ic > 3) return false;
    if (d->sh < 1 || d->sh > 2 || d->sw < 1 || d->sw > 2 || d->kh > 7 || d->kw > 7) return false;
    const int hp = smallc_hp(d);
    return hp == 5 || hp == 9 || hp == 11;  // 3x3x3 (MobileNet stem), 6x6x3 (YOLOv5 stem), 7x7x3 (ResNet stem)
}

size_t si_conv_smallc_weight_elems(const SiConv2dDesc* d) {
    if (stem_roll(d)) return si_conv_stemroll_weight_elems(d);
    const int ocp = (d->oc + 31) / 32 * 32;
    return (size_t)d->kh * 2 * smallc_hp(d) * ocp;
}

// OIHW -> [ky][j = kx*c + ch, padded to 2*HP][ocp], zero filled
void si_conv_smallc_pack(const SiConv2dDesc* d, const float* w_oihw, float* w_packed) {
    if (stem_roll(d)) return si_conv_stemroll_pack(d, w_oihw, w_packed);
    const int hp2 = 2 * smallc_hp(d);
    const int ocp = (d->oc + 31) / 32 * 32;
    const size_t total = si_conv_smallc_weight_elems(d);
    for (size_t i = 0; i < total; ++i) w_packed[i] = 0.0f;
    for (int o = 0; o < d->oc; ++o)
        for (int c = 0; c < d->ic; ++c)
            for (int ky = 0; ky < d->kh; ++ky)
                for (int kx = 0; kx < d->kw; ++kx)
                    w_packed[((size_t)ky * hp2 + kx * d->ic + c) * ocp + o] =
                        w_oihw[(((size_t)o * d->ic + c) * d->kh + ky) * d->kw + kx];
}

const char* si_conv_smallc_name(const SiConv2dDesc* d) {
    if (stem_roll(d)) return si_conv_stemroll_name(d);
    const int hp = smallc_hp(d);
    if (hp == 5) return d->oc > 32 ? "conv_smallc_rows_kernel<4, 2, 2, 5, 28>" : "conv_smallc_rows_kernel<4, 1, 2, 5, 28>";
    if (hp == 9)
        return d->oc > 32 ? "conv_smallc_rows_kernel<4, 2, 2, 9, 36>"
                          : ((d->ow % 160 == 0 || d->ow > 128) ? "conv_smallc_rows_kernel<5, 1, 1, 9, 28>" : "conv_smallc_rows_kernel<4, 1, 1, 9, 28>");
    return d->oc > 32 ? "conv_smallc_rows_kernel<4, 2, 2, 11, 36>" : "conv_smallc_rows_kernel<4, 1, 2, 11, 36>";
}

template <typename OutT>
static int smallc_launch_t(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                           const float* residual, void* out, hipStream_t s) {
    SmallCArgs a;
    a.in = in; a.w = w_packed; a.bias = d->has_bias ? bias : nullptr; a.res = d->has_residual ? residual : nullptr;
    a.out = out;
    a.n = d->n; a.ih = d->ih; a.iw = d->iw; a.c = d->ic; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.oc = d->oc; a.ocp = (d->oc + 31) / 32 * 32; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.pl = d->pl;
    a.row_len = a.n_in_rows = a.w_tiles = a.oc_tiles = a.row_blocks = a.items = 0;
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    const unsigned long long in_bytes = (unsigned long long)d->n * d->ih * d->iw * d->in_ld * 4ull;
    if (in_bytes >= 0xFFFFFF00ull) return SI_E_UNSUPPORTED;  // > 4 GB input image tensor
    a.in_bytes = (unsigned)in_bytes;
    // Variants measured on the YOLOv5s stem at batch 32 (MI355X): 5 waves x 1 output row per item (37 KB LDS, 4 resident
    // workgroups per CU) 0.40 ms; 2 rows per item 0.63 ms; 4 waves 0.45 ms.  Residency beats halo reuse here too.
    if (smallc_hp(d) == 5) {
        if (d->oc > 32) return launch_smallc<4, 2, 2, 5, 28, OutT>(a, s);
        return launch_smallc<4, 1, 2, 5, 28, OutT>(a, s);
    }
    if (smallc_hp(d) == 9) {
        if (d->oc > 32) return launch_smallc<4, 2, 2, 9, 36, OutT>(a, s);
        if (d->ow % 160 == 0 || d->ow > 128) return launch_smallc<5, 1, 1, 9, 28, OutT>(a, s);
        return launch_smallc<4, 1, 1, 9, 28, OutT>(a, s);
    }
    if (d->oc > 32) return launch_smallc<4, 2, 2, 11, 36, OutT>(a, s);
    return launch_smallc<4, 1, 2, 11, 36, OutT>(a, s);
}

int si_conv_smallc_launch(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                          const float* residual, float* out, hipStream_t s) {
    if (stem_roll(d)) return si_conv_stemroll_launch(d, in, w_packed, bias, residual, out, s);
    return smallc_launch_t<float>(d, in, w_packed, bias, residual, out, s);
}

