// conv_smallc.hip -- first-layer ("stem") convolution: very few input channels (RGB), large kernel, stride 2.
//
// YOLOv5s conv0 is 640x640x3 -> 320x320x32, 6x6 s2 p2 (reference shape: test/test_layer/test_conv_2d.cpp:279-293);
// ResNet18 conv1 is 224x224x3 -> 112x112x64, 7x7 s2 p3.  With 3 channels the im2col row of one tap is 12 bytes, so
// the implicit-GEMM kernel's 16-byte channel vectors do not apply and its scalar gather ran at 37 TFLOP/s.
//
// Here the NHWC layout is exploited the other way round: for one output pixel and one kernel row, the KW taps x C
// channels are KW*C CONTIGUOUS floats of the input row, and neighbouring output pixels overlap in all but
// stride*C of them.  So a workgroup
//   1. stages the KH (+ stride per extra output row) input rows that its output pixels need into LDS with
//      perfectly coalesced dword loads (each input byte is fetched ~1.5x instead of KH*KW/stride^2 = 9x),
//   2. stages the layer's weights transposed to [K][OC] (B operand: lanes = output channels, conflict free),
//   3. runs 32x32x2 MFMAs whose A operand is read straight from the staged rows:
//        A[m][k] = row[kh][(m*sw)*C + (kw*C + c)],   k = (kh, kw, c)
//      -- no im2col tile is ever formed, in HBM or in LDS.
// One wave owns 32 consecutive output pixels of one output row and all (<= 64) output channels.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "si_hip.h"
#include "si_hip_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct SmallCArgs {
    const float* in;
    const float* w;     // packed [oc][kh*kw][cpad] (the layout si_hip_conv2d_pack_weight_host produces)
    const float* bias;
    const float* res;
    float* out;
    int n, ih, iw, c, in_ld;
    int oh, ow, oc, out_ld, res_ld;
    int kh, kw, sh, sw, pt, pl;
    int cpad;           // channel padding of the packed weights
    int K;              // kh*kw*c
    int row_len;        // floats staged per input row
    int w_tiles;        // ceil(ow / (32*NW))
    unsigned magic_per_oc;  // ceil(2^32 / (kh*kw*4))
    unsigned magic_kw;      // ceil(2^32 / kw)
    int act1, act2;
    float act_param;
    int ablate;  // timing experiments (SI_CONV_ABLATE): 1 no input staging, 2 no weight staging, 4 no MFMA loop, 8 no stores
};

__device__ __forceinline__ float act_any(int act, float v, float p) {
    switch (act) {
        case SI_ACT_RELU: return fmaxf(v, 0.0f);
        case SI_ACT_SILU: return v / (1.0f + __expf(-v));
        case SI_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        case SI_ACT_HARDSIGMOID: return fminf(fmaxf(v * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f);
        case SI_ACT_HARDSWISH: return v * fminf(fmaxf(v * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f);
        case SI_ACT_LEAKYRELU: return v > 0.0f ? v : v * p;
        default: return v;
    }
}

// NW waves per workgroup (32*NW output pixels along W), NT 32-wide output-channel tiles per wave,
// RB output rows per workgroup (amortises the weight staging and the vertical halo).
// HP = (kw*c rounded up to even)/2 = MFMA steps per kernel row, a compile-time constant so the step loop unrolls
// into HP independent LDS reads followed by HP back-to-back MFMAs (9 for 6x6x3, 11 for 7x7x3).
template <int NW, int NT, int RB, int HP>
__global__ __launch_bounds__(NW * 64) void conv_smallc_rows_kernel(const SmallCArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TOW = 32 * NW;
    constexpr int OCW = 32 * NT;        // output channels handled by this workgroup
    constexpr int WLD = OCW + 1;        // +1: conflict-free transposing store
    const int n_in_rows = (RB - 1) * a.sh + a.kh;
    float* rows = smem;                                   // [n_in_rows][row_len]
    float* wl = smem + n_in_rows * a.row_len;             // [kh][2*HP][WLD]; rows j >= kw*c are zero

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int tile_w = blockIdx.x % a.w_tiles;
    const int oc0 = (blockIdx.x / a.w_tiles) * OCW;
    const int oy0 = blockIdx.y * RB;
    const int img = blockIdx.z;
    const int ox0 = tile_w * TOW;
    const int ix0 = ox0 * a.sw - a.pl;   // first staged input pixel (may be negative)
    const int iy0 = oy0 * a.sh - a.pt;

    // ---- stage input rows (zero outside the image).  The input is dense (in_ld == c), so a staged row is one
    // contiguous span of the image row: no pixel/channel decomposition, consecutive lanes -> consecutive dwords.
    {
        const int lo = ix0 < 0 ? -ix0 * a.c : 0;                 // first valid element of the span
        const int hi = min(a.row_len, (a.iw - ix0) * a.c);       // one past the last valid element
        for (int r = 0; r < n_in_rows && !(a.ablate & 1); ++r) {
            const int y = iy0 + r;
            const bool yok = (unsigned)y < (unsigned)a.ih;
            const float* src = a.in + ((size_t)(img * a.ih + (yok ? y : 0)) * a.iw) * a.c + (ptrdiff_t)ix0 * a.c;
            float* dst = rows + r * a.row_len;
            for (int e = tid; e < a.row_len; e += NW * 64) dst[e] = (yok && e >= lo && e < hi) ? src[e] : 0.0f;
        }
    }
    // ---- stage weights transposed: wl[ky*2HP + j][o] = w[oc0+o][ky][kx][ch], j = kx*c + ch.  cpad == 4 for c <= 4.
    {
        const int RL = a.kw * a.c;
        for (int i = tid; i < a.kh * 2 * HP * WLD; i += NW * 64) wl[i] = 0.0f;
        __syncthreads();
        const int per_oc = a.kh * a.kw * 4;
        const int total = OCW * per_oc;
        for (int i = tid; i < total && !(a.ablate & 2); i += NW * 64) {
            const int o = (int)__umulhi((unsigned)i, a.magic_per_oc);  // i / per_oc, exact for i*per_oc < 2^32
            const int rem = i - o * per_oc;
            const int tap = rem >> 2, ch = rem & 3;
            const int ky = (int)__umulhi((unsigned)tap, a.magic_kw);
            const int kx = tap - ky * a.kw;
            if (ch < a.c && oc0 + o < a.oc) wl[(ky * 2 * HP + kx * a.c + ch) * WLD + o] = a.w[(size_t)(oc0 + o) * per_oc + rem];
        }
        (void)RL;
    }
    __syncthreads();

    const int px_off = (wave * 32 + l31) * a.sw * a.c + lh;   // lane half h reads element 2*jj + h of the kernel row

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
        const float* wbase = wl + lh * WLD + l31;
        for (int ky = 0; ky < a.kh && !(a.ablate & 4); ++ky) {
            float av[HP], bv[HP][NT];
#pragma unroll
            for (int jj = 0; jj < HP; ++jj) {
                av[jj] = rbase[2 * jj];  // elements past kw*c belong to the next pixels: finite, and their weights are 0
#pragma unroll
                for (int u = 0; u < NT; ++u) bv[jj][u] = wbase[2 * jj * WLD + u * 32];
            }
#pragma unroll
            for (int jj = 0; jj < HP; ++jj)
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bv[jj][u], acc[u], 0, 0, 0);
            rbase += a.row_len;
            wbase += 2 * HP * WLD;
        }

        // ---- epilogue: C/D map col = lane&31 (channel), row = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int o = oc0 + u * 32 + l31;
            if (o >= a.oc) continue;
            const float bv = a.bias ? a.bias[o] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ox = ox0 + wave * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
                if (ox < a.ow && !(a.ablate & 8)) {
                    const size_t m = (size_t)(img * a.oh + oy) * a.ow + ox;
                    float v = acc[u][e] + bv;
                    v = act_any(a.act1, v, a.act_param);
                    if (a.res) v += a.res[m * a.res_ld + o];
                    v = act_any(a.act2, v, a.act_param);
                    a.out[m * a.out_ld + o] = v;
                }
            }
        }
    }
}

template <int NW, int NT, int RB, int HP>
int launch_smallc(SmallCArgs a, hipStream_t s) {
    constexpr int TOW = 32 * NW;
    a.w_tiles = (a.ow + TOW - 1) / TOW;
    a.row_len = ((TOW - 1) * a.sw + a.kw) * a.c + 2;  // +2: the even-padded kernel row may read one element past
    const int n_in_rows = (RB - 1) * a.sh + a.kh;
    const size_t lds = ((size_t)n_in_rows * a.row_len + 2 + (size_t)a.kh * 2 * HP * (32 * NT + 1)) * sizeof(float);
    if (lds > 160 * 1024) return SI_E_UNSUPPORTED;
    const int oc_tiles = (a.oc + 32 * NT - 1) / (32 * NT);
    dim3 grid(a.w_tiles * oc_tiles, (a.oh + RB - 1) / RB, a.n);
    auto kern = conv_smallc_rows_kernel<NW, NT, RB, HP>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

// eligibility: dense small-channel input, no groups / dilation, kernel rows fit comfortably in LDS
bool si_conv_smallc_ok(const SiConv2dDesc* d) {
    if (d->groups != 1 || d->dh != 1 || d->dw != 1) return false;
    if (d->ic > 4 || d->ic % 4 == 0) return false;  // 1..3 channels (4 takes the vector path)
    if (d->in_ld != d->ic) return false;            // staged rows must be contiguous spans
    if (d->kh * d->kw * d->ic > 512) return false;
    if (d->n > 65535 || d->oh > 65535) return false;
    const int hp = (d->kw * d->ic + 1) / 2;
    return hp == 9 || hp == 11;  // instantiated row lengths: 6x6x3 (YOLOv5 stem), 7x7x3 (ResNet stem)
}

const char* si_conv_smallc_name(const SiConv2dDesc* d) {
    const int hp = (d->kw * d->ic + 1) / 2;
    if (hp == 9) return d->oc > 32 ? "conv_smallc_rows_kernel<4, 2, 2, 9>" : (d->ow % 160 == 0 ? "conv_smallc_rows_kernel<5, 1, 2, 9>" : "conv_smallc_rows_kernel<4, 1, 2, 9>");
    return d->oc > 32 ? "conv_smallc_rows_kernel<4, 2, 2, 11>" : "conv_smallc_rows_kernel<4, 1, 2, 11>";
}

int si_conv_smallc_launch(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                          const float* residual, float* out, hipStream_t s) {
    SmallCArgs a;
    a.in = in; a.w = w_packed; a.bias = d->has_bias ? bias : nullptr; a.res = d->has_residual ? residual : nullptr;
    a.out = out;
    a.n = d->n; a.ih = d->ih; a.iw = d->iw; a.c = d->ic; a.in_ld = d->in_ld;
    a.oh = d->oh; a.ow = d->ow; a.oc = d->oc; a.out_ld = d->out_ld; a.res_ld = d->res_ld;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->sh; a.sw = d->sw; a.pt = d->pt; a.pl = d->pl;
    a.cpad = (d->ic + 3) & ~3;
    a.K = d->kh * d->kw * d->ic;
    a.row_len = 0; a.w_tiles = 0;
    a.magic_per_oc = (unsigned)((0x100000000ull + (unsigned long long)(d->kh * d->kw * 4) - 1) / (unsigned long long)(d->kh * d->kw * 4));
    a.magic_kw = (unsigned)((0x100000000ull + (unsigned long long)d->kw - 1) / (unsigned long long)d->kw);
    a.act1 = d->act1; a.act2 = d->act2; a.act_param = d->act_param;
    static const int ablate = [] { const char* e = getenv("SI_CONV_ABLATE"); return e ? atoi(e) : 0; }();
    a.ablate = ablate;
    const int hp = (d->kw * d->ic + 1) / 2;
    if (hp == 9) {
        if (d->oc > 32) return launch_smallc<4, 2, 2, 9>(a, s);
        if (d->ow % 160 == 0) return launch_smallc<5, 1, 2, 9>(a, s);
        return launch_smallc<4, 1, 2, 9>(a, s);
    }
    if (d->oc > 32) return launch_smallc<4, 2, 2, 11>(a, s);
    return launch_smallc<4, 1, 2, 11>(a, s);
}
