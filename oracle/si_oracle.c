/*
 * si_oracle.c -- CPU restatement of SimpleInfer's operator algorithms.
 *
 * TEST INFRASTRUCTURE ONLY (see si_oracle.h).  Plain C11 + OpenMP, fp32.
 * Each function cites the reference file:line it follows.  Nothing here is
 * copied from the reference: loops are restated from the documented
 * arithmetic (evaluation order kept where it decides rounding).
 */
#include "si_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ========================================================================
 * GEMM pack-4.  Reference: src/layer/simd/gemm.cpp:295-385 (driver),
 * :72-157 (4x12 micro-kernel), :405-424 (scalar Ref).
 * Every C element is an independent chain c = fma(a[m][k], b[k][n], c) for
 * k = 0..K-1 starting at 0, then a plain store (C is overwritten).  The
 * register blocking of the reference only changes which elements are
 * computed together, never the per-element chain, so any blocking gives the
 * same bits.
 * ======================================================================== */
void orc_gemm_pack4_f32(size_t M, size_t N, size_t K, const float* A,
                        size_t lda, const float* Bp, float* C, size_t ldc) {
    const size_t n_packs = (N + 3) / 4;
    for (size_t jp = 0; jp < n_packs; ++jp) {
        const float* B = Bp + jp * K * 4; /* ldb = K*4, gemm.cpp:303 */
        const size_t ncols = (jp * 4 + 4 <= N) ? 4 : (N - jp * 4);
        size_t i = 0;
        for (; i + 8 <= M; i += 8) {
            float c[8][4];
            memset(c, 0, sizeof(c));
            for (size_t k = 0; k < K; ++k) {
                const float* b = B + k * 4;
                for (int r = 0; r < 8; ++r) {
                    const float a = A[(i + r) * lda + k];
                    for (int l = 0; l < 4; ++l) c[r][l] = fmaf(a, b[l], c[r][l]);
                }
            }
            for (int r = 0; r < 8; ++r)
                for (size_t l = 0; l < ncols; ++l)
                    C[(i + r) * ldc + jp * 4 + l] = c[r][l];
        }
        for (; i < M; ++i) {
            float c[4] = {0.f, 0.f, 0.f, 0.f};
            for (size_t k = 0; k < K; ++k) {
                const float a = A[i * lda + k];
                const float* b = B + k * 4;
                for (int l = 0; l < 4; ++l) c[l] = fmaf(a, b[l], c[l]);
            }
            for (size_t l = 0; l < ncols; ++l) C[i * ldc + jp * 4 + l] = c[l];
        }
    }
}

/* gemm.cpp:405-424 -- separate multiply and add, as written there. */
void orc_gemm_pack4_f32_ref(size_t M, size_t N, size_t K, const float* A,
                            size_t lda, const float* Bp, float* C,
                            size_t ldc) {
    for (size_t m = 0; m < M; ++m) {
        for (size_t n = 0; n < N; ++n) {
            const size_t n4 = n / 4, nres = n % 4;
            volatile float acc = 0.0f; /* volatile: forbid fma contraction */
            for (size_t k = 0; k < K; ++k) {
                volatile float prod = A[m * lda + k] * Bp[n4 * K * 4 + k * 4 + nres];
                acc = acc + prod;
            }
            C[m * ldc + n] = acc;
        }
    }
}

/* ========================================================================
 * Winograd F(2,3).  Reference: src/layer/simd/winograd_helper.cpp
 * ======================================================================== */

/* U = G g G^T with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], evaluated in
 * the order of winograd_helper.cpp:86-129, then packed
 * [16][oc_up4/4][ic][4] (:132-142).  g is indexed t[kh*3+kw]. */
void orc_wino23_transform_kernel_pack4(const float* hwio, size_t ic, size_t oc,
                                       float* dst) {
    const size_t oc4 = (oc + 3) / 4 * 4;
    const size_t plane = ic * oc4;
    for (size_t i = 0; i < ic; ++i) {
        for (size_t j = 0; j < oc4; ++j) {
            float t[9];
            for (int k = 0; k < 9; ++k)
                t[k] = (j < oc) ? hwio[(size_t)k * ic * oc + i * oc + j] : 0.0f;
            float u[16];
            const float r2 = 0.5f, r4 = 0.25f;
            {
                const float a02 = t[0] + t[2];
                u[0] = t[0];
                u[1] = (a02 + t[1]) * r2;
                u[2] = (a02 - t[1]) * r2;
                u[3] = t[2];
            }
            {
                const float a063 = (t[0] + t[6]) + t[3];
                const float a285 = (t[2] + t[8]) + t[5];
                const float a174 = (t[1] + t[7]) + t[4];
                u[4] = a063 * r2;
                u[5] = ((a063 + a285) + a174) * r4;
                u[6] = ((a063 + a285) - a174) * r4;
                u[7] = a285 * r2;
            }
            {
                const float s063 = (t[0] + t[6]) - t[3];
                const float s285 = (t[2] + t[8]) - t[5];
                const float s174 = (t[1] + t[7]) - t[4];
                u[8] = s063 * r2;
                u[9] = ((s063 + s285) + s174) * r4;
                u[10] = ((s063 + s285) - s174) * r4;
                u[11] = s285 * r2;
            }
            {
                const float a68 = t[6] + t[8];
                u[12] = t[6];
                u[13] = (a68 + t[7]) * r2;
                u[14] = (a68 - t[7]) * r2;
                u[15] = t[8];
            }
            /* [16][oc4/4][ic][4] */
            for (int k = 0; k < 16; ++k)
                dst[(size_t)k * plane + (j / 4) * ic * 4 + i * 4 + (j % 4)] = u[k];
        }
    }
}

/* V = B^T d B written exactly as winograd_helper.cpp:188-239 orders it. */
static inline void wino23_bt_d_b(const float d[16], float v[16]) {
    v[0] = (d[0] - d[8]) - (d[2] - d[10]);
    v[1] = (d[1] - d[9]) + (d[2] - d[10]);
    v[2] = (d[2] - d[10]) - (d[1] - d[9]);
    v[3] = (d[1] - d[9]) - (d[3] - d[11]);
    v[4] = (d[4] + d[8]) - (d[6] + d[10]);
    v[5] = (d[5] + d[9]) + (d[6] + d[10]);
    v[6] = (d[6] + d[10]) - (d[5] + d[9]);
    v[7] = (d[5] + d[9]) - (d[7] + d[11]);
    v[8] = (d[8] - d[4]) - (d[10] - d[6]);
    v[9] = (d[9] - d[5]) + (d[10] - d[6]);
    v[10] = (d[10] - d[6]) - (d[9] - d[5]);
    v[11] = (d[9] - d[5]) - (d[11] - d[7]);
    v[12] = (d[4] - d[12]) - (d[6] - d[14]);
    v[13] = (d[5] - d[13]) + (d[6] - d[14]);
    v[14] = (d[6] - d[14]) - (d[5] - d[13]);
    v[15] = (d[5] - d[13]) - (d[7] - d[15]);
}

/* winograd_helper.cpp:413-580.  Tiles step by 2 over the virtually
 * zero-padded image, row-major over (tile_h, tile_w); the reference's
 * nose/body/tail split (:462-578) only decides which of the 16 taps are
 * loaded vs zero-filled (:169-186), which is what the bounds test below does.
 * Output layout [16][tiles][ic], plane stride dst_stride (conv_2d.cpp:419).  */
void orc_wino23_transform_input(const float* src, size_t ih, size_t iw,
                                size_t ic, int pad, float* dst,
                                size_t dst_stride, int q1_bug) {
    const long p = pad ? 1 : 0;
    const size_t oh = pad ? ih : ih - 2;
    const size_t ow = pad ? iw : iw - 2;
    const size_t th_n = (oh + 1) / 2, tw_n = (ow + 1) / 2;

    /* Q1: the tail tile-row is guarded by `row < ow` (:540) instead of
     * `row < oh`.  tail_row is the value of `row` after the body loop. */
    size_t drop_last_row = 0;
    if (q1_bug) {
        size_t oh2 = oh / 2 * 2;
        const size_t start = pad ? 2 : 0;
        if (pad && oh == oh2) oh2 -= 2;
        size_t row = start;
        while (row < oh2) row += 2;
        const int tail_exists = row < oh;
        if (tail_exists && !(row < ow)) drop_last_row = 1;
    }

    for (size_t th = 0; th < th_n; ++th) {
        for (size_t tw = 0; tw < tw_n; ++tw) {
            float* out = dst + (th * tw_n + tw) * ic;
            if (drop_last_row && th == th_n - 1) {
                for (int k = 0; k < 16; ++k)
                    memset(out + (size_t)k * dst_stride, 0, ic * sizeof(float));
                continue;
            }
            for (size_t c = 0; c < ic; ++c) {
                float d[16], v[16];
                for (int r = 0; r < 4; ++r) {
                    const long y = (long)(2 * th) - p + r;
                    for (int q = 0; q < 4; ++q) {
                        const long x = (long)(2 * tw) - p + q;
                        d[r * 4 + q] =
                            (y >= 0 && y < (long)ih && x >= 0 && x < (long)iw)
                                ? src[((size_t)y * iw + (size_t)x) * ic + c]
                                : 0.0f;
                    }
                }
                wino23_bt_d_b(d, v);
                for (int k = 0; k < 16; ++k) out[(size_t)k * dst_stride + c] = v[k];
            }
        }
    }
}

/* Y = A^T m A: row pass t0 = s0+s1+s2, t1 = s1-s2-s3 then column pass
 * (winograd_helper.cpp:582-615, scalar form :616-652); stores clipped to
 * oh/ow for odd sizes (:664-675, :839-872). */
void orc_wino23_transform_output(const float* src, size_t src_stride,
                                 float* dst, size_t oh, size_t ow, size_t oc) {
    const size_t th_n = (oh + 1) / 2, tw_n = (ow + 1) / 2;
    for (size_t th = 0; th < th_n; ++th) {
        for (size_t tw = 0; tw < tw_n; ++tw) {
            const float* in = src + (th * tw_n + tw) * oc;
            for (size_t c = 0; c < oc; ++c) {
                float s[16], t[8], y[4];
                for (int k = 0; k < 16; ++k) s[k] = in[(size_t)k * src_stride + c];
                t[0] = (s[0] + s[1]) + s[2];
                t[1] = (s[1] - s[2]) - s[3];
                t[2] = (s[4] + s[5]) + s[6];
                t[3] = (s[5] - s[6]) - s[7];
                t[4] = (s[8] + s[9]) + s[10];
                t[5] = (s[9] - s[10]) - s[11];
                t[6] = (s[12] + s[13]) + s[14];
                t[7] = (s[13] - s[14]) - s[15];
                y[0] = (t[0] + t[2]) + t[4];
                y[1] = (t[1] + t[3]) + t[5];
                y[2] = (t[2] - t[4]) - t[6];
                y[3] = (t[3] - t[5]) - t[7];
                for (int r = 0; r < 2; ++r)
                    for (int q = 0; q < 2; ++q) {
                        const size_t yy = 2 * th + r, xx = 2 * tw + q;
                        if (yy < oh && xx < ow)
                            dst[(yy * ow + xx) * oc + c] = y[r * 2 + q];
                    }
            }
        }
    }
}

/* src/layer/simd/binary.cpp:38-53 */
void orc_add_bias_nhwc(const float* bias, size_t spatial, size_t oc,
                       float* dst) {
    for (size_t s = 0; s < spatial; ++s)
        for (size_t c = 0; c < oc; ++c) dst[s * oc + c] += bias[c];
}

/* ========================================================================
 * Conv2d.  Reference: src/layer/conv_2d.cpp
 * ======================================================================== */
void orc_conv2d_out_shape(const OrcConv2d* p, int* oh, int* ow) {
    const int ekh = (p->kh - 1) * p->dh + 1, ekw = (p->kw - 1) * p->dw + 1;
    *oh = (p->ih + p->pt + p->pb - ekh) / p->sh + 1;
    *ow = (p->iw + p->pl + p->pr - ekw) / p->sw + 1;
}

/* OIHW -> HWIO (conv_2d.cpp:126-150); weight [oc][icg][kh][kw] */
static float* oihw_to_hwio(const OrcConv2d* p, const float* w) {
    const int icg = p->ic / p->groups;
    const size_t total = (size_t)p->oc * icg * p->kh * p->kw;
    float* t = (float*)malloc(total * sizeof(float));
    for (int o = 0; o < p->oc; ++o)
        for (int i = 0; i < icg; ++i)
            for (int h = 0; h < p->kh; ++h)
                for (int x = 0; x < p->kw; ++x)
                    t[(((size_t)h * p->kw + x) * icg + i) * p->oc + o] =
                        w[(((size_t)o * icg + i) * p->kh + h) * p->kw + x];
    return t;
}

/* ForwardIm2Col (conv_2d.cpp:207-283) and ForwardIm2ColWithGroup (:285-380):
 * out[m, o] = sum_k patch[m, k] * W[k, o] (+ bias afterwards), K ordered
 * (kh, kw, ic) -- the row-major NHWC patch layout -- zero padding (:260).
 * PyTorch padding semantics (pad_t on H): SURVEY Q2.  The contraction order
 * inside Eigen is not specified; restated as a k-ordered fp32 fma chain. */
int orc_conv2d_im2col(const OrcConv2d* p, const float* in, const float* w_oihw,
                      const float* bias, float* out) {
    int oh, ow;
    orc_conv2d_out_shape(p, &oh, &ow);
    if (oh <= 0 || ow <= 0 || p->groups <= 0) return -1;
    const int G = p->groups, icg = p->ic / G, ocg = p->oc / G;
    float* hwio = oihw_to_hwio(p, w_oihw); /* [kh][kw][icg][oc] */
    const size_t M = (size_t)p->n * oh * ow;
    enum { PB = 8 };
#pragma omp parallel
    {
        float* acc = (float*)malloc((size_t)PB * ocg * sizeof(float));
        const float* ap[PB];
#pragma omp for schedule(static)
        for (long mb = 0; mb < (long)((M + PB - 1) / PB); ++mb) {
            const size_t m0 = (size_t)mb * PB;
            const int np = (int)((m0 + PB <= M) ? PB : (M - m0));
            for (int g = 0; g < G; ++g) {
                memset(acc, 0, (size_t)PB * ocg * sizeof(float));
                for (int kh = 0; kh < p->kh; ++kh)
                    for (int kw = 0; kw < p->kw; ++kw) {
                        int any = 0;
                        for (int q = 0; q < np; ++q) {
                            const size_t m = m0 + q;
                            const int b = (int)(m / ((size_t)oh * ow));
                            const int r = (int)(m % ((size_t)oh * ow));
                            const int y = (r / ow) * p->sh - p->pt + kh * p->dh;
                            const int x = (r % ow) * p->sw - p->pl + kw * p->dw;
                            if (y >= 0 && y < p->ih && x >= 0 && x < p->iw) {
                                ap[q] = in + (((size_t)b * p->ih + y) * p->iw + x) * p->ic + (size_t)g * icg;
                                any = 1;
                            } else {
                                ap[q] = NULL;
                            }
                        }
                        if (!any) continue; /* fma(0, w, acc) == acc */
                        const float* wk = hwio + ((size_t)kh * p->kw + kw) * icg * p->oc + (size_t)g * ocg;
                        for (int c = 0; c < icg; ++c) {
                            const float* wr = wk + (size_t)c * p->oc;
                            for (int q = 0; q < np; ++q) {
                                if (!ap[q]) continue;
                                const float a = ap[q][c];
                                float* ac = acc + (size_t)q * ocg;
                                for (int o = 0; o < ocg; ++o) ac[o] = fmaf(a, wr[o], ac[o]);
                            }
                        }
                    }
                for (int q = 0; q < np; ++q) {
                    float* dst = out + (m0 + q) * p->oc + (size_t)g * ocg;
                    const float* ac = acc + (size_t)q * ocg;
                    if (p->use_bias && bias)
                        for (int o = 0; o < ocg; ++o) dst[o] = ac[o] + bias[g * ocg + o];
                    else
                        for (int o = 0; o < ocg; ++o) dst[o] = ac[o];
                }
            }
        }
        free(acc);
    }
    free(hwio);
    return 0;
}

static int wino_eligible(const OrcConv2d* p) {
    /* conv_2d.cpp:183-187 */
    return p->kh == 3 && p->kw == 3 && p->sh == 1 && p->sw == 1 && p->dh == 1 &&
           p->dw == 1 && p->groups == 1 && p->pt == p->pb && p->pt == p->pl &&
           p->pt == p->pr && (p->pt == 0 || p->pt == 1);
}

/* ForwardWinograd23 (conv_2d.cpp:382-487): per image, input transform ->
 * 16 GEMMs (M=tiles, N=oc, K=ic) -> output transform -> add bias. */
int orc_conv2d_winograd23(const OrcConv2d* p, const float* in,
                          const float* w_oihw, const float* bias, float* out,
                          int q1_bug) {
    if (!wino_eligible(p)) return -1;
    int oh, ow;
    orc_conv2d_out_shape(p, &oh, &ow);
    const size_t ic = p->ic, oc = p->oc;
    const size_t oc4 = (oc + 3) / 4 * 4;
    const size_t tiles = (size_t)((oh + 1) / 2) * ((ow + 1) / 2);
    float* hwio = oihw_to_hwio(p, w_oihw);
    float* U = (float*)calloc(16 * ic * oc4, sizeof(float));
    orc_wino23_transform_kernel_pack4(hwio, ic, oc, U);
    free(hwio);
    const size_t in_plane = tiles * ic, out_plane = tiles * oc;
    float* V = (float*)calloc(16 * in_plane, sizeof(float));
    float* Mb = (float*)calloc(16 * tiles * oc4, sizeof(float));
    for (int b = 0; b < p->n; ++b) {
        const float* src = in + (size_t)b * p->ih * p->iw * ic;
        float* dst = out + (size_t)b * oh * ow * oc;
        orc_wino23_transform_input(src, p->ih, p->iw, ic, p->pt == 1, V, in_plane, q1_bug);
        /* conv_2d.cpp:451-467: 16 independent GEMMs */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < 16; ++i)
            orc_gemm_pack4_f32(tiles, oc, ic, V + (size_t)i * in_plane, ic,
                               U + (size_t)i * ic * oc4, Mb + (size_t)i * out_plane, oc);
        orc_wino23_transform_output(Mb, out_plane, dst, oh, ow, oc);
        if (p->use_bias && bias) orc_add_bias_nhwc(bias, (size_t)oh * ow, oc, dst);
    }
    free(U);
    free(V);
    free(Mb);
    return 0;
}

/* Conv2d::Forward (conv_2d.cpp:108-118) */
int orc_conv2d_forward(const OrcConv2d* p, const float* in,
                       const float* w_oihw, const float* bias, float* out) {
    if (wino_eligible(p)) return orc_conv2d_winograd23(p, in, w_oihw, bias, out, 0);
    return orc_conv2d_im2col(p, in, w_oihw, bias, out);
}

/* test/test_layer/test_conv_2d.cpp:100-131 (loop order c, h, w; skip
 * out-of-range taps; bias added last). */
int orc_conv2d_naive(const OrcConv2d* p, const float* in, const float* w_oihw,
                     const float* bias, float* out, int acc64) {
    int oh, ow;
    orc_conv2d_out_shape(p, &oh, &ow);
    const int G = p->groups, icg = p->ic / G, ocg = p->oc / G;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < p->n; ++b)
        for (int j = 0; j < oh; ++j)
            for (int k = 0; k < ow; ++k)
                for (int l = 0; l < p->oc; ++l) {
                    const int g = l / ocg;
                    double sd = 0.0;
                    float sf = 0.0f;
                    for (int c = 0; c < icg; ++c)
                        for (int h = 0; h < p->kh; ++h)
                            for (int x = 0; x < p->kw; ++x) {
                                const int y = j * p->sh - p->pt + h * p->dh;
                                const int xx = k * p->sw - p->pl + x * p->dw;
                                if (y < 0 || y >= p->ih || xx < 0 || xx >= p->iw) continue;
                                const float a = in[(((size_t)b * p->ih + y) * p->iw + xx) * p->ic + g * icg + c];
                                const float wv = w_oihw[(((size_t)l * icg + c) * p->kh + h) * p->kw + x];
                                if (acc64)
                                    sd += (double)a * (double)wv;
                                else
                                    sf += a * wv;
                            }
                    if (p->use_bias && bias) {
                        sd += bias[l];
                        sf += bias[l];
                    }
                    out[(((size_t)b * oh + j) * ow + k) * p->oc + l] = acc64 ? (float)sd : sf;
                }
    return 0;
}

/* NOT a reference algorithm: the accumulation order of the DEVICE's fp32
 * implicit-GEMM convolution (simpleinfer_amd/csrc/hip/conv_igemm.hip),
 * restated as a scalar fmaf chain.  gfx950's fp32 MFMAs round exactly like
 * a sequential fma chain over k (tools/mfma_chain_test.hip), so this loop
 * predicts the kernel's output BIT FOR BIT, whatever workgroup tile or MFMA
 * shape a launch uses -- the instrument behind the engine's claim that an
 * image's bits do not depend on the batch it rides in.  Order of k:
 *   - channel-block major (c/32, kh, kw, c%32) when ic/groups is a multiple
 *     of 32 and the kernel is larger than 1x1, else (kh, kw, c) with the
 *     channel axis padded to x4 (x32 for ungrouped 1x1 convs whose channel
 *     count is a multiple of 4, >= 8);
 *   - inside every 16-wide block of that sequence: j, 4+j, 8+j, 12+j for
 *     j = 0..3.
 * Out-of-image taps and pad channels are fma(0, w, acc) = acc on the device
 * and skipped here.  Bias is one separate add after the chain.
 * Ungrouped / plain grouped convolutions only (not the merged-group, stem,
 * depthwise or Winograd kernels, which have their own orders). */
int orc_conv2d_chain(const OrcConv2d* p, const float* in, const float* w_oihw,
                     const float* bias, float* out) {
    int oh, ow;
    orc_conv2d_out_shape(p, &oh, &ow);
    const int G = p->groups, icg = p->ic / G, ocg = p->oc / G;
    const int ntaps = p->kh * p->kw;
    int icp = (icg + 3) & ~3;
    if (ntaps == 1 && G == 1 && icg % 4 == 0 && icg % 32 != 0 && icg >= 8) icp = (icg + 31) / 32 * 32;
    const int cbm = (icg % 32 == 0) && ntaps > 1;
    const int Kp = ntaps * icp, Kt = (Kp + 31) / 32 * 32;
    /* device k -> (tap, channel) or -1 */
    int* ktap = (int*)malloc(sizeof(int) * (size_t)Kt);
    int* kch = (int*)malloc(sizeof(int) * (size_t)Kt);
    if (!ktap || !kch) { free(ktap); free(kch); return -1; }
    int n = 0;
    for (int k0 = 0; k0 < Kt; k0 += 16)
        for (int j = 0; j < 4; ++j)
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + 4 * q + j;
                int tap = -1, c = -1;
                if (k < Kp) {
                    if (cbm) {
                        const int blk = k >> 5, cb = blk / ntaps;
                        tap = blk - cb * ntaps;
                        c = cb * 32 + (k & 31);
                    } else {
                        tap = k / icp;
                        c = k - tap * icp;
                    }
                    if (c >= icg) tap = -1;
                }
                ktap[n] = tap;
                kch[n] = c;
                ++n;
            }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < p->n; ++b)
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x)
                for (int o = 0; o < p->oc; ++o) {
                    const int g = o / ocg;
                    float acc = 0.0f;
                    for (int i = 0; i < Kt; ++i) {
                        if (ktap[i] < 0) continue;
                        const int ky = ktap[i] / p->kw, kx = ktap[i] - ky * p->kw;
                        const int iy = y * p->sh - p->pt + ky * p->dh, ix = x * p->sw - p->pl + kx * p->dw;
                        if (iy < 0 || iy >= p->ih || ix < 0 || ix >= p->iw) continue;
                        const float a = in[(((size_t)b * p->ih + iy) * p->iw + ix) * p->ic + g * icg + kch[i]];
                        const float wv = w_oihw[(((size_t)o * icg + kch[i]) * p->kh + ky) * p->kw + kx];
                        acc = fmaf(a, wv, acc);
                    }
                    if (p->use_bias && bias) acc = acc + bias[o];
                    out[(((size_t)b * oh + y) * ow + x) * p->oc + o] = acc;
                }
    free(ktap);
    free(kch);
    return 0;
}

/* ========================================================================
 * Other layers
 * ======================================================================== */

/* src/layer/linear.cpp:101-113 */
void orc_linear(const float* x, int n, int in_f, const float* w,
                const float* b, int out_f, float* y) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < n; ++i)
        for (int o = 0; o < out_f; ++o) {
            float acc = 0.0f;
            for (int k = 0; k < in_f; ++k)
                acc = fmaf(x[(size_t)i * in_f + k], w[(size_t)o * in_f + k], acc);
            y[(size_t)i * out_f + o] = b ? acc + b[o] : acc;
        }
}

/* src/layer/max_pool_2d.cpp:102-118: patches padded with lowest(), max over
 * the window.  ceil_mode / return_indices are parsed but ignored (:17-24). */
void orc_maxpool2d(const float* in, int n, int ih, int iw, int c, int kh,
                   int kw, int sh, int sw, int dh, int dw, int pt, int pl,
                   float* out, int oh, int ow) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x) {
                float* o = out + (((size_t)b * oh + y) * ow + x) * c;
                for (int ch = 0; ch < c; ++ch) o[ch] = -FLT_MAX;
                for (int i = 0; i < kh; ++i)
                    for (int j = 0; j < kw; ++j) {
                        const int yy = y * sh - pt + i * dh, xx = x * sw - pl + j * dw;
                        if (yy < 0 || yy >= ih || xx < 0 || xx >= iw) continue;
                        const float* s = in + (((size_t)b * ih + yy) * iw + xx) * c;
                        for (int ch = 0; ch < c; ++ch)
                            if (s[ch] > o[ch]) o[ch] = s[ch];
                    }
            }
}

/* src/layer/adaptive_avg_pool_2d.cpp:78-112 */
int orc_adaptive_avgpool2d(const float* in, int n, int ih, int iw, int c,
                           float* out, int oh, int ow) {
    if (oh <= 0 || ow <= 0 || ih % oh != 0 || iw % ow != 0) return -1;
    const int kh = ih / oh, kw = iw / ow;
    for (int b = 0; b < n; ++b)
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x)
                for (int ch = 0; ch < c; ++ch) {
                    float s = 0.0f;
                    for (int i = 0; i < kh; ++i)
                        for (int j = 0; j < kw; ++j)
                            s += in[(((size_t)b * ih + y * kh + i) * iw + x * kw + j) * c + ch];
                    out[(((size_t)b * oh + y) * ow + x) * c + ch] = s / (float)(kh * kw);
                }
    return 0;
}

/* src/layer/upsample.cpp:76-99 (Nearest4D) with inv = 1.0f/scale (:124-125) */
void orc_upsample_nearest(const float* in, int n, int ih, int iw, int c,
                          float scale_h, float scale_w, float* out, int oh,
                          int ow) {
    const float ih_inv = 1.0f / scale_h, iw_inv = 1.0f / scale_w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int y = 0; y < oh; ++y) {
            int ys = (int)((float)y * ih_inv);
            ys = ys < 0 ? 0 : (ys > ih - 1 ? ih - 1 : ys);
            for (int x = 0; x < ow; ++x) {
                int xs = (int)((float)x * iw_inv);
                xs = xs < 0 ? 0 : (xs > iw - 1 ? iw - 1 : xs);
                memcpy(out + (((size_t)b * oh + y) * ow + x) * c,
                       in + (((size_t)b * ih + ys) * iw + xs) * c, (size_t)c * sizeof(float));
            }
        }
}

void orc_cat_channels(const float* in, size_t pixels, int c_in, float* out,
                      int c_out, int c_off) {
    for (size_t p = 0; p < pixels; ++p)
        memcpy(out + p * c_out + c_off, in + p * c_in, (size_t)c_in * sizeof(float));
}

/* src/layer/cat.cpp:86-105 */
void orc_cat_axis(const float* in, const int s[4], float* out, const int o[4],
                  int axis, int offset) {
    for (int a = 0; a < s[0]; ++a)
        for (int b = 0; b < s[1]; ++b)
            for (int c = 0; c < s[2]; ++c)
                for (int d = 0; d < s[3]; ++d) {
                    int idx[4] = {a, b, c, d};
                    idx[axis] += offset;
                    out[(((size_t)idx[0] * o[1] + idx[1]) * o[2] + idx[2]) * o[3] + idx[3]] =
                        in[(((size_t)a * s[1] + b) * s[2] + c) * s[3] + d];
                }
}

/* src/layer/binary_op.cpp:52-94: Eigen broadcast(factor) tiles the input, so
 * element i of the output reads input index i % in_dim on each axis. */
/* The reference layer has add (0) and mul (2) only (binary_op.cpp:17-31).  The other codes are what pnnx's expression lowering
 * writes into BinaryOp's param "0" (src/pnnx/expand_expression.cpp:198-216: 1 sub, 3 div, 6 pow, 10 atan2; with a literal first
 * operand the reversed forms 7 rsub, 8 rdiv, 9 rpow, 11 ratan2).  No reference arithmetic exists for them: the restatement is
 * the C operator / libm function the code names, in float. */
static inline float orc_binary_apply(int op, float x, float y) {
    switch (op) {
        case 0: return x + y;
        case 1: return x - y;
        case 2: return x * y;
        case 3: return x / y;
        case 6: return powf(x, y);
        case 7: return y - x;
        case 8: return y / x;
        case 9: return powf(y, x);
        case 10: return atan2f(x, y);
        case 11: return atan2f(y, x);
        default: return x;
    }
}
static int orc_binary_known(int op) { return (op >= 0 && op <= 3) || (op >= 6 && op <= 11); }

int orc_binary_op(int op, const float* a, const int as[4], const float* b,
                  const int bs[4], float* out, const int os[4]) {
    if (!orc_binary_known(op)) return -1;
    for (int d = 0; d < 4; ++d)
        if (as[d] <= 0 || bs[d] <= 0 || os[d] % as[d] != 0 || os[d] % bs[d] != 0) return -2;
#pragma omp parallel for schedule(static)
    for (int i0 = 0; i0 < os[0]; ++i0)
        for (int i1 = 0; i1 < os[1]; ++i1)
            for (int i2 = 0; i2 < os[2]; ++i2)
                for (int i3 = 0; i3 < os[3]; ++i3) {
                    const float x = a[(((size_t)(i0 % as[0]) * as[1] + (i1 % as[1])) * as[2] + (i2 % as[2])) * as[3] + (i3 % as[3])];
                    const float y = b[(((size_t)(i0 % bs[0]) * bs[1] + (i1 % bs[1])) * bs[2] + (i2 % bs[2])) * bs[3] + (i3 % bs[3])];
                    out[(((size_t)i0 * os[1] + i1) * os[2] + i2) * os[3] + i3] = orc_binary_apply(op, x, y);
                }
    return 0;
}

/* BinaryOp's `with_scalar` form (expand_expression.cpp:206-236: params "1" = 1, "2" = the literal): out = in (op) scalar */
int orc_binary_scalar(int op, const float* in, float scalar, float* out, size_t count) {
    if (!orc_binary_known(op)) return -1;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)count; ++i) out[i] = orc_binary_apply(op, in[i], scalar);
    return 0;
}

/* UnaryOp as emitted by expand_expression.cpp:123-165 (ncnn's operator numbering).  The reference registers no such layer
 * (LoadModel returns kEmpty): the restatement is the libm function each name stands for, in float. */
int orc_unary_op(int op, const float* in, float* out, size_t count) {
    if (op < 0 || op > 17) return -1;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)count; ++i) {
        const float x = in[i];
        float y;
        switch (op) {
            case 0: y = fabsf(x); break;
            case 1: y = -x; break;
            case 2: y = floorf(x); break;
            case 3: y = ceilf(x); break;
            case 4: y = x * x; break;
            case 5: y = sqrtf(x); break;
            case 6: y = 1.0f / sqrtf(x); break;
            case 7: y = expf(x); break;
            case 8: y = logf(x); break;
            case 9: y = sinf(x); break;
            case 10: y = cosf(x); break;
            case 11: y = tanf(x); break;
            case 12: y = asinf(x); break;
            case 13: y = acosf(x); break;
            case 14: y = atanf(x); break;
            case 15: y = 1.0f / x; break;
            case 16: y = tanhf(x); break;
            default: y = log10f(x); break;
        }
        out[i] = y;
    }
    return 0;
}

int orc_activation(int act, const float* in, float* out, size_t count) {
    if (act < 1 || act > 5) return -1;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)count; ++i) {
        const float x = in[i];
        float y;
        switch (act) {
            case 1: y = x > 0.0f ? x : 0.0f; break;                 /* relu.cpp:64 cwiseMax(0) */
            case 2: y = x / (1.0f + expf(-x)); break;               /* silu.cpp:58-59 */
            case 3: y = 1.0f / (1.0f + expf(-x)); break;            /* sigmoid.cpp:64 */
            case 4: {                                               /* hard_sigmoid.cpp:70-74, alpha=1/6 beta=.5 (:17-19) */
                float t = x * (1.0f / 6.0f) + 0.5f;
                y = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
                break;
            }
            default: {                                              /* hard_swish.cpp:70-76 */
                float t = x * (1.0f / 6.0f) + 0.5f;
                t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
                y = x * t;
                break;
            }
        }
        out[i] = y;
    }
    return 0;
}

/* src/layer/batch_norm_2d.cpp:119-134: ((x - mean) * rsqrt(var + eps)) * gamma + beta */
void orc_batchnorm2d(const float* in, size_t pixels, int c, const float* mean,
                     const float* var, const float* gamma, const float* beta,
                     float eps, float* out) {
    float* inv = (float*)malloc((size_t)c * sizeof(float));
    for (int i = 0; i < c; ++i) inv[i] = 1.0f / sqrtf(var[i] + eps);
    for (size_t p = 0; p < pixels; ++p)
        for (int i = 0; i < c; ++i)
            out[p * c + i] = (in[p * c + i] - mean[i]) * inv[i] * gamma[i] + beta[i];
    free(inv);
}

/* src/layer/flatten.cpp:72-80 */
void orc_flatten_nhwc(const float* in, int n, int h, int w, int c, float* out) {
    for (int b = 0; b < n; ++b)
        for (int ch = 0; ch < c; ++ch)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x)
                    out[(((size_t)b * c + ch) * h + y) * w + x] =
                        in[(((size_t)b * h + y) * w + x) * c + ch];
}

/* src/layer/yolo_detect.cpp:204-272 for one level.  The 1x1 conv runs the
 * im2col path (kernel 1x1 is not Winograd-eligible).  Row order [h][w][a]
 * (SURVEY Q4): conv output [n][h][w][na*ne] is reshaped, not permuted
 * (:223-233); grids are reshuffled [1,na,h,w,2] -> [h][w][na][2] (:75-79). */
void orc_yolo_detect_level(const float* feat, int n, int h, int w, int cin,
                           const float* w_oihw, const float* bias, int na,
                           int ne, const float* grid_attr,
                           const float* anchor_attr, float stride, float* out,
                           int rows_total, int row_off) {
    OrcConv2d p;
    memset(&p, 0, sizeof(p));
    p.n = n; p.ih = h; p.iw = w; p.ic = cin; p.oc = na * ne; p.kh = p.kw = 1;
    p.sh = p.sw = p.dh = p.dw = 1; p.groups = 1; p.use_bias = 1;
    float* conv = (float*)malloc((size_t)n * h * w * na * ne * sizeof(float));
    orc_conv2d_im2col(&p, feat, w_oihw, bias, conv);
    const size_t rows = (size_t)h * w * na;
    for (int b = 0; b < n; ++b)
        for (size_t r = 0; r < rows; ++r) {
            const float* src = conv + ((size_t)b * rows + r) * ne;
            float* dst = out + ((size_t)b * rows_total + row_off + r) * ne;
            const size_t a = r % na, x = (r / na) % w, y = r / ((size_t)na * w);
            const float* g = grid_attr + (((a * h + y) * w + x) * 2);
            const float* ag = anchor_attr + (((a * h + y) * w + x) * 2);
            for (int e = 0; e < ne; ++e) {
                const float s = 1.0f / (1.0f + expf(-src[e]));
                if (e < 2)
                    dst[e] = (s * 2.0f + g[e]) * stride;           /* :258-259 */
                else if (e < 4)
                    dst[e] = powf(s * 2.0f, 2.0f) * ag[e - 2];     /* :261-263 */
                else
                    dst[e] = s;
            }
        }
    free(conv);
}

/* ---- letterbox + detection post-processing (test/test_yolo/test_yolo.cpp) ------------------- */
void orc_letterbox_geometry(int height_origin, int width_origin, int height_new, int width_new, int* height_resize,
                            int* width_resize, float* scale, int* padding_t, int* padding_l) {
    /* test_yolo.cpp:194-213 */
    int hr = height_new, wr = width_new;
    float s = 1.0f;
    if (height_new * width_origin < width_new * height_origin) {
        s = (float)height_new / (float)height_origin;
        wr = (int)(width_origin * s);
    } else {
        s = (float)width_new / (float)width_origin;
        hr = (int)(height_origin * s);
    }
    /* :234-241 */
    *height_resize = hr;
    *width_resize = wr;
    *scale = s;
    *padding_t = (height_new - hr) / 2;
    *padding_l = (width_new - wr) / 2;
}

void orc_letterbox_u8(const unsigned char* src, int hr, int wr, float* out, int H, int W, int pt, int pl) {
    /* :220-252: reverse channel dim, pad with 114, cast<float>, / 255.0f */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) {
                const int sy = y - pt, sx = x - pl;
                unsigned char v = 114;
                if (sy >= 0 && sy < hr && sx >= 0 && sx < wr) v = src[((size_t)sy * wr + sx) * 3 + (2 - c)];
                out[((size_t)y * W + x) * 3 + c] = (float)v / 255.0f;
            }
}

typedef struct {
    float x, y, w, h;
    int label;
    float prob;
} OrcObject;

/* test_yolo.cpp:28-59 */
static void orc_qsort_descent(OrcObject* o, int left, int right) {
    int i = left, j = right;
    const float p = o[(left + right) / 2].prob;
    while (i <= j) {
        while (o[i].prob > p) i++;
        while (o[j].prob < p) j--;
        if (i <= j) {
            const OrcObject t = o[i];
            o[i] = o[j];
            o[j] = t;
            i++;
            j--;
        }
    }
    if (left < j) orc_qsort_descent(o, left, j);
    if (i < right) orc_qsort_descent(o, i, right);
}

/* cv::Rect_<float> a & b, then area() (simpleocv) */
static float orc_intersection_area(const OrcObject* a, const OrcObject* b) {
    const float x1 = a->x > b->x ? a->x : b->x;
    const float y1 = a->y > b->y ? a->y : b->y;
    const float ax2 = a->x + a->w, bx2 = b->x + b->w;
    const float ay2 = a->y + a->h, by2 = b->y + b->h;
    const float w = (ax2 < bx2 ? ax2 : bx2) - x1;
    const float h = (ay2 < by2 ? ay2 : by2) - y1;
    if (w <= 0.0f || h <= 0.0f) return 0.0f;
    return w * h;
}

static float orc_clipf(float v, float lo, float hi) {
    const float m = v < hi ? v : hi;
    return lo > m ? lo : m;
}

int orc_yolo_postprocess(const float* pred, int rows, int ne, float prob_threshold, float nms_threshold, int agnostic,
                         const float* adjust, float* dets, int max_det) {
    const int num_class = ne - 5;
    OrcObject* objs = (OrcObject*)malloc(sizeof(OrcObject) * (size_t)(rows > 0 ? rows : 1));
    int n = 0;
    /* :341-377 */
    for (int e = 0; e < rows; ++e) {
        const float* r = pred + (size_t)e * ne;
        const float box_score = r[4];
        int class_index = -1;
        float class_score = -FLT_MAX;
        for (int k = 0; k < num_class; ++k) {
            const float score = r[k + 5];
            if (score > class_score) {
                class_index = k;
                class_score = score;
            }
        }
        const float confidence = box_score * class_score;
        if (confidence >= prob_threshold) {
            const float x0 = r[0] - r[2] * 0.5f, y0 = r[1] - r[3] * 0.5f;
            const float x1 = r[0] + r[2] * 0.5f, y1 = r[1] + r[3] * 0.5f;
            objs[n].x = x0;
            objs[n].y = y0;
            objs[n].w = x1 - x0;
            objs[n].h = y1 - y0;
            objs[n].label = class_index;
            objs[n].prob = confidence;
            ++n;
        }
    }
    /* :380 */
    if (n > 0) orc_qsort_descent(objs, 0, n - 1);
    /* :68-104 */
    int* picked = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    float* areas = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int np_ = 0;
    for (int i = 0; i < n; ++i) areas[i] = objs[i].w * objs[i].h;
    for (int i = 0; i < n; ++i) {
        int keep = 1;
        for (int j = 0; j < np_; ++j) {
            const OrcObject* b = &objs[picked[j]];
            if (!agnostic && objs[i].label != b->label) continue;
            const float inter_area = orc_intersection_area(&objs[i], b);
            const float union_area = areas[i] + areas[picked[j]] - inter_area;
            if (inter_area / union_area > nms_threshold) keep = 0;
        }
        if (keep) picked[np_++] = i;
    }
    /* :386-416 */
    for (int i = 0; i < np_ && i < max_det; ++i) {
        const OrcObject* o = &objs[picked[i]];
        float* d = dets + (size_t)i * 6;
        if (adjust) {
            const float pl = adjust[0], pt = adjust[1], sc = adjust[2];
            float x0 = (o->x - pl) / sc, y0 = (o->y - pt) / sc;
            float x1 = (o->x + o->w - pl) / sc, y1 = (o->y + o->h - pt) / sc;
            x0 = orc_clipf(x0, 0.0f, adjust[3] - 1.0f);
            y0 = orc_clipf(y0, 0.0f, adjust[4] - 1.0f);
            x1 = orc_clipf(x1, 0.0f, adjust[3] - 1.0f);
            y1 = orc_clipf(y1, 0.0f, adjust[4] - 1.0f);
            d[0] = x0; d[1] = y0; d[2] = x1 - x0; d[3] = y1 - y0;
        } else {
            d[0] = o->x; d[1] = o->y; d[2] = o->w; d[3] = o->h;
        }
        d[4] = o->prob;
        d[5] = (float)o->label;
    }
    free(areas);
    free(picked);
    free(objs);
    return np_;
}
