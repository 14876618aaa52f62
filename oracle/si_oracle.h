/*
 * si_oracle.h -- CPU restatement of SimpleInfer's operator algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; nothing under simpleinfer_amd/ links, imports or calls it.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose algorithm it restates.  All tensors are dense NHWC fp32, exactly the
 * in-memory layout the reference uses (src/engine_impl.cpp:182-189).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - GEMM pack-4: pinned by the reference's known-answer test
 *     (test/test_3rdparty/test_gemm.cpp:56-91, all-ones operands, exact).
 *   - Winograd F(2,3) / im2col conv / pool / upsample / cat / activations /
 *     linear / flatten / batchnorm: pinned by the reference tests' in-test
 *     naive loops (test/test_layer/\*.cpp) restated in tests/, at the
 *     reference's own tolerances, and cross-checked against torch-CPU golden
 *     fixtures (tests/golden/).
 *   - Whole-graph YOLOv5s output: PARITY UNPINNED by the reference (it stores
 *     no golden outputs and its model files are an absent submodule).
 */
#ifndef SI_ORACLE_H_
#define SI_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- GEMM (src/layer/simd/gemm.cpp) ------------------------------------ */

/* C[M,N] = A[M,K](lda) * Bpacked[N/4][K][4]; C overwritten; k-ordered fmaf
 * chain per element, as the 4x12 / 4x4 micro-kernels compute it
 * (gemm.cpp:72-157, 295-385).  */
void orc_gemm_pack4_f32(size_t M, size_t N, size_t K, const float* A,
                        size_t lda, const float* Bp, float* C, size_t ldc);

/* scalar "Ref" variant with separate mul+add (gemm.cpp:405-424) */
void orc_gemm_pack4_f32_ref(size_t M, size_t N, size_t K, const float* A,
                            size_t lda, const float* Bp, float* C, size_t ldc);

/* ---- Winograd F(2,3) (src/layer/simd/winograd_helper.cpp) -------------- */

/* HWIO [3][3][ic][oc] -> [16][oc_up4/4][ic][4]   (winograd_helper.cpp:40-143) */
void orc_wino23_transform_kernel_pack4(const float* hwio, size_t ic, size_t oc,
                                       float* dst);
/* NHWC image [ih][iw][ic] -> [16][tiles][ic]   (winograd_helper.cpp:413-580).
 * q1_bug != 0 replicates the `row < ow` tail-row condition of :540;
 * q1_bug == 0 uses the intended `row < oh`.  */
void orc_wino23_transform_input(const float* src, size_t ih, size_t iw,
                                size_t ic, int pad, float* dst,
                                size_t dst_stride, int q1_bug);
/* [16][tiles][oc] -> NHWC [oh][ow][oc]   (winograd_helper.cpp:806-874) */
void orc_wino23_transform_output(const float* src, size_t src_stride,
                                 float* dst, size_t oh, size_t ow, size_t oc);
/* dst[s][c] += bias[c]   (src/layer/simd/binary.cpp:38-53) */
void orc_add_bias_nhwc(const float* bias, size_t spatial, size_t oc,
                       float* dst);

/* ---- Conv2d (src/layer/conv_2d.cpp) ------------------------------------ */

typedef struct OrcConv2d {
    int n, ih, iw, ic;      /* input NHWC */
    int oc, kh, kw;         /* weight OIHW: [oc][ic/groups][kh][kw] */
    int sh, sw, dh, dw;
    int pt, pb, pl, pr;
    int groups;
    int use_bias;
} OrcConv2d;

void orc_conv2d_out_shape(const OrcConv2d* p, int* oh, int* ow);

/* Conv2d::Forward dispatch (conv_2d.cpp:108-118): Winograd23 when eligible
 * (:182-205), else im2col+GEMM (:207-283) or grouped im2col (:285-380).
 * w_oihw is the pnnx attribute layout; the OIHW->HWIO shuffle of :126-150 is
 * performed inside.  */
int orc_conv2d_forward(const OrcConv2d* p, const float* in,
                       const float* w_oihw, const float* bias, float* out);
/* force the im2col path (what test_conv_2d.cpp exercises: InitWinograd is
 * not called there) */
int orc_conv2d_im2col(const OrcConv2d* p, const float* in, const float* w_oihw,
                      const float* bias, float* out);
/* force the Winograd path; returns -1 if not eligible */
int orc_conv2d_winograd23(const OrcConv2d* p, const float* in,
                          const float* w_oihw, const float* bias, float* out,
                          int q1_bug);
/* The naive loop the reference tests use as THEIR oracle
 * (test/test_layer/test_conv_2d.cpp:100-131, test_winograd.cpp:101-131),
 * with a double accumulator (acc64 != 0) or the tests' float accumulator. */
int orc_conv2d_naive(const OrcConv2d* p, const float* in, const float* w_oihw,
                     const float* bias, float* out, int acc64);
/* NOT a reference algorithm: the device implicit-GEMM kernel's accumulation
 * order as a scalar fmaf chain (bit-exact predictor of conv_igemm.hip; see
 * the definition).  */
int orc_conv2d_chain(const OrcConv2d* p, const float* in, const float* w_oihw,
                     const float* bias, float* out);

/* ---- other layers ------------------------------------------------------- */

/* y = x W^T + b, x:[n,in], W:[out,in]   (src/layer/linear.cpp:74-117) */
void orc_linear(const float* x, int n, int in_f, const float* w,
                const float* b, int out_f, float* y);

/* window max, pad value = lowest()   (src/layer/max_pool_2d.cpp:77-121) */
void orc_maxpool2d(const float* in, int n, int ih, int iw, int c, int kh,
                   int kw, int sh, int sw, int dh, int dw, int pt, int pl,
                   float* out, int oh, int ow);

/* uniform k = in/out mean pooling   (src/layer/adaptive_avg_pool_2d.cpp:54-116) */
int orc_adaptive_avgpool2d(const float* in, int n, int ih, int iw, int c,
                           float* out, int oh, int ow);

/* nearest, src = clamp((int)((float)dst * (1/scale)))   (src/layer/upsample.cpp:76-99,164-165) */
void orc_upsample_nearest(const float* in, int n, int ih, int iw, int c,
                          float scale_h, float scale_w, float* out, int oh,
                          int ow);

/* copy `in` [n,h,w,c_in] into channels [c_off, c_off+c_in) of out [n,h,w,c_out]
 * -- one slice-assign of Cat::Forward for NHWC dim 3 (src/layer/cat.cpp:86-105) */
void orc_cat_channels(const float* in, size_t pixels, int c_in, float* out,
                      int c_out, int c_off);
/* generic rank-4 concat along NHWC axis `axis` (cat.cpp:86-105) */
void orc_cat_axis(const float* in, const int in_shape[4], float* out,
                  const int out_shape[4], int axis, int offset);

/* op: 0 add, 2 mul (src/layer/binary_op.cpp:17-31); broadcast by integer
 * factors out/in per dim (:60-75) */
int orc_binary_op(int op, const float* a, const int a_shape[4], const float* b,
                  const int b_shape[4], float* out, const int out_shape[4]);
/* BinaryOp `with_scalar` form and UnaryOp, as pnnx's expression lowering emits them (src/pnnx/expand_expression.cpp:123-165,
 * 198-244); no reference layer exists for either -- float libm semantics. */
int orc_binary_scalar(int op, const float* in, float scalar, float* out, size_t count);
int orc_unary_op(int op, const float* in, float* out, size_t count);

/* act: 1 relu (relu.cpp:55-67), 2 silu (silu.cpp:49-62), 3 sigmoid
 * (sigmoid.cpp:55-67), 4 hardsigmoid (hard_sigmoid.cpp:62-78), 5 hardswish
 * (hard_swish.cpp:62-80) */
int orc_activation(int act, const float* in, float* out, size_t count);

/* (x-mean)*rsqrt(var+eps)*gamma+beta per channel (src/layer/batch_norm_2d.cpp:84-137) */
void orc_batchnorm2d(const float* in, size_t pixels, int c, const float* mean,
                     const float* var, const float* gamma, const float* beta,
                     float eps, float* out);

/* rank-4: NHWC -> NCHW then flatten (src/layer/flatten.cpp:55-88) */
void orc_flatten_nhwc(const float* in, int n, int h, int w, int c, float* out);

/* One level of YoloDetect::Forward (src/layer/yolo_detect.cpp:204-272):
 * feat [n,h,w,cin] -> 1x1 conv (w_oihw [na*ne, cin,1,1], bias) -> sigmoid ->
 * rows [n][h*w*na][ne] written at row offset `row_off` of out [n][rows_total][ne];
 * xy=(2s+grid)*stride, wh=(2s)^2*anchor.  grid/anchor are the pnnx attrs
 * [1,na,h,w,2] (reshuffled to [h][w][na][2] as :60-120 does). */
void orc_yolo_detect_level(const float* feat, int n, int h, int w, int cin,
                           const float* w_oihw, const float* bias, int na,
                           int ne, const float* grid_attr,
                           const float* anchor_attr, float stride, float* out,
                           int rows_total, int row_off);

/* Letterbox of PreProcess after cv::resize (test/test_yolo/test_yolo.cpp:194-259): geometry (:194-213, 234-241) and
 * the reverse(bgr->rgb) / pad(114) / cast / divide-by-255 expression (:220-252) for one image. */
void orc_letterbox_geometry(int height_origin, int width_origin, int height_new, int width_new, int* height_resize,
                            int* width_resize, float* scale, int* padding_t, int* padding_l);
void orc_letterbox_u8(const unsigned char* resized_bgr, int hr, int wr, float* out, int height_new, int width_new,
                      int padding_t, int padding_l);

/* Detection post-processing of ONE image exactly as test_yolo.cpp:337-428 does it: confidence filter (:341-377),
 * qsort_descent_inplace (:28-66, the same unstable quicksort), nms_sorted_bboxes (:68-104; rect intersection as
 * simpleocv's `a & b`, an absent ncnn-derived submodule: empty when width <= 0 or height <= 0), and -- when adjust !=
 * NULL {padding_l, padding_t, scale, cols, rows} -- the un-letterbox + clip of :390-416.
 * pred [rows][ne]; dets [max_det][6] = {x, y, w, h, prob, label}.  Returns the number of boxes picked (only the
 * first max_det are stored). */
int orc_yolo_postprocess(const float* pred, int rows, int ne, float prob_threshold, float nms_threshold, int agnostic,
                         const float* adjust, float* dets, int max_det);

int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif

#endif /* SI_ORACLE_H_ */
