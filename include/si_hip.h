/*
 * si_hip.h -- thin C-ABI over the hand-written gfx950 (MI355X / CDNA4) HIP
 * kernels that replace SimpleInfer's src/layer operator set.
 *
 * Plain pointers and sizes only: no C++ or torch types cross this boundary.
 * All activation tensors are NHWC fp32 in HBM.  Every tensor argument carries
 * a *pixel stride* `ld` (elements between consecutive pixels, >= C) so a
 * tensor can live inside a wider buffer -- this is how the engine makes
 * torch.cat zero-copy (producers write straight into their channel slice).
 *
 * Return value: 0 on success; a positive hipError_t; or a negative SI_E_*.
 * Every launch is asynchronous on `stream` (a hipStream_t; NULL = default
 * stream).  Nothing here allocates or synchronises unless its name says so,
 * so the whole forward can be captured into a hipGraph.
 *
 * Each entry point cites the reference routine (file:line under
 * /root/reference) it replaces.
 */
#ifndef SI_HIP_H_
#define SI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* si_stream_t; /* hipStream_t */
typedef void* si_event_t;  /* hipEvent_t  */
typedef void* si_graph_t;  /* hipGraphExec_t */

#define SI_E_BADARG (-1)
#define SI_E_UNSUPPORTED (-2)
#define SI_E_NODEVICE (-3)

/* activation codes used by fused epilogues and si_hip_activation_f32 */
enum {
    SI_ACT_NONE = 0,
    SI_ACT_RELU = 1,        /* src/layer/relu.cpp:55-67 */
    SI_ACT_SILU = 2,        /* src/layer/silu.cpp:49-62 */
    SI_ACT_SIGMOID = 3,     /* src/layer/sigmoid.cpp:55-67 */
    SI_ACT_HARDSIGMOID = 4, /* src/layer/hard_sigmoid.cpp:62-78 */
    SI_ACT_HARDSWISH = 5,   /* src/layer/hard_swish.cpp:62-80 */
    SI_ACT_LEAKYRELU = 6    /* north_star extension (no reference layer); slope in desc */
};

/* ---- runtime ----------------------------------------------------------- */
const char* si_hip_version(void);
const char* si_hip_error_string(int code);
int si_hip_device_count(int* count);
int si_hip_set_device(int device);
int si_hip_get_device(int* device);
/* name (<=255 chars), CU count, HBM bytes, core clock kHz */
int si_hip_device_info(int device, char* name, int* cus, size_t* hbm_bytes, int* clock_khz);
int si_hip_malloc(void** ptr, size_t bytes);
int si_hip_free(void* ptr);
int si_hip_host_alloc(void** ptr, size_t bytes); /* pinned */
int si_hip_host_free(void* ptr);
int si_hip_host_register(void* ptr, size_t bytes); /* pin caller-owned host memory in place (hipHostRegister) */
int si_hip_host_unregister(void* ptr);
int si_hip_memset_async(void* ptr, int value, size_t bytes, si_stream_t stream);
int si_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, si_stream_t stream);
int si_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, si_stream_t stream);
int si_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, si_stream_t stream);
int si_hip_stream_create(si_stream_t* stream);
/* ... with a priority: -1 the device's lowest, +1 its highest, 0 the default (round 5: engine option detect_priority) */
int si_hip_stream_create_priority(si_stream_t* stream, int level);
int si_hip_stream_destroy(si_stream_t stream);
int si_hip_stream_sync(si_stream_t stream);
int si_hip_device_sync(void);
int si_hip_event_create(si_event_t* ev);
int si_hip_event_destroy(si_event_t ev);
int si_hip_event_record(si_event_t ev, si_stream_t stream);
int si_hip_event_sync(si_event_t ev);
int si_hip_event_elapsed_ms(si_event_t start, si_event_t stop, float* ms);
int si_hip_stream_wait_event(si_stream_t stream, si_event_t ev); /* later work on `stream` waits for `ev` */
/* Device memory shared between the per-GPU processes of one node -- the transport of the direct (non-ring) output
 * all-gather of include/si_shard.h (no reference counterpart: SimpleInfer is single-process).
 * `handle` is SI_IPC_HANDLE_BYTES opaque bytes to be carried to the peer process by any host transport. */
#define SI_IPC_HANDLE_BYTES 64
int si_hip_ipc_get_mem_handle(void* dptr, void* handle);
int si_hip_ipc_open_mem_handle(const void* handle, void** dptr);
int si_hip_ipc_close_mem_handle(void* dptr);
int si_hip_enable_peer_access(int peer_device); /* current device -> peer_device; 0 when already enabled or same device */
/* A device index only means something inside one process (HIP_VISIBLE_DEVICES may differ per rank); the PCI bus id
 * ("0000:c1:00.0", buf of >= 16 bytes) names the GPU for every process of the node.  _by_pci_bus_id: index of the
 * visible device with that id, -1 when it is hidden from this process. */
int si_hip_device_pci_bus_id(int device, char* buf, int len);
int si_hip_device_by_pci_bus_id(const char* bus_id);
/* stream capture -> executable graph (replaces the CGraph pipeline of
 * src/engine_impl.cpp:336-437 for launch-bound small batches) */
int si_hip_graph_begin_capture(si_stream_t stream);
int si_hip_graph_end_capture(si_stream_t stream, si_graph_t* exec);
int si_hip_graph_launch(si_graph_t exec, si_stream_t stream);
int si_hip_graph_destroy(si_graph_t exec);

/* ---- Conv2d ------------------------------------------------------------ */
/* Replaces Conv2d::ForwardIm2Col / ForwardIm2ColWithGroup / ForwardWinograd23
 * (src/layer/conv_2d.cpp:207-283, :285-380, :382-487) and the separate
 * AddBiasNHWC / activation / residual passes (src/layer/simd/binary.cpp:38-53,
 * src/layer/silu.cpp:49-62, src/layer/binary_op.cpp:52-94) with one
 * implicit-GEMM kernel on v_mfma_f32_32x32x2_f32:
 *   y = act2( act1(conv(x, w) + bias) + residual )
 */
typedef struct SiConv2dDesc {
    int n, ih, iw, ic;  /* input  NHWC, ic = total input channels */
    int in_ld;          /* input pixel stride (elements) */
    int oh, ow, oc;     /* output NHWC, oc = total output channels */
    int out_ld;         /* output pixel stride */
    int kh, kw, sh, sw, dh, dw;
    int pt, pl;         /* zero padding on top / left (bottom/right implied by oh/ow) */
    int groups;
    int has_bias;
    int act1;           /* applied to conv+bias */
    int has_residual;   /* residual tensor [n,oh,ow,oc] with pixel stride res_ld */
    int res_ld;
    int act2;           /* applied after the residual add */
    float act_param;    /* leaky-relu slope */
    /* -- trailing fields; all-zero (memset / fewer initialisers) = the defaults ------------------------------------------------------ */
    const struct SiConvPlan* plan; /* kernel-form choices for THIS call (tests, sweeps, an engine's plan); NULL: the shape / launch-size policy */
    unsigned int* range_flag;      /* f32_split entry points (si_hip_conv2d_split3_* / _wino23_split_f32 / _stem_split3_f32) only: NULL, or a word the device can
                                    * write (device memory or pinned host memory) that the kernel sets to 1 when an operand left fp16's range
                                    * on its way through the split -- see the f32_split section below.  Never written otherwise. */
} SiConv2dDesc;

/* Kernel-form choices of ONE call (round 6; VERDICT r05 item 7: these were process-global setters and environment switches until then).  Every
 * form of a family computes an output element as the same fma chain in the same k order -- the choice never changes a result (tests/test_gpu_tiles.py,
 * test_gpu_f16.py hold every form to the same bits) -- so a plan is a performance choice only.  A field at its default leaves that choice to the
 * policy (launch size, CU count, shape).  Initialise with SI_CONV_PLAN_DEFAULT. */
typedef struct SiConvPlan {
    int f32_tile;        /* si_hip_conv2d_f32 family: -1 policy, 0..22 a tile of conv_igemm.hip's table (a few ids are retired: SI_E_BADARG) */
    int wino23_form;     /* si_hip_conv2d_wino23_f32: 0 policy, 32 / 16 the 32-tile / 16-tile work unit (16 needs ic % 32 == 0) */
    int wino23_ocg;      /* ... 32-tile form: 0 policy, 1 / 2 output-channel groups of 32 per workgroup (2 needs oc % 64 == 0) */
    int f16_tile;        /* si_hip_conv2d_f16 family: -1 policy; 0-2 the one-stage kernel (64x64, 128x64, 128x128); 3, 7, 9, 10, 11 the kernels
                          * with lane-order weights (128x128 as 2x2 waves, 128x32 as 4x1, 128x128 as 1x4, 64x128 as 1x4, 9 at three waves per SIMD) */
    int f16_detect_tile; /* si_hip_conv2d_yolo_f16: -1 policy (the Detect tile where it applies), 0 the generic tiles */
    int f16_s2c32;       /* 3x3 over one 32 / 64-channel block: -1 policy (the persistent patch kernel), 0 the generic tiles */
    int f16_slab;        /* 3x3 s1 over 128 / 256 channels: -1 policy (row slabs), 0 the generic tiles */
    int f16_slab_w2;     /* ... the slab kernel's 128-channel form: -1 policy (two waves per SIMD), 0 one wave per SIMD */
    int f16_pw_patch;    /* 64 / 32-channel bottleneck pair: -1 policy (si_hip_conv2d_pw_slab_f16_supported may say 2), 0 never 2 */
    int split3_bm;       /* si_hip_conv2d_split3_f32 family: 0 policy (32-row tiles; 64 for <= 64 output columns; a Detect level over 128 / 256 channels as
                          * 64-pixel runs of the output from four such tiles per CU on), 32 / 64 / 128 rows per workgroup tile (a Detect level then takes
                          * the generic kernel's decode epilogue), -1 the 64-pixel Detect tile whatever the launch size (other convs: the policy) */
} SiConvPlan;
#define SI_CONV_PLAN_DEFAULT { -1, 0, 0, -1, -1, -1, -1, -1, -1, 0 }

/* Weight layout expected by the kernels: [oc][kh][kw][icg_pad] ("OHWI",
 * K = kh*kw*icg_pad contiguous per output channel), icg_pad = ic/groups
 * rounded up to a multiple of 4 when ic/groups is not one (zero filled), so
 * that every 16-byte K-vector stays inside one (kh,kw) tap.  (Stem shapes -- 1..3 input channels, see
 * csrc/hip/conv_smallc.hip -- use a transposed [kh][kw*ic padded][oc padded] image instead; callers never
 * need to know: always size with si_hip_conv2d_weight_elems and fill with si_hip_conv2d_pack_weight_host, passing
 * a descriptor whose ic, oc, kh, kw, sh, sw, dh, dw, groups equal the ones used at launch -- the layout is a function
 * of exactly those fields.)
 * si_hip_conv2d_weight_elems returns the element count of that layout and
 * si_hip_conv2d_pack_weight_host re-lays an OIHW host tensor (the pnnx
 * attribute layout; replaces the OIHW->HWIO shuffle of conv_2d.cpp:126-150). */
size_t si_hip_conv2d_weight_elems(const SiConv2dDesc* d);
int si_hip_conv2d_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, float* w_packed);
int si_hip_conv2d_f32(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                      const float* residual, float* out, si_stream_t stream);
/* A 1x1 stride-1 conv whose input is torch.cat(..., nn.Upsample(x, nearest), ...) reads the upsampled channels straight
 * from the LOW-RESOLUTION tensor (dual-source A rows): channels [c0, c0 + c) of the conv's K axis come from `src`
 * [n, ih, iw, c] (pixel stride ld) at src = clamp(int(float(dst) * inv_scale)) -- the reference's index rule,
 * src/layer/upsample.cpp:85-92 -- the remaining channels from `in` (the concat buffer) as usual.  Replaces Upsample::Forward +
 * the Cat copy (src/layer/upsample.cpp:101-170, cat.cpp:59-108) in front of such a conv: the upsampled tensor is never
 * written.  c0 and c multiples of 32.  split_oc > 0: sibling-fused form (as si_hip_conv2d_split_f32). */
typedef struct SiConv2dUpsampledSource {
    const float* src;
    int ih, iw, c, ld;
    int c0;
    float inv_scale_h, inv_scale_w;
} SiConv2dUpsampledSource;
int si_hip_conv2d_upcat_f32(const SiConv2dDesc* d, const float* in, const SiConv2dUpsampledSource* up,
                            const float* w_packed, const float* bias, float* out, int split_oc, float* out2,
                            int out2_ld, si_stream_t stream);
/* 1 when si_hip_conv2d_upcat_f32 can serve this problem given 16-byte aligned buffers (shapes, strides, channel
 * granularity, both tensors below 4 GiB; up->src is not looked at), else 0.  A scheduler that wants to drop the
 * upsample launch asks this BEFORE doing so: the fused form has no fallback at Forward() time. */
int si_hip_conv2d_upcat_supported(const SiConv2dDesc* d, const SiConv2dUpsampledSource* up);
/* name of the kernel instantiation si_hip_conv2d_f32 would launch for this problem (exactly as rocprofv3 prints
 * it, minus the namespace), so profiles can be joined with per-layer timings.  _form: 0 si_hip_conv2d_f32 /
 * _split_f32, 1 si_hip_conv2d_upcat_f32, 2 si_hip_conv2d_yolo_f32 */
const char* si_hip_conv2d_kernel_name(const SiConv2dDesc* d, const float* in);
const char* si_hip_conv2d_kernel_name_form(const SiConv2dDesc* d, const float* in, int form);
/* Tile policy of si_hip_conv2d_f32's implicit-GEMM kernels: the workgroup tile follows the launch size and the CU count (small batches run
 * 32-row tiles on the 16x16x4 MFMA so that the chip is covered); d->plan->f32_tile = 0..22 (table in conv_igemm.hip; a few ids are retired:
 * SI_E_BADARG) forces one tile for THAT call -- tuning sweeps and the tests that hold every tile to the same bits.  The tile never changes a
 * result: all tiles accumulate an output element as one fma chain in the same k order.  (Round 6: there is no process-global setter any more.) */

/* ---- fp32 convolution on the fp16 matrix cores by operand splitting (round 5, csrc/hip/conv_split3.hip; OPT-IN) -------------------
 * Every operand as two fp16 halves, a = a_hi + 2^-11 a_lo (22 significant bits), a product from three exact fp16 MFMA products
 * accumulated in fp32 in two accumulator sets (Ootomo & Yokota 2022).  fp32 tensors in and out; another arithmetic than the fp32
 * kernels (not bit-compatible with them).  Dense convs with ic % 32 == 0, at most 32 taps.  The weights are split
 * once: two lane-order fp16 images (_weight_elems counts halves).  Same descriptor / epilogue convention as si_hip_conv2d_f32.
 * RANGE (round 6, VERDICT r05 missing 2): a value whose magnitude rounds to fp16 infinity (|x| >= 65520; for the Winograd form below the
 * TRANSFORMED input B^T d B, i.e. sums of four inputs, and U = G g G^T) cannot be split.
 *   - weights: _pack_weight_host returns SI_E_UNSUPPORTED when a weight (a U value) is not finite or out of range -- the caller keeps the layer
 *     on the fp32 kernels;
 *   - activations: such a value becomes Inf in its hi half and Inf / NaN in its lo half, so every output it contributes to leaves the matrix
 *     cores non-finite; the kernels test their combined accumulators (before bias / activation, which could hide it: relu(NaN) = 0) and, when one is
 *     not finite, write 1 to d->range_flag (when given).  The cost when nothing trips is one v_cmp_class per output element in the epilogue; a
 *     genuinely non-finite fp32 result trips the flag as well.  What the caller does then is its policy: the engine (f32_split option) re-runs
 *     the step with that layer on the true-fp32 kernels and keeps it there (csrc/host/engine_impl.cpp, Engine::Sync).
 * Small magnitudes: below 6.1e-5 the hi half is an fp16 subnormal; the pair still resolves 2^-35 ~ 2.9e-11 ABSOLUTE, so tensors whose scale is
 * >= ~1e-3 keep fp32-class relative accuracy and a tensor whose every value is tiny (scale 1e-6) degrades to ~1e-5 relative. */
int si_hip_conv2d_split3_supported(const SiConv2dDesc* d);
size_t si_hip_conv2d_split3_weight_elems(const SiConv2dDesc* d);
int si_hip_conv2d_split3_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, void* w_packed);
int si_hip_conv2d_split3_f32(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, const float* residual, float* out,
                             si_stream_t stream);
/* ... with a split destination, as si_hip_conv2d_split_f32 (two sibling 1x1 convs as ONE over the concatenated filters: YOLOv5 C3's cv1 | cv2):
 * output channels [0, split_oc) to `out` (stride d->out_ld), [split_oc, d->oc) to `out2` (stride out2_ld); split_oc a multiple of 32 (round 6) */
int si_hip_conv2d_split3_split_f32(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, float* out, int split_oc,
                                   float* out2, int out2_ld, si_stream_t stream);
/* ... and with dual-source rows, as si_hip_conv2d_upcat_f32 (a 1x1 conv over torch.cat(..., nn.Upsample(x, nearest), ...) reads the upsampled
 * channels from the low-resolution tensor at the reference's source pixel, src/layer/upsample.cpp:85-92): up->c0 and up->c multiples of 64 (the
 * kernel's K-tile), ic % 64 == 0, more than 64 output columns; split_oc > 0: the sibling-fused form.  _supported: shapes only (round 6) */
int si_hip_conv2d_split3_upcat_supported(const SiConv2dDesc* d, const SiConv2dUpsampledSource* up);
int si_hip_conv2d_split3_upcat_f32(const SiConv2dDesc* d, const float* in, const SiConv2dUpsampledSource* up, const void* w_packed, const float* bias,
                                   float* out, int split_oc, float* out2, int out2_ld, si_stream_t stream);

/* ---- fused Winograd F(2x2,3x3) with its plane GEMMs on the fp16 matrix cores by operand splitting (round 5, late; csrc/hip/conv_wino23_split.hip;
 * OPT-IN: engine option f32_split).  si_hip_conv2d_wino23_f32's kernel around another channel loop: the transformed input V and the
 * filter image U = G g G^T as two fp16 halves each, three fp16 MFMA products per fp32 product, fp32 accumulation in two accumulator
 * sets.  fp32 tensors in and out; another arithmetic than the fp32 Winograd kernel.  Same eligibility as si_hip_conv2d_wino23_eligible;
 * _weight_elems counts halves (two images). */
int si_hip_conv2d_wino23_split_supported(const SiConv2dDesc* d);
size_t si_hip_conv2d_wino23_split_weight_elems(const SiConv2dDesc* d);
int si_hip_conv2d_wino23_split_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, void* u_packed);
int si_hip_conv2d_wino23_split_f32(const SiConv2dDesc* d, const float* in, const void* u_packed, const float* bias, const float* residual,
                                   float* out, si_stream_t stream);

/* ---- Winograd F(2x2,3x3) for 3x3 stride-1 convolutions --------------------------------------------------------
 * One fused kernel replacing the reference's four-pass Conv2d::ForwardWinograd23 (src/layer/conv_2d.cpp:382-487):
 * Conv3x3s1Winograd23TransformInput (src/layer/simd/winograd_helper.cpp:413-580), the 16 GemmPack4F32 calls
 * (src/layer/simd/gemm.cpp:295-385), Conv3x3s1Winograd23TransformOutput (:806-874) and AddBiasNHWC
 * (src/layer/simd/binary.cpp:38-53) -- plus the fused activation / residual epilogue of si_hip_conv2d_f32.  The
 * transformed tensors live in LDS / registers only.  Same arithmetic as the reference path (same B, G, A matrices
 * and evaluation order), 2.25x fewer MFMA flops than the direct kernel.
 * Eligibility (shape only): 3x3, stride 1, dilation 1, groups 1, pad 0 or 1 on all sides (the reference's condition,
 * conv_2d.cpp:183-187) and ic % 16 == 0, oc % 32 == 0.  The filter is pre-transformed once with
 * si_hip_conv2d_wino23_pack_weight_host into U = G g G^T, 16 * ic * oc floats in the kernel's own operand order
 * ([plane row][ic / 16][oc / 32][step][oc % 32][ic parity][plane column]: opaque to the caller; replaces
 * Conv3x3s1Winograd23TransformKernelPack4, winograd_helper.cpp:40-143).  `bias` must be 16-byte aligned; `out` /
 * `residual` rows that are not get scalar stores. */
int si_hip_conv2d_wino23_eligible(const SiConv2dDesc* d);
/* The kernel's work-unit form -- 32 tiles on v_mfma_f32_32x32x2_f32 or 16 tiles on v_mfma_f32_16x16x4_f32 (half-size units for launches that
 * do not fill the chip evenly; needs ic % 32 == 0) -- follows a round model of the launch; d->plan->wino23_form = 16 / 32 forces one for that
 * call, ->wino23_ocg the output groups per workgroup.  The form never changes a result. */
/* eligible AND measured faster than si_hip_conv2d_f32 on MI355X (currently: ic >= 32) */
int si_hip_conv2d_wino23_preferred(const SiConv2dDesc* d);
size_t si_hip_conv2d_wino23_weight_elems(const SiConv2dDesc* d);
int si_hip_conv2d_wino23_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, float* u);
int si_hip_conv2d_wino23_f32(const SiConv2dDesc* d, const float* in, const float* u, const float* bias,
                             const float* residual, float* out, si_stream_t stream);

/* The same layers through fused Winograd F(4x4, 3x3) -- the larger tile BASELINE.json's north_star names; the
 * reference itself stops at F(2,3).  6x6 transform domain, 4x fewer MFMA flops than the direct kernel, standard
 * matrices for the points {0, +-1, +-2, inf}; fp32 throughout (measured error vs a float64 direct convolution: a few
 * 1e-6 of the output scale, inside the 1e-4 parity bar).  Same eligibility rule, same epilogue, same calling
 * convention as the wino23 entry points; U = G g G^T is 36 * ic * oc floats. */
int si_hip_conv2d_wino43_eligible(const SiConv2dDesc* d);
int si_hip_conv2d_wino43_preferred(const SiConv2dDesc* d);
size_t si_hip_conv2d_wino43_weight_elems(const SiConv2dDesc* d);
int si_hip_conv2d_wino43_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, float* u);
int si_hip_conv2d_wino43_f32(const SiConv2dDesc* d, const float* in, const float* u, const float* bias,
                             const float* residual, float* out, si_stream_t stream);

/* Two convolutions that read the SAME input with the same geometry (YOLOv5 C3: cv1 and cv2, both 1x1) run as one launch:
 * the weights / biases are concatenated along oc by the caller (d->oc = oc_a + oc_b); output channels [0, split_oc)
 * are written to `out` (stride d->out_ld) and [split_oc, d->oc) to `out2` (stride out2_ld).  split_oc must be a
 * multiple of 32; groups == 1; no residual.  Replaces two Conv2d::Forward calls (src/layer/conv_2d.cpp:108-118). */
int si_hip_conv2d_split_f32(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                            float* out, int split_oc, float* out2, int out2_ld, si_stream_t stream);

/* Detect level in one launch: conv (normally the 1x1 of YoloDetect, src/layer/yolo_detect.cpp:214) with the sigmoid /
 * grid / anchor decode of si_hip_yolo_decode_f32 and the concat into detect_out [n][rows_total][ne] done in the
 * conv epilogue -- the conv output never goes to HBM.  d->oc must equal na*ne; d->out_ld is ignored.  Returns
 * SI_E_UNSUPPORTED when the shape is not eligible for the fast kernel (caller then runs conv + decode). */
typedef struct SiYoloLevel {
    int na, ne;       /* anchors per cell, elements per anchor (85) */
    int rows_total;   /* rows of the detect tensor per image (25200) */
    int row_off;      /* first row of this level */
    float stride;
} SiYoloLevel;
int si_hip_conv2d_yolo_f32(const SiConv2dDesc* d, const float* in, const float* w_packed, const float* bias,
                           const SiYoloLevel* level, const float* grid_hwa2, const float* anchor_hwa2,
                           float* detect_out, si_stream_t stream);
/* the same on the f32_split arithmetic (si_hip_conv2d_split3_f32 above: weights from si_hip_conv2d_split3_pack_weight_host; channel counts
 * that are multiples of 64, more than 64 columns) -- engine option f32_split */
int si_hip_conv2d_split3_yolo_f32(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias,
                                  const SiYoloLevel* level, const float* grid_hwa2, const float* anchor_hwa2,
                                  float* detect_out, si_stream_t stream);

/* ---- Linear ------------------------------------------------------------ */
/* y[n,out] = x[n,in] W[out,in]^T + b   (src/layer/linear.cpp:74-117) */
int si_hip_linear_f32(const float* x, int n, int in_features, const float* w, const float* bias, int out_features,
                      float* y, si_stream_t stream);

/* ---- pooling / resampling ---------------------------------------------- */
typedef struct SiPool2dDesc {
    int n, ih, iw, c, in_ld;
    int oh, ow, out_ld;
    int kh, kw, sh, sw, dh, dw, pt, pl;
} SiPool2dDesc;
/* window max, padding = lowest()   (src/layer/max_pool_2d.cpp:77-121) */
int si_hip_maxpool2d_f32(const SiPool2dDesc* d, const float* in, float* out, si_stream_t stream);
/* uniform-window mean, kernel = in/out   (src/layer/adaptive_avg_pool_2d.cpp:54-116) */
int si_hip_adaptive_avgpool2d_f32(const float* in, int n, int ih, int iw, int c, int in_ld, float* out, int oh,
                                  int ow, int out_ld, si_stream_t stream);
/* nearest: src = clamp((int)((float)dst * (1.0f/scale)))   (src/layer/upsample.cpp:76-99,164-165) */
int si_hip_upsample_nearest_f32(const float* in, int n, int ih, int iw, int c, int in_ld, float scale_h,
                                float scale_w, float* out, int oh, int ow, int out_ld, si_stream_t stream);

/* ---- data movement ----------------------------------------------------- */
/* strided channel-slice copy: out[p*out_ld + i] = in[p*in_ld + i], i < c  -- one slice-assign of
 * Cat::Forward on the channel axis (src/layer/cat.cpp:86-105) */
int si_hip_copy_channels_f32(const float* in, size_t pixels, int c, int in_ld, float* out, int out_ld,
                             si_stream_t stream);
/* generic rank-4 slice copy for cat along any NHWC axis (src/layer/cat.cpp:86-105); dense tensors */
int si_hip_cat_axis_f32(const float* in, const int in_shape[4], float* out, const int out_shape[4], int axis,
                        int offset, si_stream_t stream);
/* NHWC -> NCHW + flatten (src/layer/flatten.cpp:55-88) */
int si_hip_nhwc_to_nchw_f32(const float* in, int n, int h, int w, int c, int in_ld, float* out,
                            si_stream_t stream);

/* ---- elementwise ------------------------------------------------------- */
/* y = act(x) over [pixels, c] with pixel strides */
int si_hip_activation_f32(int act, float act_param, const float* in, size_t pixels, int c, int in_ld, float* out,
                          int out_ld, si_stream_t stream);
/* out = a (op) b with Eigen-style tiling broadcast by integer factors out/in per dim (src/layer/binary_op.cpp:60-75).
 * op: the codes pnnx's expression lowering writes into BinaryOp's param "0" (src/pnnx/expand_expression.cpp:198-216):
 * 0 add, 1 sub, 2 mul, 3 div, 6 pow, 10 atan2, and operand-reversed 7 (b - a), 8 (b / a), 9 (pow(b, a)), 11.  The reference
 * layer implements 0 and 2 only (binary_op.cpp:17-31) and fails LoadModel on the rest; here they all run.
 * Shapes are rank-4 NHWC; *_ld pixel strides. */
int si_hip_binary_f32(int op, const float* a, const int a_shape[4], int a_ld, const float* b,
                      const int b_shape[4], int b_ld, float* out, const int out_shape[4], int out_ld,
                      si_stream_t stream);
/* out = in (op) scalar over [pixels, c]: BinaryOp's `with_scalar` form (params "1" = 1, "2" = the literal;
 * src/pnnx/expand_expression.cpp:206-236), same op codes; 7 / 8 / 9 / 11 put the scalar on the left.  The reference's
 * BinaryOp::Init does not read those params. */
int si_hip_binary_scalar_f32(int op, const float* in, size_t pixels, int c, int in_ld, float scalar, float* out,
                             int out_ld, si_stream_t stream);
/* out = f(in) over [pixels, c]: the UnaryOp operator pnnx's expression lowering emits (src/pnnx/expand_expression.cpp:
 * 123-165) and the reference never registered (LoadModel returns kEmpty, src/engine_impl.cpp:247-250).  op: 0 abs, 1 neg,
 * 2 floor, 3 ceil, 4 square, 5 sqrt, 6 rsqrt, 7 exp, 8 log, 9 sin, 10 cos, 11 tan, 12 asin, 13 acos, 14 atan, 15 reciprocal,
 * 16 tanh, 17 log10. */
int si_hip_unary_f32(int op, const float* in, size_t pixels, int c, int in_ld, float* out, int out_ld,
                     si_stream_t stream);
/* (x-mean)*rsqrt(var+eps)*gamma+beta   (src/layer/batch_norm_2d.cpp:84-137) */
int si_hip_batchnorm2d_f32(const float* in, size_t pixels, int c, int in_ld, const float* mean, const float* var,
                           const float* gamma, const float* beta, float eps, float* out, int out_ld,
                           si_stream_t stream);

/* ---- YOLOv5 Detect decode ---------------------------------------------- */
/* One level of YoloDetect::Forward after its 1x1 conv (src/layer/yolo_detect.cpp:223-266):
 * conv [n][h][w][na*ne] (dense) -> sigmoid -> rows [h][w][a] of out [n][rows_total][ne] at row_off;
 * xy = (2s + grid) * stride; wh = (2s)^2 * anchor.  grid / anchor are already re-laid [h][w][na][2]
 * (the shuffle of :75-79 is done on the host at Init). */
int si_hip_yolo_decode_f32(const float* conv, int n, int h, int w, int na, int ne, const float* grid_hwa2,
                           const float* anchor_hwa2, float stride, float* out, int rows_total, int row_off,
                           si_stream_t stream);

/* ---- the steps either side of Forward() in the reference's application ----------------------- */
/* Letterbox geometry of PreProcess (test/test_yolo/test_yolo.cpp:194-213, 234-241): aspect-preserving resize target,
 * scale, and the top / left padding (Adjust, :18-22).  Host-only arithmetic, no device work. */
void si_letterbox_geometry(int height_origin, int width_origin, int height_new, int width_new, int* height_resize,
                           int* width_resize, float* scale, int* padding_t, int* padding_l);
/* The rest of PreProcess after cv::resize (test_yolo.cpp:220-259): BGR u8 [height_resize][width_resize][3] ->
 * reverse to RGB, pad to [height_new][width_new] with 114, cast to float, divide by 255; written to one image slot
 * `out` of the NHWC input tensor.  (The bilinear resize in front of it: si_hip_resize_bilinear_u8c3 /
 * si_hip_resize_letterbox_batch_u8_f32 below.) */
int si_hip_letterbox_u8_f32(const unsigned char* resized_bgr, int height_resize, int width_resize, float* out,
                            int height_new, int width_new, int padding_t, int padding_l, si_stream_t stream);
/* ... for the n images of a batch that share one geometry (frames of one camera), image b at resized_bgr + b *
 * image_stride_bytes, written to slots 0..n-1 of the NHWC input tensor `out`: one launch */
int si_hip_letterbox_batch_u8_f32(const unsigned char* resized_bgr, int n, size_t image_stride_bytes, int height_resize,
                                  int width_resize, float* out, int height_new, int width_new, int padding_t,
                                  int padding_l, si_stream_t stream);
/* The cv::resize of PreProcess (test_yolo.cpp:213-216) on the device (round 4).  cv::resize comes from the reference's simpleocv
 * submodule, which is ABSENT from the checkout: this implements the published 8-bit INTER_LINEAR algorithm that library shares with
 * OpenCV / ncnn (half-pixel centres, 11-bit fixed-point weights, 16-bit horizontal pass, + 2 >> 2 vertical pass) and is pinned to
 * a numpy restatement of THAT in the test suite, not to the reference.  n images of src_h x src_w x 3 bytes at src + b *
 * src_stride_bytes -> dst_h x dst_w x 3 at dst + b * dst_stride_bytes. */
int si_hip_resize_bilinear_u8c3(const unsigned char* src, int n, size_t src_stride_bytes, int src_h, int src_w, unsigned char* dst,
                                size_t dst_stride_bytes, int dst_h, int dst_w, si_stream_t stream);
/* PreProcess whole for n camera frames that share one size: si_letterbox_geometry, the bilinear resize above (its output is never
 * written), BGR -> RGB, pad(114), float, / 255 -- into slots 0..n-1 of the NHWC fp32 input tensor [n][height_new][width_new][3].
 * Equals si_hip_resize_bilinear_u8c3 followed by si_hip_letterbox_batch_u8_f32, bit for bit. */
int si_hip_resize_letterbox_batch_u8_f32(const unsigned char* frames_bgr, int n, size_t image_stride_bytes, int height_origin,
                                         int width_origin, float* out, int height_new, int width_new, si_stream_t stream);
/* Detection post-processing of test_yolo.cpp:337-428 on the device, for all images of a batch:
 *   confidence = pred[.,4] * max_k pred[.,5+k] (first maximum), kept when >= prob_threshold (:341-377);
 *   sorted by confidence, descending (:380; equal confidences are ordered by element index here, by an unstable
 *   quicksort in the reference); greedy NMS against every box picked so far, same label only unless `agnostic`
 *   (:68-104, IoU > nms_threshold suppresses); then, when `adjust` != NULL, the un-letterbox + clip of :390-416.
 * pred [n][rows][ne] fp32 on the device (the Detect output).  adjust: NULL or device [n][5] floats
 * {padding_l, padding_t, scale, image cols, image rows}.  dets: device [n][max_det][6] floats
 * {x, y, width, height, confidence, label}, in picked order.  counts: device [n] ints = number of boxes picked (the
 * reference has no cap: when counts[b] > max_det only the first max_det were stored).  workspace: device scratch of
 * si_hip_yolo_postprocess_workspace_bytes(n, rows, ne) bytes. */
size_t si_hip_yolo_postprocess_workspace_bytes(int n, int rows, int ne);
int si_hip_yolo_postprocess_f32(const float* pred, int n, int rows, int ne, float prob_threshold, float nms_threshold,
                                int agnostic, const float* adjust, float* dets, int* counts, int max_det,
                                void* workspace, size_t workspace_bytes, si_stream_t stream);

/* ---- fp16 storage path (BASELINE.json configs[3]; the reference is fp32 only, so these have no reference routine --
 * they are the fp32 entry points above with half-precision activations / weights and fp32 accumulation) ------------- */
/* host-side conversions (round to nearest even), used to prepare weights and by the tests */
int si_hip_f32_to_f16_host(const float* src, void* dst, size_t n);
int si_hip_f16_to_f32_host(const void* src, float* dst, size_t n);
/* 0: no fp16 kernel for this shape; 1: implicit GEMM on v_mfma_f32_32x32x16_f16 (needs ic/groups % 32 == 0);
 * 2: stem (ic <= 3; 6x6, 7x7 or 3x3 RGB kernels): fp32 input image, fp16 output, weights packed by
 *    si_hip_conv2d_stem_f16_pack_weight_host (csrc/hip/conv_stem_f16.hip)
 * 3: depthwise (groups == ic == oc, ic % 8 == 0): si_hip_conv2d_depthwise_f16
 * (1 also covers ungrouped convs with ic % 8 == 0: every tap's channels are zero-padded to whole 32-channel blocks; round 5: any kernel
 * size, until then 1x1 only) */
int si_hip_conv2d_f16_supported(const SiConv2dDesc* d);
size_t si_hip_conv2d_f16_weight_elems(const SiConv2dDesc* d);
/* OIHW fp32 -> two fp16 images of the same weights, one behind the other (si_hip_conv2d_f16_weight_elems counts both):
 * [oc][K] rows, K order (c/B, kh, kw, c%B), B = 64 when ic/groups % 64 == 0 else 32; then the MFMA B-operand LANE ORDER
 * [group][oc/32][K/16][lane 0..63][8] (lane l = channel l & 31, k = 8 (l >> 5) .. + 7 of the 16-deep step) that the kernels
 * with weights fetched straight from L2 read with one coalesced 16-byte load per lane (round 4) */
int si_hip_conv2d_f16_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, void* w_packed);
/* Tile variant of the fp16 implicit GEMM: the launch-size policy, or d->plan->f16_tile for one call (tests, sweeps): 0-2 the one-stage kernel
 * (64x64, 128x64, 128x128); 3, 7, 9, 10, 11 the kernels with lane-order weights (128x128 as 2x2 waves, 128x32 as 4x1, 128x128 as 1x4, 64x128 as
 * 1x4, 9 at three waves per SIMD).  Every variant produces the same bits.  An unknown or retired id (4, 5, 6, 8 and >= 12 were measured and
 * removed, profiles/r04_f16_bd_sweep.txt) makes the launch return SI_E_BADARG.
 * _tile_variant: the variant the policy (or the plan's id) picks for this shape; -1 when the shape has no fp16 implicit-GEMM kernel */
int si_hip_conv2d_f16_tile_variant(const SiConv2dDesc* d);
/* name of the instantiation si_hip_conv2d_f16 (form 0) / si_hip_conv2d_upcat_f16 (form 1) launches for this problem, exactly as
 * rocprofv3 prints it minus the namespace (as si_hip_conv2d_kernel_name_form for fp32); "" when there is no fp16 kernel */
const char* si_hip_conv2d_f16_kernel_name(const SiConv2dDesc* d, int form);
/* as si_hip_conv2d_f32; in / residual / out fp16 (strides in elements), bias fp32; out_is_f32 != 0 stores fp32 (graph
 * outputs) */
int si_hip_conv2d_f16(const SiConv2dDesc* d, const void* in, const void* w_packed, const float* bias, const void* residual,
                      void* out, int out_is_f32, si_stream_t stream);
/* depthwise convolution (groups == ic == oc, ic % 8 == 0) with fp16 activations (si_hip_conv2d_f16_supported == 3): w_packed is
 * the FP32 depthwise image of si_hip_conv2d_pack_weight_host ([kh*kw][c]), bias fp32, fp32 tap sums, bias / activation / residual
 * fused, outputs rounded to fp16 once */
int si_hip_conv2d_depthwise_f16(const SiConv2dDesc* d, const void* in, const float* w_packed, const float* bias,
                                const void* residual, void* out, si_stream_t stream);
/* stem of the fp16 path: the reference's first Conv2d (src/layer/conv_2d.cpp:207-283 on a 3-channel image) with the
 * image read as fp32, rounded to fp16 on the way into LDS, contracted on v_mfma_f32_32x32x16_f16, bias / activation in
 * fp32, fp16 activations out.  Weights: OIHW fp32 -> [step][lane half][oc padded][8] fp16 B fragments. */
size_t si_hip_conv2d_stem_f16_weight_elems(const SiConv2dDesc* d);
int si_hip_conv2d_stem_f16_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, void* w_packed);
int si_hip_conv2d_stem_f16(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, void* out,
                           si_stream_t stream);
/* ... the same stem under f32_split (round 6; shapes: si_hip_conv2d_f16_supported == 2, a dense image -- in_ld == ic): fp32 image in, FP32
 * activations out, every product from three fp16 MFMA products on operands split hi + 2^-11 lo (the scheme and the RANGE contract of
 * si_hip_conv2d_split3_f32: d->range_flag is set when an accumulator left the matrix cores non-finite; the pack function returns
 * SI_E_UNSUPPORTED for a weight fp16 cannot hold).  Weights: the B fragments above twice, hi image then lo image. */
size_t si_hip_conv2d_stem_split3_weight_elems(const SiConv2dDesc* d);
int si_hip_conv2d_stem_split3_pack_weight_host(const SiConv2dDesc* d, const float* w_oihw, void* w_packed);
int si_hip_conv2d_stem_split3_f32(const SiConv2dDesc* d, const float* in, const void* w_packed, const float* bias, float* out,
                                  si_stream_t stream);
/* si_hip_conv2d_upcat_f32 with fp16 storage (round 4): `up->src` points at HALF data (cast to the struct's pointer type), up->ld /
 * up->c / up->c0 in elements; c0 and c multiples of the K block (64 when ic % 64 == 0, else 32), up->ld and in_ld multiples of 8.
 * Same index rule, same bits as running si_hip_upsample_nearest on the half tensor, the concat copy and si_hip_conv2d_f16. */
int si_hip_conv2d_upcat_f16(const SiConv2dDesc* d, const void* in, const SiConv2dUpsampledSource* up, const void* w_packed,
                            const float* bias, void* out, int split_oc, void* out2, int out2_ld, si_stream_t stream);
int si_hip_conv2d_upcat_f16_supported(const SiConv2dDesc* d, const SiConv2dUpsampledSource* up);
int si_hip_conv2d_split_f16(const SiConv2dDesc* d, const void* in, const void* w_packed, const float* bias, void* out,
                            int split_oc, void* out2, int out2_ld, si_stream_t stream);
/* fp16 features in, fp32 [n][rows_total][ne] detections out */
int si_hip_conv2d_yolo_f16(const SiConv2dDesc* d, const void* in, const void* w_packed, const float* bias,
                           const SiYoloLevel* level, const float* grid_hwa2, const float* anchor_hwa2, float* detect_out,
                           si_stream_t stream);
/* Round 4: a Detect level over 128 / 256 / 512 channels runs as a tile shape of its own (64 consecutive pixels x all na*ne columns
 * per workgroup, decoded rows staged through LDS and written as one contiguous run; conv_igemm_f16.hip detect_f16_tile_kernel) --
 * same bits as the generic tiles.  d->plan->f16_detect_tile = 0 forces the generic tiles for a call (tests, A/B runs); _tile() (declared below)
 * tells which form si_hip_conv2d_yolo_f16 launches for this problem (1: the Detect tile). */
/* Round 4: a 3x3 stride-2 pad-1 conv over 32 input channels to 32 / 64 output channels (YOLOv5's second conv) runs as a persistent
 * spatial-tile kernel (conv_igemm_f16.hip conv_s2c32_f16_kernel: the input patch of the next tile in flight while this one is
 * computed, weights resident in registers) -- same bits as the generic tiles.  d->plan->f16_s2c32 = 0 forces the generic tiles for a call. */
/* Round 5: 3x3 stride-1 pad-1 layers over 128 / 256 input channels (output channels a multiple of 128) as one-shot row slabs
 * (conv_slab_f16.hip: a workgroup stages the input rows of its output rows once for every channel block, crosses one barrier and
 * runs the whole K loop from LDS with the weights streamed from L2 in lane order); same bits as the generic tiles.  d->plan->f16_slab = 0
 * forces the generic tiles for a call, ->f16_slab_w2 = 0 the one-wave-per-SIMD form of the 128-channel slab kernel. */
/* Round 5: the C3 bottleneck's two convs in ONE launch with fp16 storage -- `pw`: 1x1, stride 1, c -> c (c = 128 / 256), bias, SiLU;
 * `conv`: the 3x3 stride-1 pad-1 conv over its output that si_hip_conv2d_f16 runs as row slabs, SiLU, optional shortcut.  The slab kernel
 * computes the 1x1 conv for the pixels of its input patch straight into LDS (conv_slab_f16.hip): the intermediate tensor is never
 * written, one launch less.  Weights: si_hip_conv2d_f16_pack_weight_host of each conv.  Same bits as the two launches.
 * _supported: 0 no, 1 the pair can run fused, 2 ... and it is the form measured FASTER than two launches (7 pixel blocks over 128
 * channels on two waves per SIMD, on a grid covering the chip): what an engine fuses on.
 * The same entry points take the 64-channel pair (c = 64 on maps of whole 4 x 16-pixel tiles, YOLOv5s' 80x80 level): there the 1x1 is
 * computed into the LDS patch of the persistent 3x3 patch kernel (conv_pw_patch_f16.hip); same bits as the two launches. */
int si_hip_conv2d_pw_slab_f16_supported(const SiConv2dDesc* pw, const SiConv2dDesc* conv);
int si_hip_conv2d_pw_slab_f16(const SiConv2dDesc* pw, const SiConv2dDesc* conv, const void* in, const void* pw_w_packed, const float* pw_bias,
                              const void* w_packed, const float* bias, const void* residual, void* out, si_stream_t stream);
/* Round 6: ... and the C3's CLOSING conv behind its last bottleneck pair in the same launch (the 64-channel pair form, conv_pw_patch_f16.hip).
 * `cv3`: 1x1, stride 1, 128 -> 128, SiLU over torch.cat([y, z], channels) with y the pair's output and `z` ([n][h][w][z_ld] halves, 64 channels,
 * z_ld a multiple of 8) the C3's other branch.  The pair's output tile goes to LDS beside z's pixels and is multiplied there: neither y nor the
 * concat buffer is written -- one launch and two tensor round trips less per C3.  Weights: the three convs' si_hip_conv2d_f16_pack_weight_host
 * images.  Same bits as si_hip_conv2d_pw_slab_f16 into a concat slice followed by si_hip_conv2d_f16 (replaces three Conv2d::Forward calls, an
 * add and a cat of the reference: src/layer/conv_2d.cpp:108-118, binary_op.cpp:52-94, cat.cpp:59-108). */
int si_hip_conv2d_pw_cv3_f16_supported(const SiConv2dDesc* pw, const SiConv2dDesc* conv, const SiConv2dDesc* cv3);
int si_hip_conv2d_pw_cv3_f16(const SiConv2dDesc* pw, const SiConv2dDesc* conv, const SiConv2dDesc* cv3, const void* in, const void* pw_w_packed,
                             const float* pw_bias, const void* w_packed, const float* bias, const void* residual, const void* z, int z_ld,
                             const void* cv3_w_packed, const float* cv3_bias, void* out, si_stream_t stream);
/* Round 4: YOLOv5's first two convs in one persistent kernel (conv_stem_s2c32_f16.hip): `stem` = 6x6 s2 p2, 3 -> 32, SiLU on the
 * dense fp32 image (src/layer/conv_2d.cpp:207-283), `conv` = 3x3 s2 p1, 32 -> 32 / 64, SiLU; the 32-channel intermediate is computed
 * tile by tile into LDS and never written.  Weights: si_hip_conv2d_stem_f16_pack_weight_host(stem) and
 * si_hip_conv2d_f16_pack_weight_host(conv).  Same bits as si_hip_conv2d_stem_f16 followed by si_hip_conv2d_f16. */
int si_hip_conv2d_stem_s2c32_f16_supported(const SiConv2dDesc* stem, const SiConv2dDesc* conv);
int si_hip_conv2d_stem_s2c32_f16(const SiConv2dDesc* stem, const SiConv2dDesc* conv, const float* in, const void* stem_w_packed,
                                 const float* stem_bias, const void* conv_w_packed, const float* conv_bias, void* out,
                                 si_stream_t stream);
/* Round 6: ... and the 1x1 conv behind them in the same launch -- `pw`: 1x1, stride 1, 64 -> 64 over conv's 64 output channels, SiLU (YOLOv5's
 * first C3: cv1 | cv2 as one conv over the concatenated filters, as si_hip_conv2d_split_f16 takes them).  The tile of `conv`'s output is
 * multiplied while it is still in the CU (LDS [pixel][channel], four 16-deep MFMA steps): the 64-channel tensor between the two -- 105 MB at
 * batch 32, written and read back by the two launches this replaces -- never exists.  split_oc = 32: output channels [0, 32) to `out` (stride
 * pw->out_ld), [32, 64) to `out2` (stride out2_ld); split_oc = 0: all 64 to `out`.  Weights: the three convs' own pack functions.  Same bits as
 * si_hip_conv2d_stem_s2c32_f16 followed by si_hip_conv2d_split_f16 / si_hip_conv2d_f16 (replaces three Conv2d::Forward calls of the
 * reference, src/layer/conv_2d.cpp:108-118, 207-283). */
int si_hip_conv2d_stem_s2c32_pw_f16_supported(const SiConv2dDesc* stem, const SiConv2dDesc* conv, const SiConv2dDesc* pw, int split_oc);
int si_hip_conv2d_stem_s2c32_pw_f16(const SiConv2dDesc* stem, const SiConv2dDesc* conv, const SiConv2dDesc* pw, const float* in,
                                    const void* stem_w_packed, const float* stem_bias, const void* conv_w_packed, const float* conv_bias,
                                    const void* pw_w_packed, const float* pw_bias, void* out, int split_oc, void* out2, int out2_ld,
                                    si_stream_t stream);
int si_hip_conv2d_yolo_f16_tile(const SiConv2dDesc* d, const SiYoloLevel* level);
int si_hip_activation_f16(int act, float act_param, const void* in, size_t pixels, int c, int in_ld, void* out, int out_ld,
                          si_stream_t stream);
/* same-shape add (op 0) / mul (op 2) */
/* UnaryOp on fp16 tensors (round 5): si_hip_unary_f32's function on the widened value, rounded once */
int si_hip_unary_f16(int op, const void* in, size_t pixels, int c, int in_ld, void* out, int out_ld, si_stream_t stream);
int si_hip_binary_same_f16(int op, const void* a, int a_ld, const void* b, int b_ld, void* out, int out_ld, size_t pixels,
                           int c, si_stream_t stream);
/* out[b][p][c] = a[b][p][c] (op) s[b][c] (op: 0 add, 2 mul): the squeeze-excite scale -- BinaryOp whose second operand is
 * broadcast over H, W -- with fp16 storage.  c and the strides multiples of 8, 16-byte aligned pointers. */
int si_hip_binary_bcast_f16(int op, const void* a, int a_ld, const void* s, int s_ld, void* out, int out_ld, int n,
                            size_t pixels_per_image, int c, si_stream_t stream);
int si_hip_maxpool2d_f16(const SiPool2dDesc* d, const void* in, void* out, si_stream_t stream);
/* SPPF pool chain: out1 = maxpool5(in), out2 = maxpool5(out1), out3 = maxpool5(out2), all 5x5 stride 1 pad 2 on [n,h,w,c]
 * maps -- three consecutive MaxPool2d::Forward calls of the reference (src/layer/max_pool_2d.cpp:77-121) in one launch
 * that reads `in` once (csrc/hip/pool_chain.hip).  Strides in elements; every tensor 16-byte aligned with c and the
 * strides multiples of 4 (f32) / 8 (f16), else SI_E_UNSUPPORTED (also for maps too large for LDS): run three pools. */
int si_hip_maxpool5_chain3_f32(const float* in, int n, int h, int w, int c, int in_ld, float* out1, int out1_ld, float* out2,
                               int out2_ld, float* out3, int out3_ld, si_stream_t stream);
int si_hip_maxpool5_chain3_f16(const void* in, int n, int h, int w, int c, int in_ld, void* out1, int out1_ld, void* out2,
                               int out2_ld, void* out3, int out3_ld, si_stream_t stream);
int si_hip_adaptive_avgpool2d_f16(const void* in, int n, int ih, int iw, int c, int in_ld, void* out, int oh, int ow,
                                  int out_ld, si_stream_t stream);
int si_hip_convert_f32_f16(const float* in, size_t pixels, int c, int in_ld, void* out, int out_ld, si_stream_t stream);
int si_hip_convert_f16_f32(const void* in, size_t pixels, int c, int in_ld, float* out, int out_ld, si_stream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* SI_HIP_H_ */
