// layer/operators.h -- the small operators of the path in one place: pooling, batch-norm, add / mul, concat, flatten,
// linear, nearest upsample.  Each class keeps the reference layer's name, public fields and virtuals (src/layer/<name>.h),
// because the reference's own tests construct layers directly and poke those fields; what differs
// is underneath: Forward() enqueues one HIP kernel from include/si_hip.h on the context's stream, in fp32 or -- when the
// engine runs with fp16 storage -- in fp16.  The per-layer headers (layer/cat.h, ...) forward here so that plugin code
// written against the reference's include paths keeps compiling.  Conv2d and YoloDetect have their own headers.
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

// ----------------------------------------------------------------------------------------------------
// layer/adaptive_avg_pool_2d.h -- nn.AdaptiveAvgPool2d: global mean for 1x1, else uniform windows
// k = in/out with divisibility required (reference src/layer/adaptive_avg_pool_2d.cpp:54-116).
class AdaptiveAvgPool2d : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "avgpool"; }

public:
    int output_h_ = 0;
    int output_w_ = 0;
};

// ----------------------------------------------------------------------------------------------------
// layer/batch_norm_2d.h -- nn.BatchNorm2d inference form (reference src/layer/batch_norm_2d.cpp:11-47
// Init, :84-137 Forward): (x - mean) * rsqrt(var + eps) * weight + bias per channel.
class BatchNorm2d : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Deinit() override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "batchnorm"; }

    Status PrepareDevice();

public:
    float eps_        = 1e-5f;
    int num_features_ = 0;
    bool use_affine_  = true;
    std::vector<float> running_mean_, running_var_, weight_, bias_;

private:
    DeviceBuffer params_dev_;  // [mean | var | weight | bias]
    bool device_ready_ = false;
};

// ----------------------------------------------------------------------------------------------------
// layer/binary_op.h -- BinaryOp emitted by pnnx::expand_expression: add (code 0) / mul (code 2) with
// broadcast by integer factors (reference src/layer/binary_op.cpp:11-32, :52-94; other codes and the
// scalar form are kUnsupport there and here).
class BinaryOp : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const std::vector<Tensor>& inputs, Tensor& output) override;
    virtual const char* KernelName() const override { return "binary"; }

public:
    enum class BinaryOpType { kAdd = 0, kMul = 2 } binary_op_type_ = BinaryOpType::kAdd;
};

Status BroadcastShape(const std::vector<int>& shape0, const std::vector<int>& shape1, std::vector<int>& broadcast_shape);

// ----------------------------------------------------------------------------------------------------
// layer/cat.h -- torch.cat over rank-4 tensors; NCHW dim -> NHWC axis map 1->3, 2->1, 3->2
// (reference src/layer/cat.cpp:59-108).  When the engine aliases the producers into this layer's
// output buffer (zero-copy cat) the corresponding input is skipped here.
class Cat : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const std::vector<Tensor>& inputs, Tensor& output) override;
    virtual const char* KernelName() const override { return "copy_channels"; }

    // NHWC axis the layer concatenates along
    int NhwcAxis() const;

public:
    int dim_ = 0;
};

// ----------------------------------------------------------------------------------------------------
// layer/flatten.h -- torch.flatten: rank-4 input is re-ordered NHWC -> NCHW then flattened, other ranks
// are a plain copy; start_dim / end_dim are parsed and ignored (reference src/layer/flatten.cpp:17-21,
// :55-88).
class Flatten : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "nhwc_to_nchw"; }

public:
    int start_dim_ = 0;
    int end_dim_   = -1;
};

// ----------------------------------------------------------------------------------------------------
// layer/linear.h -- nn.Linear: y = x W^T + b, x [N,in], W [out,in] (reference src/layer/linear.cpp:11-43
// Init -- note the `bias` ATTRIBUTE is required even when bias=False, SURVEY Q3 -- and :74-117 Forward).
// Runs as a 1x1 convolution on the MFMA implicit-GEMM kernel.
class Linear : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Deinit() override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "conv_igemm_f32"; }
    virtual double Flops() const override;

    Status PrepareDevice(bool half = false);

public:
    int in_features_  = 0;
    int out_features_ = 0;
    bool use_bias_    = false;
    std::vector<float> weight_;  // [out][in]
    std::vector<float> bias_;

private:
    DeviceBuffer weight_dev_, bias_dev_;
    bool prepared_half_ = false;
    bool device_ready_ = false;
};

// ----------------------------------------------------------------------------------------------------
// layer/max_pool_2d.h -- nn.MaxPool2d (reference src/layer/max_pool_2d.cpp:11-46, :77-121):
// window max with lowest() padding; ceil_mode / return_indices are parsed and ignored, as there.
class MaxPool2d : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    // SPPF: this pool and the two that follow it (same 5x5 s1 p2 window, each fed by the previous one) as one launch;
    // outputs[k] is the k-th pool's operand.  Falls back to three launches where the fused kernel does not apply.
    virtual Status Forward(const Tensor& input, std::vector<Tensor>& outputs) override;
    virtual const char* KernelName() const override { return chain_.empty() ? "maxpool" : "maxpool5_chain3"; }
    bool ChainHead(const MaxPool2d& next) const;  // `next` may follow this pool in a fused chain
    void SetChain(MaxPool2d* second, MaxPool2d* third) { chain_ = {second, third}; }

public:
    std::vector<MaxPool2d*> chain_;
    bool ceil_mode_      = false;
    bool return_indices_ = false;
    int padding_t_ = 0, padding_b_ = 0, padding_l_ = 0, padding_r_ = 0;
    int kernel_h_ = 0, kernel_w_ = 0;
    int stride_h_ = 1, stride_w_ = 1;
    int dilation_h_ = 1, dilation_w_ = 1;
};

// ----------------------------------------------------------------------------------------------------
// layer/upsample.h -- nn.Upsample, nearest only, scale_factor only (reference src/layer/upsample.cpp:18-45
// Init, :76-99 index rule src = clamp(int(float(dst) * (1/scale)))); `size=` stays unsupported as there.
class Upsample : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "upsample_nearest"; }

public:
    enum class UpsampleMode { kNearest = 0 } upsample_mode_ = UpsampleMode::kNearest;
    float scale_factor_h_ = 1.0f;
    float scale_factor_w_ = 1.0f;
};

// ----------------------------------------------------------------------------------------------------
// layer/output_cast.cpp -- not a pnnx operator: with fp16 storage the engine appends one after any layer other than
// Conv2d / Linear / Detect (those convert in their own epilogue) that produces a graph output, so Extract() stays fp32.
class OutputCast : public Layer {
public:
    explicit OutputCast(const std::string& producer);
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "convert_f16_f32"; }

private:
    pnnx::Operator op_storage_;  // the schedule / profile name of this step
};

}  // namespace SimpleInfer
