#include "linear.h"

#include <cstring>

#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(Linear);

Status Linear::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL(CheckParam(op, "in_features", 2));
    in_features_ = op->params.at("in_features").i;
    CHECK_BOOL(CheckParam(op, "out_features", 2));
    out_features_ = op->params.at("out_features").i;
    CHECK_BOOL(CheckParam(op, "bias", 1));
    use_bias_ = op->params.at("bias").b;

    CHECK_BOOL(CheckAttr(op, "weight", 1));
    const pnnx::Attribute& w = op->attrs.at("weight");
    CHECK_BOOL(2 == w.shape.size() && w.shape[0] == out_features_ && w.shape[1] == in_features_);
    CHECK_BOOL(w.data.size() == (size_t)out_features_ * in_features_ * sizeof(float));
    weight_.resize((size_t)out_features_ * in_features_);
    memcpy(weight_.data(), w.data.data(), w.data.size());

    CHECK_BOOL(CheckAttr(op, "bias", 1));  // required unconditionally, as in the reference
    const pnnx::Attribute& b = op->attrs.at("bias");
    CHECK_BOOL(1 == b.shape.size() && b.data.size() == (size_t)b.shape[0] * sizeof(float));
    bias_.resize(b.shape[0]);
    memcpy(bias_.data(), b.data.data(), b.data.size());
    device_ready_ = false;
    return Status::kSuccess;
}

Status Linear::Deinit() {
    weight_dev_.Free();
    bias_dev_.Free();
    device_ready_ = false;
    return Status::kSuccess;
}

Status Linear::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(1, 1));
    if (Status::kSuccess != ValidateFloat()) {
        LOG(ERROR) << "Linear::Validate fail [unsupport input/output data type]";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

Status Linear::PrepareDevice(bool half) {
    if (device_ready_ && half == prepared_half_) return Status::kSuccess;
    device_ready_ = false;
    prepared_half_ = half;
    CHECK_BOOL(in_features_ > 0 && out_features_ > 0);
    CHECK_BOOL(weight_.size() == (size_t)in_features_ * out_features_);
    SiConv2dDesc d;
    memset(&d, 0, sizeof(d));
    d.ic = in_features_; d.oc = out_features_; d.kh = d.kw = d.sh = d.sw = d.dh = d.dw = 1; d.groups = 1;
    if (half) {
        if (si_hip_conv2d_f16_supported(&d) != 1) {
            LOG(ERROR) << "Linear: no fp16 kernel for in_features " << in_features_ << " (needs a multiple of 8)";
            return Status::kUnsupport;
        }
        std::vector<uint16_t> packed(si_hip_conv2d_f16_weight_elems(&d));
        CHECK_STATUS(CheckHip(si_hip_conv2d_f16_pack_weight_host(&d, weight_.data(), packed.data()), "pack fp16 weight"));
        CHECK_STATUS(CheckHip(weight_dev_.Upload(packed.data(), packed.size() * sizeof(uint16_t)), "upload weight"));
    } else {
        std::vector<float> packed(si_hip_conv2d_weight_elems(&d));
        CHECK_STATUS(CheckHip(si_hip_conv2d_pack_weight_host(&d, weight_.data(), packed.data()), "pack weight"));
        CHECK_STATUS(CheckHip(weight_dev_.Upload(packed.data(), packed.size() * sizeof(float)), "upload weight"));
    }
    if (use_bias_) {
        CHECK_BOOL(bias_.size() == (size_t)out_features_);
        CHECK_STATUS(CheckHip(bias_dev_.Upload(bias_.data(), bias_.size() * sizeof(float)), "upload bias"));
    }
    device_ready_ = true;
    return Status::kSuccess;
}

Status Linear::Forward(const Tensor& input, Tensor& output) {
    return RunOnDevice({&input}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        const bool half = IsHalf(in[0]);
        CHECK_STATUS(PrepareDevice(half));
        const std::vector<int> is = in[0].ShapeAs(2), os = out[0].ShapeAs(2);
        if (is[1] != in_features_ || os[1] != out_features_ || is[0] != os[0]) return Status::kErrorShape;
        SiConv2dDesc d;
        memset(&d, 0, sizeof(d));
        d.n = is[0]; d.ih = d.iw = 1; d.ic = in_features_; d.in_ld = in[0].PixelStride();
        d.oh = d.ow = 1; d.oc = out_features_; d.out_ld = out[0].PixelStride();
        d.kh = d.kw = d.sh = d.sw = d.dh = d.dw = 1; d.groups = 1;
        d.has_bias = use_bias_ ? 1 : 0;
        if (half)  // fp16 features; the result is stored in the output tensor's own precision (fp32 for a graph output)
            return CheckHip(si_hip_conv2d_f16(&d, in[0].RawData(), weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                              nullptr, out[0].RawData(), IsHalf(out[0]) ? 0 : 1, Stream()),
                            "Linear fp16");
        if (IsHalf(out[0])) return Status::kUnsupport;
        return CheckHip(si_hip_conv2d_f32(&d, in[0].Data<float>(), weight_dev_.As<float>(),
                                          use_bias_ ? bias_dev_.As<float>() : nullptr, nullptr, out[0].Data<float>(),
                                          Stream()),
                        "Linear");
    });
}

double Linear::Flops() const {
    if (output_tensor_nodes_.empty()) return 0.0;
    return 2.0 * (double)output_tensor_nodes_[0]->tensor.NumElements() * in_features_;
}

bool Linear::HalfStorageOk(std::string& why) const {
    const bool half_in = !input_tensor_nodes_.empty() && IsHalf(input_tensor_nodes_[0]->tensor);
    const bool half_out = !output_tensor_nodes_.empty() && IsHalf(output_tensor_nodes_[0]->tensor);
    if (half_out && !half_in) { why = "Linear writes fp16 only from fp16 features"; return false; }
    if (half_in && in_features_ % 8 != 0) {
        why = "Linear's fp16 kernel needs in_features % 8 == 0";
        return false;
    }
    return true;
}

}  // namespace SimpleInfer
