#include "activation.h"

#include "layer_util.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(ReLU);
DEFINE_LAYER_REGISTRY(SiLU);
DEFINE_LAYER_REGISTRY(Sigmoid);
DEFINE_LAYER_REGISTRY(HardSigmoid);
DEFINE_LAYER_REGISTRY(HardSwish);
DEFINE_LAYER_REGISTRY(LeakyReLU);

Status ActivationLayer::Init(const pnnx::Operator* op) { return Layer::Init(op); }

Status LeakyReLU::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    if (CheckParam(op, "negative_slope", 3)) act_param_ = op->params.at("negative_slope").f;
    return Status::kSuccess;
}

Status ActivationLayer::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(1, 1));
    if (Status::kSuccess != ValidateFloat()) {
        LOG(ERROR) << label_ << "::Validate fail [unsupport input/output data type]";
        return Status::kUnsupport;
    }
    if (!IsSameShape(input_tensor_nodes_[0]->tensor.Shape(), output_tensor_nodes_[0]->tensor.Shape())) {
        LOG(ERROR) << label_ << "::Validate fail [error input/output shape]";
        return Status::kErrorShape;
    }
    return Status::kSuccess;
}

Status ActivationLayer::Forward(const Tensor& input, Tensor& output) {
    return RunOnDevice({&input}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        size_t pixels = 0;
        int c = 0;
        if (!GetPixelsChannels(in[0], pixels, c) || in[0].NumElements() != out[0].NumElements()) return Status::kErrorShape;
        if (IsHalf(in[0]) != IsHalf(out[0])) return Status::kUnsupport;
        if (IsHalf(in[0]))
            return CheckHip(si_hip_activation_f16(act_, act_param_, in[0].RawData(), pixels, c, in[0].PixelStride(),
                                                  out[0].RawData(), out[0].PixelStride(), Stream()),
                            label_);
        return CheckHip(si_hip_activation_f32(act_, act_param_, in[0].Data<float>(), pixels, c, in[0].PixelStride(),
                                              out[0].Data<float>(), out[0].PixelStride(), Stream()),
                        label_);
    });
}

}  // namespace SimpleInfer
