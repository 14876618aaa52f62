#include "batch_norm_2d.h"

#include <cstring>

#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(BatchNorm2d);

static bool ReadVec(const pnnx::Operator* op, const char* key, std::vector<float>& dst) {
    if (!CheckAttr(op, key, 1)) return false;
    const pnnx::Attribute& a = op->attrs.at(key);
    if (1 != a.shape.size() || a.data.size() != (size_t)a.shape[0] * sizeof(float)) return false;
    dst.resize(a.shape[0]);
    memcpy(dst.data(), a.data.data(), a.data.size());
    return true;
}

Status BatchNorm2d::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL(CheckParam(op, "eps", 3));
    eps_ = op->params.at("eps").f;
    CHECK_BOOL(CheckParam(op, "num_features", 2));
    num_features_ = op->params.at("num_features").i;
    CHECK_BOOL(CheckParam(op, "affine", 1));
    use_affine_ = op->params.at("affine").b;
    CHECK_BOOL(ReadVec(op, "running_mean", running_mean_));
    CHECK_BOOL(ReadVec(op, "running_var", running_var_));
    CHECK_BOOL(ReadVec(op, "weight", weight_));
    CHECK_BOOL(ReadVec(op, "bias", bias_));
    device_ready_ = false;
    return Status::kSuccess;
}

Status BatchNorm2d::Deinit() {
    params_dev_.Free();
    device_ready_ = false;
    return Status::kSuccess;
}

Status BatchNorm2d::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(1, 1));
    if (Status::kSuccess != ValidateFloat32()) {
        LOG(ERROR) << "BatchNorm2d::Validate fail [unsupport input/output data type]";
        return Status::kUnsupport;
    }
    if (!IsSameShape(input_tensor_nodes_[0]->tensor.Shape(), output_tensor_nodes_[0]->tensor.Shape())) {
        LOG(ERROR) << "BatchNorm2d::Validate fail [error input/output shape]";
        return Status::kErrorShape;
    }
    return Status::kSuccess;
}

Status BatchNorm2d::PrepareDevice() {
    if (device_ready_) return Status::kSuccess;
    const size_t c = running_mean_.size();
    CHECK_BOOL(c > 0 && running_var_.size() == c && weight_.size() == c && bias_.size() == c);
    std::vector<float> all;
    all.reserve(4 * c);
    all.insert(all.end(), running_mean_.begin(), running_mean_.end());
    all.insert(all.end(), running_var_.begin(), running_var_.end());
    all.insert(all.end(), weight_.begin(), weight_.end());
    all.insert(all.end(), bias_.begin(), bias_.end());
    CHECK_STATUS(CheckHip(params_dev_.Upload(all.data(), all.size() * sizeof(float)), "upload batchnorm"));
    device_ready_ = true;
    return Status::kSuccess;
}

Status BatchNorm2d::Forward(const Tensor& input, Tensor& output) {
    return RunOnDevice({&input}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        CHECK_STATUS(PrepareDevice());
        size_t pixels = 0;
        int c = 0;
        if (!GetPixelsChannels(in[0], pixels, c) || (size_t)c != running_mean_.size()) return Status::kErrorShape;
        const float* p = params_dev_.As<float>();
        return CheckHip(si_hip_batchnorm2d_f32(in[0].Data<float>(), pixels, c, in[0].PixelStride(), p, p + c, p + 2 * c,
                                               p + 3 * c, eps_, out[0].Data<float>(), out[0].PixelStride(), Stream()),
                        "BatchNorm2d");
    });
}

bool BatchNorm2d::HalfStorageOk(std::string& why) const {
    for (auto* n : input_tensor_nodes_) if (IsHalf(n->tensor)) { why = "BatchNorm2d has no fp16 kernel"; return false; }
    for (auto* n : output_tensor_nodes_) if (IsHalf(n->tensor)) { why = "BatchNorm2d has no fp16 kernel"; return false; }
    return true;
}

}  // namespace SimpleInfer
