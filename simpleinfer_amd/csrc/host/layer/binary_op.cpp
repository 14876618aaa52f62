#include "binary_op.h"

#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(BinaryOp);

Status BinaryOp::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL(CheckParam(op, "0", 2));
    const int code = op->params.at("0").i;
    if (code == 0) {
        binary_op_type_ = BinaryOpType::kAdd;
    } else if (code == 2) {
        binary_op_type_ = BinaryOpType::kMul;
    } else {
        LOG(ERROR) << "unsupport BinaryOp type [" << code << "]";
        return Status::kUnsupport;
    }
    if (CheckParam(op, "1", 2) && op->params.at("1").i != 0) {
        // scalar operand form ("1"=with_scalar, "2"=value): the reference layer has no handling for it
        LOG(ERROR) << "unsupport BinaryOp with scalar operand";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

Status BinaryOp::Validate() {
    CHECK_STATUS(Layer::Validate());
    return ValidateShape(2, 1);
}

Status BinaryOp::Forward(const std::vector<Tensor>& inputs, Tensor& output) {
    if (inputs.size() != 2) return Status::kErrorShape;
    return RunOnDevice({&inputs[0], &inputs[1]}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        const std::vector<int> a = in[0].ShapeAs(4), b = in[1].ShapeAs(4), o = out[0].ShapeAs(4);
        for (int i = 0; i < 4; ++i)
            if (a[i] <= 0 || b[i] <= 0 || o[i] % a[i] != 0 || o[i] % b[i] != 0) return Status::kErrorShape;
        if (IsHalf(in[0]) || IsHalf(in[1]) || IsHalf(out[0])) {
            // fp16 path: same-shape add / mul only (what residual blocks need)
            if (!(IsHalf(in[0]) && IsHalf(in[1]) && IsHalf(out[0])) || a != o || b != o) return Status::kUnsupport;
            return CheckHip(si_hip_binary_same_f16((int)binary_op_type_, in[0].RawData(), in[0].PixelStride(), in[1].RawData(),
                                                   in[1].PixelStride(), out[0].RawData(), out[0].PixelStride(),
                                                   (size_t)o[0] * o[1] * o[2], o[3], Stream()),
                            "BinaryOp");
        }
        return CheckHip(si_hip_binary_f32((int)binary_op_type_, in[0].Data<float>(), a.data(), in[0].PixelStride(),
                                          in[1].Data<float>(), b.data(), in[1].PixelStride(), out[0].Data<float>(),
                                          o.data(), out[0].PixelStride(), Stream()),
                        "BinaryOp");
    });
}

// reference src/layer/binary_op.cpp:96-126
Status BroadcastShape(const std::vector<int>& s0, const std::vector<int>& s1, std::vector<int>& out) {
    if (s0.size() != s1.size()) {
        LOG(ERROR) << "BroadcastShape: different shape size [" << s0.size() << "][" << s1.size() << "]";
        return Status::kUnsupport;
    }
    out.resize(s0.size());
    for (size_t i = 0; i < s0.size(); ++i) {
        if (s0[i] == s1[i] || 1 == s1[i]) {
            out[i] = s0[i];
        } else if (1 == s0[i]) {
            out[i] = s1[i];
        } else {
            LOG(ERROR) << "BroadcastShape: different dim size [" << s0[i] << "][" << s1[i] << "] at dimension [" << i << "]";
            return Status::kUnsupport;
        }
    }
    return Status::kSuccess;
}

}  // namespace SimpleInfer
