#include "binary_op.h"

#include "layer_util.h"

#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(BinaryOp);

Status BinaryOp::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL(CheckParam(op, "0", 2));
    const int code = op->params.at("0").i;
    switch (code) {
        case 0: case 1: case 2: case 3: case 6: case 7: case 8: case 9: case 10: case 11:
            binary_op_type_ = static_cast<BinaryOpType>(code);
            break;
        default:  // 4 / 5 (max / min) are never emitted by expand_expression
            LOG(ERROR) << "unsupport BinaryOp type [" << code << "]";
            return Status::kUnsupport;
    }
    with_scalar_ = false;
    if (CheckParam(op, "1", 2) && op->params.at("1").i != 0) {
        // scalar operand form (expand_expression.cpp:206-236): "1" = with_scalar, "2" = the literal as a float
        CHECK_BOOL(CheckParam(op, "2", 3));
        with_scalar_ = true;
        scalar_ = op->params.at("2").f;
    }
    return Status::kSuccess;
}

Status BinaryOp::Validate() {
    CHECK_STATUS(Layer::Validate());
    return ValidateShape(with_scalar_ ? 1 : 2, 1);
}

Status BinaryOp::Forward(const Tensor& input, Tensor& output) {
    if (!with_scalar_) return Status::kErrorShape;
    return RunOnDevice({&input}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        if (IsHalf(in[0]) || IsHalf(out[0])) return Status::kUnsupport;  // fp16 storage: not built for this operator
        size_t pixels = 0, opix = 0;
        int c = 0, oc = 0;
        if (!GetPixelsChannels(in[0], pixels, c) || !GetPixelsChannels(out[0], opix, oc) || pixels != opix || c != oc) return Status::kErrorShape;
        return CheckHip(si_hip_binary_scalar_f32((int)binary_op_type_, in[0].Data<float>(), pixels, c, in[0].PixelStride(), scalar_,
                                                 out[0].Data<float>(), out[0].PixelStride(), Stream()),
                        "BinaryOp (scalar)");
    });
}

Status BinaryOp::Forward(const std::vector<Tensor>& inputs, Tensor& output) {
    if (inputs.size() != 2) return Status::kErrorShape;
    return RunOnDevice({&inputs[0], &inputs[1]}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        const std::vector<int> a = in[0].ShapeAs(4), b = in[1].ShapeAs(4), o = out[0].ShapeAs(4);
        for (int i = 0; i < 4; ++i)
            if (a[i] <= 0 || b[i] <= 0 || o[i] % a[i] != 0 || o[i] % b[i] != 0) return Status::kErrorShape;
        if (IsHalf(in[0]) || IsHalf(in[1]) || IsHalf(out[0])) {
            // fp16 path: add / mul of same-shape tensors (residual blocks) or with one operand broadcast over H and W (the
            // squeeze-excite scale [N,1,1,C] x [N,H,W,C])
            if (!(IsHalf(in[0]) && IsHalf(in[1]) && IsHalf(out[0]))) return Status::kUnsupport;
            if (binary_op_type_ != BinaryOpType::kAdd && binary_op_type_ != BinaryOpType::kMul) return Status::kUnsupport;
            auto per_image_vector = [&](const std::vector<int>& s) { return s[0] == o[0] && s[1] == 1 && s[2] == 1 && s[3] == o[3]; };
            if (a == o && per_image_vector(b) && b != o)
                return CheckHip(si_hip_binary_bcast_f16((int)binary_op_type_, in[0].RawData(), in[0].PixelStride(), in[1].RawData(), in[1].PixelStride(),
                                                        out[0].RawData(), out[0].PixelStride(), o[0], (size_t)o[1] * o[2], o[3], Stream()),
                                "BinaryOp (broadcast)");
            if (b == o && per_image_vector(a) && a != o)   // add and mul commute
                return CheckHip(si_hip_binary_bcast_f16((int)binary_op_type_, in[1].RawData(), in[1].PixelStride(), in[0].RawData(), in[0].PixelStride(),
                                                        out[0].RawData(), out[0].PixelStride(), o[0], (size_t)o[1] * o[2], o[3], Stream()),
                                "BinaryOp (broadcast)");
            if (a != o || b != o) return Status::kUnsupport;
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

bool BinaryOp::HalfStorageOk(std::string& why) const {
    bool any = false, all = true;
    for (auto* n : input_tensor_nodes_) { any = any || IsHalf(n->tensor); all = all && IsHalf(n->tensor); }
    for (auto* n : output_tensor_nodes_) { any = any || IsHalf(n->tensor); all = all && IsHalf(n->tensor); }
    if (!any) return true;
    const bool addmul = binary_op_type_ == BinaryOpType::kAdd || binary_op_type_ == BinaryOpType::kMul;
    bool same = !with_scalar_ && input_tensor_nodes_.size() == 2 && output_tensor_nodes_.size() == 1;
    if (same) {
        const std::vector<int>& o = output_tensor_nodes_[0]->tensor.Shape();
        same = input_tensor_nodes_[0]->tensor.Shape() == o && input_tensor_nodes_[1]->tensor.Shape() == o;
    }
    if (all && addmul && same) return true;
    if (all && addmul && !with_scalar_ && input_tensor_nodes_.size() == 2 && output_tensor_nodes_.size() == 1) {
        // one operand a per-image channel vector [N,1,1,C] (squeeze-excite), channels a multiple of 8
        const std::vector<int> a = input_tensor_nodes_[0]->tensor.ShapeAs(4), b = input_tensor_nodes_[1]->tensor.ShapeAs(4);
        const std::vector<int> o = output_tensor_nodes_[0]->tensor.ShapeAs(4);
        auto vec = [&](const std::vector<int>& s) { return s[0] == o[0] && s[1] == 1 && s[2] == 1 && s[3] == o[3]; };
        if (o[3] % 8 == 0 && ((a == o && vec(b)) || (b == o && vec(a)))) return true;
    }
    why = "BinaryOp has fp16 kernels for add / mul of same-shape tensors or with a per-image channel vector only (no scalar form, no sub / div / pow)";
    return false;
}

}  // namespace SimpleInfer
