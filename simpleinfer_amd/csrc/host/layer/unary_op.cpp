// unary_op.cpp -- UnaryOp: the elementwise functions pnnx's expression lowering emits (reference
// src/pnnx/expand_expression.cpp:123-165).  No reference layer exists (last row): the semantics are those of
// the C library functions the operator names stand for, and parity is against the oracle's restatement with them.
#include "layer_util.h"
#include "operators.h"
#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(UnaryOp);

Status UnaryOp::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL(CheckParam(op, "0", 2));
    unary_op_type_ = op->params.at("0").i;
    if (unary_op_type_ < 0 || unary_op_type_ > 17) {
        LOG(ERROR) << "unsupport UnaryOp type [" << unary_op_type_ << "]";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

Status UnaryOp::Validate() {
    CHECK_STATUS(Layer::Validate());
    return ValidateShape(1, 1);
}

Status UnaryOp::Forward(const Tensor& input, Tensor& output) {
    return RunOnDevice({&input}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        if (IsHalf(in[0]) != IsHalf(out[0])) return Status::kUnsupport;  // (the engine puts a cast step in front of a graph output)
        size_t pixels = 0, opix = 0;
        int c = 0, oc = 0;
        if (!GetPixelsChannels(in[0], pixels, c) || !GetPixelsChannels(out[0], opix, oc) || pixels != opix || c != oc) return Status::kErrorShape;
        if (IsHalf(in[0]))
            return CheckHip(si_hip_unary_f16(unary_op_type_, in[0].RawData(), pixels, c, in[0].PixelStride(), out[0].RawData(), out[0].PixelStride(),
                                             Stream()),
                            "UnaryOp (fp16 storage)");
        return CheckHip(si_hip_unary_f32(unary_op_type_, in[0].Data<float>(), pixels, c, in[0].PixelStride(), out[0].Data<float>(),
                                         out[0].PixelStride(), Stream()),
                        "UnaryOp");
    });
}

bool UnaryOp::HalfStorageOk(std::string&) const { return true; }   // si_hip_unary_f16 (round 5)

}  // namespace SimpleInfer
