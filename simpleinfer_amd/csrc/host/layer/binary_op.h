// layer/binary_op.h -- BinaryOp emitted by pnnx::expand_expression, with broadcast by integer factors (reference
// src/layer/binary_op.cpp:11-32, :52-94).  The reference layer has add (code 0) and mul (code 2) only and no scalar form;
// here every code the lowering can emit runs.
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class BinaryOp : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const std::vector<Tensor>& inputs, Tensor& output) override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;  // the `with_scalar` form: one tensor operand
    virtual bool HalfStorageOk(std::string& why) const override;
    virtual const char* KernelName() const override { return with_scalar_ ? "binary_scalar" : "binary"; }

public:
    // param "0" as pnnx's expression lowering writes it (reference src/pnnx/expand_expression.cpp:198-216).  The reference layer
    // accepts kAdd and kMul only (src/layer/binary_op.cpp:17-31); the rest is what `x - y`, `x / 2.0`, `2.0 - x`, `x ** 0.5` in an
    // exported model lower to.
    enum class BinaryOpType { kAdd = 0, kSub = 1, kMul = 2, kDiv = 3, kPow = 6, kRSub = 7, kRDiv = 8, kRPow = 9, kAtan2 = 10, kRAtan2 = 11 }
        binary_op_type_ = BinaryOpType::kAdd;
    bool with_scalar_ = false;   // params "1" (expand_expression.cpp:206-236): the other operand is the literal in "2"
    float scalar_ = 0.0f;
};

// broadcast result shape of two same-rank shapes (reference src/layer/binary_op.cpp:96-126)
Status BroadcastShape(const std::vector<int>& shape0, const std::vector<int>& shape1, std::vector<int>& broadcast_shape);

}  // namespace SimpleInfer
