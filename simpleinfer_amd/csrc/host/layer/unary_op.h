// layer/unary_op.h -- UnaryOp: emitted by pnnx's expression lowering for abs / neg / sqrt / exp / ... (reference
// src/pnnx/expand_expression.cpp:123-165, param "0" = ncnn's operator code 0..17).  The reference never registered a layer for
// it (LoadModel fails with kEmpty, src/engine_impl.cpp:247-250); SURVEY.md section 8(f3) asks for it.
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class UnaryOp : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual bool HalfStorageOk(std::string& why) const override;
    virtual const char* KernelName() const override { return "unary"; }

public:
    int unary_op_type_ = 0;  // 0 abs 1 neg 2 floor 3 ceil 4 square 5 sqrt 6 rsqrt 7 exp 8 log 9 sin 10 cos 11 tan 12 asin 13 acos
                             // 14 atan 15 reciprocal 16 tanh 17 log10
};

}  // namespace SimpleInfer
