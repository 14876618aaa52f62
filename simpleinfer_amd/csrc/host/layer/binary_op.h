// layer/binary_op.h -- BinaryOp emitted by pnnx::expand_expression: add (code 0) / mul (code 2) with
// broadcast by integer factors (reference src/layer/binary_op.cpp:11-32, :52-94; other codes and the
// scalar form are kUnsupport there and here).
#ifndef SIMPLE_INFER_SRC_LAYER_BINARY_OP_H_
#define SIMPLE_INFER_SRC_LAYER_BINARY_OP_H_

#include "layer.h"

namespace SimpleInfer {

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

}  // namespace SimpleInfer

#endif
