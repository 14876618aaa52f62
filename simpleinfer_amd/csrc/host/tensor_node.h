// tensor_node.h -- graph operand + the tensor bound to it (reference src/tensor_node.h:9-12)
#pragma once

#include "pnnx/ir.h"
#include "tensor.h"

namespace SimpleInfer {

struct TensorNode {
    pnnx::Operand* operand = nullptr;
    Tensor tensor;
};

}  // namespace SimpleInfer
