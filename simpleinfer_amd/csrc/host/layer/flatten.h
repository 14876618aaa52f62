// layer/flatten.h -- torch.flatten: rank-4 input is re-ordered NHWC -> NCHW then flattened, other ranks
// are a plain copy; start_dim / end_dim are parsed and ignored (reference src/layer/flatten.cpp:17-21,
// :55-88).
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class Flatten : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "nhwc_to_nchw"; }

public:
    int start_dim_ = 0;
    int end_dim_   = -1;
};

}  // namespace SimpleInfer
