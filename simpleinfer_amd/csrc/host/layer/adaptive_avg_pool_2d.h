// layer/adaptive_avg_pool_2d.h -- nn.AdaptiveAvgPool2d: global mean for 1x1, else uniform windows
// k = in/out with divisibility required (reference src/layer/adaptive_avg_pool_2d.cpp:54-116).
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class AdaptiveAvgPool2d : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "avgpool"; }

public:
    int output_h_ = 0;
    int output_w_ = 0;
};

}  // namespace SimpleInfer
