// layer/upsample.h -- nn.Upsample, nearest only, scale_factor only (reference src/layer/upsample.cpp:18-45
// Init, :76-99 index rule src = clamp(int(float(dst) * (1/scale)))); `size=` stays unsupported as there.
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class Upsample : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "upsample_nearest"; }

public:
    enum class UpsampleMode { kNearest = 0 } upsample_mode_ = UpsampleMode::kNearest;
    float scale_factor_h_ = 1.0f;
    float scale_factor_w_ = 1.0f;
};

}  // namespace SimpleInfer
