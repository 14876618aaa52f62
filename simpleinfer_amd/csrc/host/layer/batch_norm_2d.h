// layer/batch_norm_2d.h -- nn.BatchNorm2d inference form (reference src/layer/batch_norm_2d.cpp:11-47
// Init, :84-137 Forward): (x - mean) * rsqrt(var + eps) * weight + bias per channel.
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class BatchNorm2d : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Deinit() override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual bool HalfStorageOk(std::string& why) const override;
    virtual const char* KernelName() const override { return "batchnorm"; }

    Status PrepareDevice();

public:
    float eps_        = 1e-5f;
    int num_features_ = 0;
    bool use_affine_  = true;
    std::vector<float> running_mean_, running_var_, weight_, bias_;

private:
    DeviceBuffer params_dev_;  // [mean | var | weight | bias]
    bool device_ready_ = false;
};

}  // namespace SimpleInfer
