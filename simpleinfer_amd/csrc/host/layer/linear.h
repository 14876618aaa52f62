// layer/linear.h -- nn.Linear: y = x W^T + b, x [N,in], W [out,in] (reference src/layer/linear.cpp:11-43
// Init -- note the `bias` ATTRIBUTE is required even when bias=False, SURVEY Q3 -- and :74-117 Forward).
// Runs as a 1x1 convolution on the MFMA implicit-GEMM kernel.
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class Linear : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Deinit() override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual bool HalfStorageOk(std::string& why) const override;
    virtual const char* KernelName() const override { return "conv_igemm_f32"; }
    virtual double Flops() const override;

    Status PrepareDevice(bool half = false);

public:
    int in_features_  = 0;
    int out_features_ = 0;
    bool use_bias_    = false;
    std::vector<float> weight_;  // [out][in]
    std::vector<float> bias_;

private:
    DeviceBuffer weight_dev_, bias_dev_;
    bool prepared_half_ = false;
    bool device_ready_ = false;
};

}  // namespace SimpleInfer
