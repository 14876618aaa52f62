// layer/max_pool_2d.h -- nn.MaxPool2d (reference src/layer/max_pool_2d.cpp:11-46, :77-121):
// window max with lowest() padding; ceil_mode / return_indices are parsed and ignored, as there.
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class MaxPool2d : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    // SPPF: this pool and the two that follow it (same 5x5 s1 p2 window, each fed by the previous one) as one launch;
    // outputs[k] is the k-th pool's operand.  Falls back to three launches where the fused kernel does not apply.
    virtual Status Forward(const Tensor& input, std::vector<Tensor>& outputs) override;
    virtual const char* KernelName() const override { return chain_.empty() ? "maxpool" : "maxpool5_chain3"; }
    bool ChainHead(const MaxPool2d& next) const;  // `next` may follow this pool in a fused chain
    void SetChain(MaxPool2d* second, MaxPool2d* third) { chain_ = {second, third}; }

public:
    std::vector<MaxPool2d*> chain_;
    bool ceil_mode_      = false;
    bool return_indices_ = false;
    int padding_t_ = 0, padding_b_ = 0, padding_l_ = 0, padding_r_ = 0;
    int kernel_h_ = 0, kernel_w_ = 0;
    int stride_h_ = 1, stride_w_ = 1;
    int dilation_h_ = 1, dilation_w_ = 1;
};

}  // namespace SimpleInfer
