// layer/max_pool_2d.h -- nn.MaxPool2d (reference src/layer/max_pool_2d.cpp:11-46, :77-121):
// window max with lowest() padding; ceil_mode / return_indices are parsed and ignored, as there.
#ifndef SIMPLE_INFER_SRC_LAYER_MAX_POOL_2D_H_
#define SIMPLE_INFER_SRC_LAYER_MAX_POOL_2D_H_

#include "layer.h"

namespace SimpleInfer {

class MaxPool2d : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "maxpool"; }

public:
    bool ceil_mode_      = false;
    bool return_indices_ = false;
    int padding_t_ = 0, padding_b_ = 0, padding_l_ = 0, padding_r_ = 0;
    int kernel_h_ = 0, kernel_w_ = 0;
    int stride_h_ = 1, stride_w_ = 1;
    int dilation_h_ = 1, dilation_w_ = 1;
};

}  // namespace SimpleInfer

#endif
