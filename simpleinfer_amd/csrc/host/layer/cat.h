// layer/cat.h -- torch.cat over rank-4 tensors; NCHW dim -> NHWC axis map 1->3, 2->1, 3->2
// (reference src/layer/cat.cpp:59-108).  When the engine aliases the producers into this layer's
// output buffer (zero-copy cat) the corresponding input is skipped here.
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class Cat : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const std::vector<Tensor>& inputs, Tensor& output) override;
    // "aliased (no launch)": every input was already in place at the last Forward() (the engine pointed the producers at their slices of this
    // layer's output) -- a profile line for such a concat times an event pair, not a kernel
    virtual const char* KernelName() const override { return last_copies_ == 0 ? "aliased (no launch)" : "copy_channels"; }

    // NHWC axis the layer concatenates along
    int NhwcAxis() const;

public:
    int dim_ = 0;

private:
    int last_copies_ = -1;   // copy launches of the most recent Forward() (-1: none yet)
};

}  // namespace SimpleInfer
