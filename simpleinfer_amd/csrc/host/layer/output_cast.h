// layer/output_cast.h -- not a pnnx operator: with fp16 storage the engine appends one after any layer other than
// Conv2d / Linear / Detect (those convert in their own epilogue) that produces a graph output, so Extract() stays fp32.
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class OutputCast : public Layer {
public:
    explicit OutputCast(const std::string& producer);
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "convert_f16_f32"; }
    virtual bool HalfStorageOk(std::string&) const override { return true; }   // fp16 in, fp32 out is what it is for

private:
    pnnx::Operator op_storage_;  // the schedule / profile name of this step
};

}  // namespace SimpleInfer
