// layer/output_cast.h -- not a pnnx operator: with fp16 storage the engine appends one after any layer other than
// Conv2d / Linear / Detect (those convert in their own epilogue) that produces a graph output, so Extract() stays fp32; and
// (round 4) puts one on either side of a layer that has no fp16 kernel, which then runs its fp32 kernel on fp32 shadows of its
// half operands instead of making LoadModel refuse the graph (EngineImpl::InsertFp32Fallbacks).  Either direction.
#pragma once

#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class OutputCast : public Layer {
public:
    explicit OutputCast(const std::string& producer, const char* suffix = ".to_f32");
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return to_half_ ? "convert_f32_f16" : "convert_f16_f32"; }
    virtual bool HalfStorageOk(std::string&) const override { return true; }   // fp16 in, fp32 out is what it is for

private:
    pnnx::Operator op_storage_;  // the schedule / profile name of this step
    bool to_half_ = false;       // direction, read off the bound tensors in Validate()
};

}  // namespace SimpleInfer
