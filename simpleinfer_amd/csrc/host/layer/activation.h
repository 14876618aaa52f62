// layer/activation.h -- elementwise activations: ReLU, SiLU, Sigmoid, HardSigmoid, HardSwish
// (reference src/layer/{relu,silu,sigmoid,hard_sigmoid,hard_swish}.cpp) and LeakyReLU (north_star
// extension, no reference layer).  One HBM-bound kernel family (si_hip_activation_f32); inside a
// loaded graph the engine usually folds these into the producing conv's epilogue instead.
#ifndef SIMPLE_INFER_SRC_LAYER_ACTIVATION_H_
#define SIMPLE_INFER_SRC_LAYER_ACTIVATION_H_

#include "layer.h"
#include "si_hip.h"

namespace SimpleInfer {

class ActivationLayer : public Layer {
public:
    explicit ActivationLayer(int act, const char* label) : act_(act), label_(label) {}

    virtual Status Init(const pnnx::Operator* op) override;
    virtual Status Validate() override;
    virtual Status Forward(const Tensor& input, Tensor& output) override;
    virtual const char* KernelName() const override { return "activation"; }

    int ActCode() const { return act_; }
    float ActParam() const { return act_param_; }

public:
    float act_param_ = 0.0f;  // LeakyReLU negative_slope

protected:
    int act_;
    const char* label_;
};

class ReLU : public ActivationLayer {
public:
    ReLU() : ActivationLayer(SI_ACT_RELU, "ReLU") {}
};

class SiLU : public ActivationLayer {
public:
    SiLU() : ActivationLayer(SI_ACT_SILU, "SiLU") {}
};

class Sigmoid : public ActivationLayer {
public:
    Sigmoid() : ActivationLayer(SI_ACT_SIGMOID, "Sigmoid") {}
};

// alpha = 1/6, beta = 0.5 hard-coded as in reference src/layer/hard_sigmoid.cpp:17-19
class HardSigmoid : public ActivationLayer {
public:
    HardSigmoid() : ActivationLayer(SI_ACT_HARDSIGMOID, "HardSigmoid") {}
};

class HardSwish : public ActivationLayer {
public:
    HardSwish() : ActivationLayer(SI_ACT_HARDSWISH, "HardSwish") {}
};

class LeakyReLU : public ActivationLayer {
public:
    LeakyReLU() : ActivationLayer(SI_ACT_LEAKYRELU, "LeakyReLU") { act_param_ = 0.01f; }
    virtual Status Init(const pnnx::Operator* op) override;
};

}  // namespace SimpleInfer

#endif
