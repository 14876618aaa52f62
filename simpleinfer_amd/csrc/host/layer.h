// layer.h -- the operator plugin base class.  Virtual surface and call order are the reference's
// (src/layer.h:15-89; engine call order creator -> Init(op) -> SetContext -> SetInputNodes ->
// SetOutputNodes -> Validate, src/engine_impl.cpp:252-306): a layer written against SimpleInfer's
// Layer compiles against this one, minus GetEigenThreadPoolDevice (there is no CPU device here).
//
// What is new: tensors handed to Forward() normally live in HBM and the layer enqueues HIP kernels
// on context_->stream().  When a caller passes HOST tensors (the reference's layer tests do:
// test/test_layer/*.cpp construct a layer, poke its public fields and call Forward(in, out)),
// RunOnDevice() stages them through temporary device buffers and synchronises, so the same test
// code still exercises the HIP kernels -- there is no CPU compute path.
#ifndef SIMPLE_INFER_SRC_LAYER_H_
#define SIMPLE_INFER_SRC_LAYER_H_

#include <functional>
#include <map>
#include <string>
#include <vector>

#include "context.h"
#include "layer_registry.h"
#include "logger.h"
#include "pnnx/ir.h"
#include "pnnx/pnnx_helper.h"
#include "tensor.h"
#include "tensor_node.h"
#include "types.h"

namespace SimpleInfer {

class Layer {
public:
    Layer();
    virtual ~Layer();

    // ---- the reference's virtual surface (src/layer.h:15-89), in the order the engine calls it --------------------
    virtual Status Init(const pnnx::Operator* op);                                   // 1. parse params / attrs of one operator
    virtual Status Init(const std::map<std::string, pnnx::Parameter>& params,       //    (the form layers usually override)
                        const std::map<std::string, pnnx::Attribute>& attrs);
    virtual void SetContext(Context* context);                                       // 2. device + stream
    virtual void SetInputNodes(const std::vector<TensorNode*>& input_tensor_nodes);  // 3.
    virtual void SetOutputNodes(const std::vector<TensorNode*>& output_tensor_nodes);  // 4.
    virtual Status Validate();                                                       // 5. arity / dtype / shape checks
    virtual Status Deinit();                                                         // before the registry destroyer

    virtual Status Forward();  // arity dispatch on the bound nodes (src/layer.cpp:45-79) into one of:
    virtual Status Forward(const Tensor& input, Tensor& output);
    virtual Status Forward(const std::vector<Tensor>& inputs, Tensor& output);
    virtual Status Forward(const Tensor& input, std::vector<Tensor>& outputs);
    virtual Status Forward(const std::vector<Tensor>& inputs, std::vector<Tensor>& outputs);

    const pnnx::Operator* GetOp();

public:
    // ---- extensions used by the engine ---------------------------------------------------
    // kernel family name for profiles ("conv_igemm", "maxpool", ...)
    virtual const char* KernelName() const { return "none"; }
    // algorithmic cost of one Forward with the currently bound nodes
    virtual double Flops() const { return 0.0; }
    // tensors a layer reads besides its bound input nodes (fusion hooks: a conv's residual, its upsampled source); the engine's
    // memory planner needs every reader of a buffer
    virtual void ExtraReads(std::vector<TensorNode*>& nodes) const { (void)nodes; }
    // rebind one of the ExtraReads nodes (the engine's fp32 fallback hands a layer fp32 shadows of its half operands); false: cannot
    virtual bool ReplaceExtraRead(TensorNode* from, TensorNode* to) { (void)from; (void)to; return false; }
    // the stream this layer's launches go to (its context's)
    si_stream_t LaunchStream() const { return Stream(); }
    // fp16 storage (Engine option "fp16"): can this layer run with the storage types its bound nodes now have?  Asked once, at
    // LoadModel, so that an unsupported combination is a load-time Status with a reason instead of a failing first Forward().
    // Default: fine when no bound tensor is fp16, or when inputs and outputs are all fp16.
    virtual bool HalfStorageOk(std::string& why) const;
    virtual double Bytes() const;

    const std::vector<TensorNode*>& InputNodes() const { return input_tensor_nodes_; }
    const std::vector<TensorNode*>& OutputNodes() const { return output_tensor_nodes_; }

protected:
    virtual Status ValidateShape(const int input_size, const int output_size);

    // all-float check shared by the concrete Validate()s (reference e.g. src/layer/conv_2d.cpp:94-101)
    Status ValidateFloat32();
    // fp32 or fp16 storage (the fp16 engine path, SetOption("fp16", 1)); inputs / outputs may differ at the graph boundary
    Status ValidateFloat();
    static bool IsHalf(const Tensor& t) { return t.GetDataType() == DataType::kFloat16; }

    si_stream_t Stream() const;

    using DeviceFn = std::function<Status(const std::vector<Tensor>&, std::vector<Tensor>&)>;
    // Calls fn with device-resident views of inputs/outputs; host tensors are staged (H2D, run,
    // D2H, sync).  Device tensors pass straight through with no copy and no sync.
    Status RunOnDevice(const std::vector<const Tensor*>& inputs, const std::vector<Tensor*>& outputs,
                       const DeviceFn& fn);

    // maps a C-ABI return code to Status, logging the HIP error string
    Status CheckHip(int rc, const char* what) const;

protected:
    Context* context_ = nullptr;

    const pnnx::Operator* op_ = nullptr;

    std::vector<TensorNode*> input_tensor_nodes_;
    std::vector<TensorNode*> output_tensor_nodes_;
};

// default layer registry entry (same macro names as reference src/layer.h:74-87)
#define DEFINE_LAYER_CREATOR(type) \
    Layer* type##_LayerCreator() { return (new type); }

#define DEFINE_LAYER_DESTROYER(type)           \
    void type##_LayerDestroyer(Layer* layer) { \
        if (nullptr != layer) { delete layer; } \
    }

#define DEFINE_LAYER_REGISTRY(type) \
    DEFINE_LAYER_CREATOR(type)      \
    DEFINE_LAYER_DESTROYER(type)

}  // namespace SimpleInfer

#endif  // SIMPLE_INFER_SRC_LAYER_H_
