#include "layer.h"

namespace SimpleInfer {

Layer::Layer() {}

Layer::~Layer() {}

Status Layer::Init(const pnnx::Operator* op) {
    if (nullptr == op) return Status::kEmpty;
    op_ = op;
    return Status::kSuccess;
}

Status Layer::Init(const std::map<std::string, pnnx::Parameter>&, const std::map<std::string, pnnx::Attribute>&) {
    return Status::kUnsupport;
}

void Layer::SetContext(Context* context) { context_ = context; }

void Layer::SetInputNodes(const std::vector<TensorNode*>& nodes) { input_tensor_nodes_ = nodes; }

void Layer::SetOutputNodes(const std::vector<TensorNode*>& nodes) { output_tensor_nodes_ = nodes; }

Status Layer::Deinit() { return Status::kSuccess; }

Status Layer::Validate() { return Status::kSuccess; }

// arity dispatch over the bound nodes (reference src/layer.cpp:45-79); tensors are passed as
// non-owning aliases, exactly as the reference's copies are
Status Layer::Forward() {
    LOG(INFO) << "Forward Layer [" << (op_ ? op_->name : std::string("?")) << "]";

    const size_t n_in = input_tensor_nodes_.size(), n_out = output_tensor_nodes_.size();
    if (n_in == 1 && n_out == 1) return Forward(input_tensor_nodes_[0]->tensor, output_tensor_nodes_[0]->tensor);

    std::vector<Tensor> outputs;
    if (n_out != 1) {
        outputs.resize(n_out);
        for (size_t i = 0; i < n_out; ++i) outputs[i] = output_tensor_nodes_[i]->tensor;
    }
    if (n_in == 1) return Forward(input_tensor_nodes_[0]->tensor, outputs);

    std::vector<Tensor> inputs(n_in);
    for (size_t i = 0; i < n_in; ++i) inputs[i] = input_tensor_nodes_[i]->tensor;
    if (n_out == 1) return Forward(inputs, output_tensor_nodes_[0]->tensor);
    return Forward(inputs, outputs);
}

Status Layer::Forward(const Tensor&, Tensor&) { return Status::kUnsupport; }
Status Layer::Forward(const std::vector<Tensor>&, Tensor&) { return Status::kUnsupport; }
Status Layer::Forward(const Tensor&, std::vector<Tensor>&) { return Status::kUnsupport; }
Status Layer::Forward(const std::vector<Tensor>&, std::vector<Tensor>&) { return Status::kUnsupport; }

const pnnx::Operator* Layer::GetOp() { return op_; }

bool Layer::HalfStorageOk(std::string& why) const {
    int half = 0, total = 0;
    for (auto* n : input_tensor_nodes_) { half += IsHalf(n->tensor); ++total; }
    for (auto* n : output_tensor_nodes_) { half += IsHalf(n->tensor); ++total; }
    if (half == 0 || half == total) return true;
    why = "mixed fp16 / fp32 operands";
    return false;
}

double Layer::Bytes() const {
    double b = 0.0;
    for (auto* n : input_tensor_nodes_) b += (double)n->tensor.ByteSize();
    for (auto* n : output_tensor_nodes_) b += (double)n->tensor.ByteSize();
    return b;
}

Status Layer::ValidateShape(const int input_size, const int output_size) {
    if (input_size >= 0 && input_size != (int)input_tensor_nodes_.size()) {
        LOG(ERROR) << "ValidateShape fail [input size error " << input_tensor_nodes_.size() << ", need " << input_size << "]";
        return Status::kErrorShape;
    }
    if (output_size >= 0 && output_size != (int)output_tensor_nodes_.size()) {
        LOG(ERROR) << "ValidateShape fail [output size error " << output_tensor_nodes_.size() << ", need " << output_size << "]";
        return Status::kErrorShape;
    }
    return Status::kSuccess;
}

Status Layer::ValidateFloat32() {
    for (auto* n : input_tensor_nodes_)
        if (!IsSameDataType<float>(n->tensor.GetDataType())) return Status::kUnsupport;
    for (auto* n : output_tensor_nodes_)
        if (!IsSameDataType<float>(n->tensor.GetDataType())) return Status::kUnsupport;
    return Status::kSuccess;
}

Status Layer::ValidateFloat() {
    auto ok = [](const Tensor& t) { return t.GetDataType() == DataType::kFloat32 || t.GetDataType() == DataType::kFloat16; };
    for (auto* n : input_tensor_nodes_)
        if (!ok(n->tensor)) return Status::kUnsupport;
    for (auto* n : output_tensor_nodes_)
        if (!ok(n->tensor)) return Status::kUnsupport;
    return Status::kSuccess;
}

si_stream_t Layer::Stream() const { return context_ ? context_->stream() : Context::Default()->stream(); }

Status Layer::CheckHip(int rc, const char* what) const {
    if (rc == 0) return Status::kSuccess;
    LOG(ERROR) << what << " [" << (op_ ? op_->name : std::string("-")) << "]: " << si_hip_error_string(rc);
    return rc == SI_E_UNSUPPORTED ? Status::kUnsupport : Status::kFail;
}

Status Layer::RunOnDevice(const std::vector<const Tensor*>& inputs, const std::vector<Tensor*>& outputs,
                          const DeviceFn& fn) {
    bool any_host = false;
    for (auto* t : inputs) any_host = any_host || t->GetMemoryType() == MemoryType::kHost;
    for (auto* t : outputs) any_host = any_host || t->GetMemoryType() == MemoryType::kHost;

    std::vector<Tensor> din(inputs.size()), dout(outputs.size());
    if (!any_host) {
        for (size_t i = 0; i < inputs.size(); ++i) din[i] = *inputs[i];
        for (size_t i = 0; i < outputs.size(); ++i) dout[i] = *outputs[i];
        return fn(din, dout);
    }

    // staged path: unit tests and host-side callers
    si_stream_t s = Stream();
    for (size_t i = 0; i < inputs.size(); ++i) {
        const Tensor& t = *inputs[i];
        if (t.GetMemoryType() == MemoryType::kDevice) {
            din[i] = t;
            continue;
        }
        if (nullptr == t.RawData()) return Status::kEmpty;
        // Tensor assignment aliases (reference semantics), so allocate in place: din[i] must own the staging buffer
        din[i] = Tensor(t.GetDataType(), t.Shape(), MemoryType::kDevice, false);
        CHECK_STATUS(din[i].Allocate());
        CHECK_STATUS(CheckHip(si_hip_memcpy_h2d(din[i].RawData(), t.RawData(), t.ByteSize(), s), "h2d"));
    }
    for (size_t i = 0; i < outputs.size(); ++i) {
        Tensor& t = *outputs[i];
        if (t.GetMemoryType() == MemoryType::kDevice) {
            dout[i] = t;
            continue;
        }
        if (nullptr == t.RawData()) return Status::kEmpty;
        dout[i] = Tensor(t.GetDataType(), t.Shape(), MemoryType::kDevice, false);
        CHECK_STATUS(dout[i].Allocate());
    }
    CHECK_STATUS(fn(din, dout));
    for (size_t i = 0; i < outputs.size(); ++i) {
        Tensor& t = *outputs[i];
        if (t.GetMemoryType() == MemoryType::kDevice) continue;
        CHECK_STATUS(CheckHip(si_hip_memcpy_d2h(t.RawData(), dout[i].RawData(), t.ByteSize(), s), "d2h"));
    }
    return CheckHip(si_hip_stream_sync(s), "sync");
}

}  // namespace SimpleInfer
