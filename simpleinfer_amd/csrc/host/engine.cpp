// engine.cpp -- pimpl forwarding (reference src/engine.cpp:8-48).
#include "engine.h"

#include "engine_impl.h"
#include "logger.h"

namespace SimpleInfer {

Engine::Engine() : impl_(new EngineImpl) {}

Engine::~Engine() {
    delete impl_;
    impl_ = nullptr;
}

Status Engine::LoadModel(const std::string& parampath, const std::string& binpath) { return impl_->LoadModel(parampath, binpath); }
Status Engine::Release() { return impl_->Release(); }
const std::vector<std::string> Engine::InputNames() { return impl_->InputNames(); }
const std::vector<std::string> Engine::OutputNames() { return impl_->OutputNames(); }
Status Engine::Input(const std::string& name, const Tensor& input) { return impl_->Input(name, input); }
Status Engine::Forward() { return impl_->Forward(); }
Status Engine::ForwardAsync() { return impl_->ForwardAsync(); }
Status Engine::Sync() { return impl_->Sync(); }
Status Engine::Extract(const std::string& name, Tensor& output) { return impl_->Extract(name, output); }

Status Engine::Output(const std::string& name, const Tensor& output) { return impl_->Output(name, output); }
Status Engine::SetOption(const std::string& key, int value) { return impl_->SetOption(key, value); }
Status Engine::OperandShape(const std::string& name, std::vector<int>& shape) { return impl_->OperandShape(name, shape); }
Status Engine::Profile(std::vector<LayerProfile>& layers) { return impl_->Profile(layers); }
void* Engine::Stream() { return impl_->Stream(); }
float Engine::LastForwardMs() { return impl_->LastForwardMs(); }

void InitializeContext() { InitializeLogger(); }

}  // namespace SimpleInfer
