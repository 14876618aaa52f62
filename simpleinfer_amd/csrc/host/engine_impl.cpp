#include "engine_impl.h"

#include <exception>

#include <algorithm>

#include "layer/activation.h"
#include "layer/binary_op.h"
#include "layer/cat.h"
#include "layer/conv_2d.h"
#include "layer/linear.h"
#include "layer/max_pool_2d.h"
#include "layer/output_cast.h"
#include "layer/upsample.h"
#include "layer/yolo_detect.h"
#include "layer_registry.h"
#include "logger.h"
#include "pnnx/expand_expression.h"

namespace SimpleInfer {

namespace {

// built-in operator types whose kernels honour a pixel stride on inputs and outputs
bool HonoursPixelStride(const std::string& type) {
    static const std::set<std::string> ok = {
        "nn.Conv2d", "nn.SiLU", "nn.ReLU", "nn.Sigmoid", "nn.Hardsigmoid", "nn.Hardswish", "nn.LeakyReLU",
        "nn.MaxPool2d", "nn.AdaptiveAvgPool2d", "nn.Upsample", "torch.cat", "BinaryOp", "UnaryOp", "nn.BatchNorm2d",
        "torch.flatten", "models.yolo.Detect", "pnnx.Output"};
    return ok.count(type) > 0;
}

#define SI_TRY_HIP(expr, what)                                             \
    {                                                                      \
        const int _rc = (expr);                                            \
        if (_rc != 0) {                                                    \
            LOG(ERROR) << what << ": " << si_hip_error_string(_rc);        \
            return Status::kFail;                                          \
        }                                                                  \
    }

}  // namespace

EngineImpl::EngineImpl() {}

EngineImpl::~EngineImpl() { Release(); }

Status EngineImpl::SetOption(const std::string& key, int value) {
    if (key == "device") opt_device_ = value;
    else if (key == "fuse") opt_fuse_ = value != 0;
    else if (key == "alias_cat") opt_alias_cat_ = value != 0;
    else if (key == "graph") opt_graph_ = value != 0;
    else if (key == "outputs_to_host") opt_outputs_to_host_ = value != 0;
    else if (key == "fp16") opt_fp16_ = value != 0;
    else if (key == "batch") opt_batch_ = value;  // > 0: re-batch the graph at load (the file bakes its batch into every shape)
    else if (key == "arena") opt_arena_ = value != 0;  // 1 (default): intermediates share one arena by lifetime; 0: one hipMalloc each
    else if (key == "fuse_pw") opt_fuse_pw_ = value;           // fp16: 1 (default) a C3 bottleneck's 1x1 conv computed inside the kernel of its 3x3 conv; 2 ... and the C3's closing conv behind its last 64-channel pair (measured slower: opt-in); 0 off
    else if (key == "fuse_stem") opt_fuse_stem_ = value;       // fp16: 1 RGB stem conv + the 3x3 s2 conv behind it in one launch; 2 (default) ... and the 1x1 conv(s) reading that; 0 off
    else if (key == "fuse_upsample") opt_fuse_upsample_ = value != 0;  // upsample -> cat -> 1x1 conv read at the source (default 1)
    else if (key == "detect_priority") opt_detect_priority_ = value < 0 ? -1 : (value > 0 ? 1 : 0);   // priority of Detect's side stream: -1 low, 0 default, +1 high
    else if (key == "detect_stream") opt_detect_stream_ = value;  // Detect's early levels on a second stream: 0 never, 1 (default) where they have enough work, 2 always
    else if (key == "f32_split") opt_f32_split_ = value != 0;   // fp32 tensors, conv contraction from three fp16 MFMA products (default 0: true fp32)
    // kernel-form choices (SiConvPlan, include/si_hip.h; every form of a family produces the same bits): A/B runs and tests.  Until round 6
    // these were process-global setters / environment switches of the kernel library.
    else if (key == "f32_tile") { opt_plan_.f32_tile = value; opt_plan_set_ = true; }
    else if (key == "wino23_form") { opt_plan_.wino23_form = value; opt_plan_set_ = true; }
    else if (key == "wino23_ocg") { opt_plan_.wino23_ocg = value; opt_plan_set_ = true; }
    else if (key == "f16_tile") { opt_plan_.f16_tile = value; opt_plan_set_ = true; }
    else if (key == "f16_detect_tile") { opt_plan_.f16_detect_tile = value; opt_plan_set_ = true; }
    else if (key == "f16_s2c32") { opt_plan_.f16_s2c32 = value; opt_plan_set_ = true; }
    else if (key == "f16_slab") { opt_plan_.f16_slab = value; opt_plan_set_ = true; }
    else if (key == "f16_slab_w2") { opt_plan_.f16_slab_w2 = value; opt_plan_set_ = true; }
    else if (key == "f16_pw_patch") { opt_plan_.f16_pw_patch = value; opt_plan_set_ = true; }
    else if (key == "split3_bm") { opt_plan_.split3_bm = value; opt_plan_set_ = true; }
    else if (key == "f32_split_policy") opt_f32_split_policy_ = value;   // which layers f32_split takes: 1 round 5's, 2 + sibling-fused / wide 1x1 from K = 256, 3 + dual-source (upsampled) layers, 4 (default) + the RGB stem
    else if (key == "winograd") opt_winograd_ = value;  // 0 off, 1 F(2,3) where faster (default), 2 F(4,3) on those layers
    else if (key == "streams") opt_streams_ = value;  // 2: two half-batch lanes on two streams, 1 (default): one stream
    else if (key == "_fail_slicer") debug_fail_slicer_ = value != 0;  // tests: the sliced pipeline's setup fails half way
    else if (key == "pin_inputs") opt_pin_inputs_ = value != 0;  // host_slices: pin a borrowed input buffer in place (default 0, see engine.h)
    else if (key == "host_slices") opt_host_slices_ = value;  // host tensors in and out: G pipelined batch slices per Forward(), 1 off, 0 (default) auto
    else {
        LOG(ERROR) << "unknown engine option [" << key << "]";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

Status EngineImpl::LoadModel(const std::string& parampath, const std::string& binpath) {
    struct Stage {
        const char* name;
        Status (EngineImpl::*fn)();
    };
    {
        Status ret = Release();
        if (Status::kSuccess != ret) {
            LOG(ERROR) << "Release fail";
            return ret;
        }
    }
    {
        Status ret = CreateContext();
        if (Status::kSuccess != ret) {
            LOG(ERROR) << "CreateContext fail";
            Release();
            return ret;
        }
    }
    {
        // The reference's loader can throw on malformed files (std::stoi / map::at); nothing may
        // propagate out of this library (there is a C-ABI on top), so a bad model is a Status.
        Status ret = Status::kFail;
        try {
            ret = CreateGraph(parampath, binpath);
        } catch (const std::exception& ex) {
            LOG(ERROR) << "malformed model file: " << ex.what();
            ret = Status::kFail;
        }
        if (Status::kSuccess != ret) {
            LOG(ERROR) << "CreateGraph fail";
            Release();
            return ret;
        }
    }
    param_path_ = parampath;
    bin_path_ = binpath;
    {
        Status ret = CreateTensorNodes();
        if (Status::kSuccess != ret) {
            LOG(ERROR) << "CreateTensorNodes fail";
            Release();
            return ret;
        }
    }
    int lanes = 1;
    {
        Status ret = PlanLanes(lanes);
        if (Status::kSuccess == ret && lanes > 1) {
            ret = LoadLanes(lanes);
            if (Status::kSuccess == ret) ret = AllocateTensorMemory();   // graph inputs / outputs only: the lanes own the rest
        }
        if (Status::kSuccess != ret) {
            LOG(ERROR) << "lane setup fail";
            Release();
            return ret;
        }
        if (lanes > 1) return Status::kSuccess;
    }
    const Stage stages[] = {{"CreateLayers", &EngineImpl::CreateLayers},
                            {"CreatePipeline", &EngineImpl::CreatePipeline},
                            {"AllocateTensorMemory", &EngineImpl::AllocateTensorMemory}};
    for (const Stage& s : stages) {
        Status ret = Status::kFail;
        try {
            ret = (this->*s.fn)();
        } catch (const std::exception& ex) {
            LOG(ERROR) << s.name << ": " << ex.what();
            ret = Status::kFail;
        }
        if (Status::kSuccess != ret) {
            LOG(ERROR) << s.name << " fail";
            Release();
            return ret;
        }
    }
    return Status::kSuccess;
}

Status EngineImpl::Release() {
    if (context_ && context_->stream()) si_hip_stream_sync(context_->stream());
    CHECK_STATUS(DestroyGraphCache());   // (a captured graph references the lanes' streams and buffers)
    CHECK_STATUS(DestroySlicer());
    slicer_failed_ = false;
    UnpinInputs();
    CHECK_STATUS(DestroyLanes());
    CHECK_STATUS(DeallocateTensorMemory());
    CHECK_STATUS(DestroyPipeline());
    CHECK_STATUS(DestroyLayers());
    CHECK_STATUS(DestroyTensorNodes());
    CHECK_STATUS(DestroyGraph());
    CHECK_STATUS(DestroyContext());
    return Status::kSuccess;
}

Status EngineImpl::CreateContext() {
    context_ = new Context;
    CHECK_STATUS(context_->Init(opt_device_));
    SI_TRY_HIP(si_hip_event_create(&ev_start_), "event create");
    SI_TRY_HIP(si_hip_event_create(&ev_stop_), "event create");
    return Status::kSuccess;
}

Status EngineImpl::DestroyContext() {
    if (ev_start_) si_hip_event_destroy(ev_start_);
    if (ev_stop_) si_hip_event_destroy(ev_stop_);
    if (ev_fork_) si_hip_event_destroy(ev_fork_);
    if (ev_join_) si_hip_event_destroy(ev_join_);
    ev_start_ = ev_stop_ = ev_fork_ = ev_join_ = nullptr;
    delete side_context_;
    side_context_ = nullptr;
    delete context_;
    context_ = nullptr;
    return Status::kSuccess;
}

Status EngineImpl::CreateGraph(const std::string& parampath, const std::string& binpath) {
    graph_ = new pnnx::Graph;
    if (0 != graph_->load(parampath, binpath)) {
        LOG(ERROR) << "load graph fail";
        return Status::kFail;
    }
    pnnx::expand_expression(*graph_);
    return Status::kSuccess;
}

Status EngineImpl::DestroyGraph() {
    delete graph_;
    graph_ = nullptr;
    return Status::kSuccess;
}

Status EngineImpl::CreateTensorNodes() {
    // Re-batch (SetOption("batch", N)): a pnnx file carries the batch it was traced with in every operand shape
    // (#0=(N,3,640,640)f32).  Every operator here is per-image, so serving another batch is a rewrite of
    // dim 0 wherever it equals the traced batch of the graph input.
    int file_batch = 0;
    if (opt_batch_ > 0) {
        for (pnnx::Operand* opd : graph_->operands)
            if (opd->producer && opd->producer->inputs.empty() && !opd->shape.empty()) {
                if (file_batch != 0 && file_batch != opd->shape[0]) {
                    LOG(ERROR) << "re-batch: graph inputs disagree on the batch dimension";
                    return Status::kUnsupport;
                }
                file_batch = opd->shape[0];
            }
        if (file_batch <= 0) {
            LOG(ERROR) << "re-batch: no graph input with a static batch dimension";
            return Status::kUnsupport;
        }
    }
    for (pnnx::Operand* opd : graph_->operands) {
        if (tensor_nodes_.count(opd->name) > 0) {
            LOG(ERROR) << "tensor node [" << opd->name << "] already exists";
            return Status::kFail;
        }
        TensorNode* node = new TensorNode;
        node->operand = opd;

        // file shapes are NCHW; tensors are NHWC: the last three dims C,H,W -> H,W,C for rank >= 4
        std::vector<int> shape = opd->shape;
        if (file_batch > 0 && !shape.empty() && shape[0] == file_batch) shape[0] = opt_batch_;
        const int rank = (int)shape.size();
        if (rank > 3) {
            shape[rank - 3] = opd->shape[rank - 2];
            shape[rank - 2] = opd->shape[rank - 1];
            shape[rank - 1] = opd->shape[rank - 3];
        }
        DataType dt = PnnxToDataType(opd->type);
        // graph inputs: produced by an operator with no inputs (pnnx.Input)
        const bool is_input = opd->producer && opd->producer->inputs.empty();
        // graph outputs: consumed by an operator with no outputs (pnnx.Output)
        bool is_output = false;
        for (pnnx::Operator* c : opd->consumers)
            if (c && c->outputs.empty()) is_output = true;
        // fp16 storage: every internal fp32 activation becomes fp16; the tensors the caller hands over / reads back
        // (Input / Extract) keep the file's type
        if (opt_fp16_ && dt == DataType::kFloat32 && !is_input && !is_output) dt = DataType::kFloat16;
        node->tensor = Tensor(dt, shape, MemoryType::kDevice, false);
        tensor_nodes_[opd->name] = node;
        if (is_input) input_tensor_nodes_[opd->name] = node;
        if (is_output) output_tensor_nodes_[opd->name] = node;
    }
    return Status::kSuccess;
}

Status EngineImpl::DestroyTensorNodes() {
    input_tensor_nodes_.clear();
    output_tensor_nodes_.clear();
    user_inputs_.clear();
    user_outputs_.clear();
    for (auto& kv : tensor_nodes_) delete kv.second;
    tensor_nodes_.clear();
    return Status::kSuccess;
}

Status EngineImpl::CreateLayers() {
    has_user_layers_ = false;
    for (pnnx::Operator* op : graph_->ops) {
        if ("pnnx.Input" == op->type || "pnnx.Output" == op->type) continue;
        if (IsUserRegisteredLayer(op->type)) has_user_layers_ = true;
        if (layers_.count(op->name) > 0) {
            LOG(ERROR) << "layer [" << op->name << "] already exists";
            return Status::kFail;
        }
        const LayerRegistryEntry* entry = GetLayerRegistry(op->type);
        if (nullptr == entry) {
            LOG(ERROR) << "layer type [" << op->type << "] not registered";
            return Status::kEmpty;
        }
        Layer* layer = entry->creator();
        if (nullptr == layer) {
            LOG(ERROR) << "create layer [" << op->type << "] fail";
            return Status::kFail;
        }
        layers_[op->name] = layer;  // registered first so a failing Init is still destroyed

        Status ret = layer->Init(op);
        if (Status::kSuccess != ret) {
            LOG(ERROR) << "layer [" << op->name << "] init fail";
            return ret;
        }
        layer->SetContext(context_);
        // f32_split: every conv that may run on the split kernels gets its word of the engine's range-guard flags (pinned host memory the
        // kernels can write; 1024 words: more convs than that simply run unguarded-by-flag on the fp32 kernels)
        auto arm_split = [&](Conv2d& cv, bool on) {
            cv.f32_split_ = on;
            cv.f32_split_level_ = opt_f32_split_policy_;
            if (opt_plan_set_) cv.SetPlan(opt_plan_);
            if (!on) return;
            if (!split_flags_) {
                void* p = nullptr;
                if (si_hip_host_alloc(&p, 1024 * sizeof(unsigned)) != 0 || !p) { cv.f32_split_ = false; return; }
                split_flags_ = static_cast<unsigned*>(p);
                for (int i = 0; i < 1024; ++i) split_flags_[i] = 0u;
            }
            if (split_convs_.size() >= 1024) { cv.f32_split_ = false; return; }
            cv.range_flag_ = split_flags_ + split_convs_.size();
            split_convs_.push_back(&cv);
        };
        if (YoloDetect* yd = dynamic_cast<YoloDetect*>(layer)) {
            yd->fuse_decode_ = opt_fuse_;
            for (Conv2d& cv : yd->conv_2d_layer_) arm_split(cv, opt_f32_split_ && !opt_fp16_ && opt_fuse_);
        }
        if (Conv2d* cv = dynamic_cast<Conv2d*>(layer)) {
            if (!opt_winograd_) cv->algo_ = Conv2d::Algo::kImplicitGemm;
            cv->prefer_wino43_ = opt_winograd_ == 2;
            arm_split(*cv, opt_f32_split_ && !opt_fp16_);
        }

        std::vector<TensorNode*> ins, outs;
        for (pnnx::Operand* r : op->inputs) {
            auto it = tensor_nodes_.find(r->name);
            if (it == tensor_nodes_.end()) {
                LOG(ERROR) << "tensor node [" << r->name << "] not exist";
                return Status::kEmpty;
            }
            ins.push_back(it->second);
        }
        for (pnnx::Operand* r : op->outputs) {
            auto it = tensor_nodes_.find(r->name);
            if (it == tensor_nodes_.end()) {
                LOG(ERROR) << "tensor node [" << r->name << "] not exist";
                return Status::kEmpty;
            }
            outs.push_back(it->second);
        }
        layer->SetInputNodes(ins);
        layer->SetOutputNodes(outs);

        ret = layer->Validate();
        if (Status::kSuccess != ret) {
            LOG(ERROR) << "layer [" << op->name << "] validate fail";
            return ret;
        }
    }
    return Status::kSuccess;
}

Status EngineImpl::DestroyLayers() {
    for (auto& kv : layers_) {
        Layer* layer = kv.second;
        const pnnx::Operator* op = layer->GetOp();
        Status ret = layer->Deinit();
        if (Status::kSuccess != ret) LOG(ERROR) << "layer [" << kv.first << "] deinit fail";
        const LayerRegistryEntry* entry = op ? GetLayerRegistry(op->type) : nullptr;
        if (entry) {
            entry->destroyer(layer);
        } else {
            delete layer;
        }
    }
    layers_.clear();
    split_convs_.clear();
    if (split_flags_) si_hip_host_free(split_flags_);
    split_flags_ = nullptr;
    return Status::kSuccess;
}

// ---- schedule ------------------------------------------------------------------------------------
Status EngineImpl::CreatePipeline() {
    // Kahn topological order over layer operators, stable w.r.t. file order (pnnx writes operators
    // in execution order, and expression lowering inserts before the expression op)
    std::vector<Step> order;
    std::set<const pnnx::Operand*> ready;
    for (auto& kv : input_tensor_nodes_) ready.insert(kv.second->operand);
    std::vector<const pnnx::Operator*> pending;
    for (pnnx::Operator* op : graph_->ops)
        if (layers_.count(op->name)) pending.push_back(op);
    while (!pending.empty()) {
        bool progressed = false;
        for (auto it = pending.begin(); it != pending.end();) {
            const pnnx::Operator* op = *it;
            bool ok = true;
            for (pnnx::Operand* r : op->inputs) ok = ok && ready.count(r) > 0;
            if (!ok) {
                ++it;
                continue;
            }
            Step s;
            s.layer = layers_[op->name];
            s.op = op;
            order.push_back(s);
            for (pnnx::Operand* r : op->outputs) ready.insert(r);
            it = pending.erase(it);
            progressed = true;
        }
        if (!progressed) {
            LOG(ERROR) << "graph has a cycle or an operand without producer near [" << pending.front()->name << "]";
            return Status::kFail;
        }
    }

    if (opt_fuse_) {
        CHECK_STATUS(FuseEpilogues(order));
        CHECK_STATUS(FuseSiblingConvs(order));
        CHECK_STATUS(FusePoolChains(order));
        if (opt_fuse_upsample_ && opt_alias_cat_) CHECK_STATUS(FuseUpsampleIntoConvs(order));   // (fp16 storage too since round 4)
        if (opt_fp16_ && opt_fuse_stem_) CHECK_STATUS(FuseStemPairs(order));
        if (opt_fp16_ && opt_fuse_stem_ > 1) CHECK_STATUS(FuseStemTriples(order));
        if (opt_fp16_ && opt_fuse_pw_) CHECK_STATUS(FuseBottleneckPairs(order));
        if (opt_fp16_ && opt_fuse_pw_ > 1 && opt_alias_cat_) CHECK_STATUS(FuseCv3IntoPairs(order));
    }
    if (opt_fp16_) {
        CHECK_STATUS(InsertOutputCasts(order));
        // every layer is asked NOW whether it has a kernel for the storage types it ended up with.  One that has none (a 3x3 conv
        // whose channel count is not a multiple of 32, UnaryOp ...) runs its fp32 kernel between two casts (round 4); only what
        // even that cannot serve makes LoadModel fail -- at load, with the layer and the reason, never at the first Forward
        CHECK_STATUS(InsertFp32Fallbacks(order));
    }
    plan_ = order;
    if (opt_alias_cat_) CHECK_STATUS(AliasConcats());
    if (opt_detect_stream_) CHECK_STATUS(PlanDetectStream());
    return Status::kSuccess;
}

// Option "detect_stream": a Detect level only needs its own feature map, and the two finer maps are final well before the last
// PAN layer.  Each such level is launched on a second stream right after the step that completes its input (fork: an event on
// the main stream), beside the layers that follow; the Detect step launches the remaining level and joins (the main stream
// waits for the side stream's event).  The output tensor is a graph output with its own allocation, so early writes into it
// touch nothing the arena shares.  Under hipGraph capture the side stream joins the capture through the same events.
Status EngineImpl::PlanDetectStream() {
    for (size_t di = 0; di < plan_.size(); ++di) {
        YoloDetect* det = dynamic_cast<YoloDetect*>(plan_[di].layer);
        if (!det) continue;
        if (det->OutputNodes().size() != 1 || !det->OutputNodes()[0]->operand ||
            !output_tensor_nodes_.count(det->OutputNodes()[0]->operand->name))
            continue;   // only when Detect writes a graph output (own buffer)
        unsigned mask = 0;
        std::vector<int> producer(det->InputNodes().size(), -1);
        for (size_t k = 0; k < det->InputNodes().size(); ++k) {
            TensorNode* n = det->InputNodes()[k];
            if (!n || !n->operand || aliases_.count(n->operand->name)) continue;   // a view into a concat buffer has several writers
            bool aliased_into = false;
            for (auto& kv : aliases_) aliased_into = aliased_into || kv.second.parent == n;
            if (aliased_into) continue;
            for (size_t j = 0; j < di; ++j) {
                std::vector<TensorNode*> outs = plan_[j].layer->OutputNodes();
                if (std::find(outs.begin(), outs.end(), n) != outs.end()) producer[k] = (int)j;
            }
            // Worth a fork / join only for a level with real work (MI355X, YOLOv5s same-box A/B: batch 32 +1.3 %, batch 8 +-0,
            // batch 1 -3 %: two more graph edges against launches of a few microseconds)
            const double level_flops = 2.0 * (double)n->tensor.NumElements() * det->num_elements_;
            if (producer[k] >= 0 && producer[k] + 1 < (int)di && (level_flops >= 4e9 || opt_detect_stream_ >= 2)) mask |= 1u << k;   // something runs in between
        }
        if (!mask) continue;
        if (!side_context_) {
            side_context_ = new Context;
            CHECK_STATUS(side_context_->Init(context_->device(), opt_detect_priority_));
            SI_TRY_HIP(si_hip_event_create(&ev_fork_), "event create");
            SI_TRY_HIP(si_hip_event_create(&ev_join_), "event create");
        }
        det->SetEarlyLevels(side_context_, mask);
        for (size_t k = 0; k < producer.size(); ++k)
            if ((mask >> k) & 1u) plan_[producer[k]].detect_levels.push_back((int)k);
        LOG(INFO) << "detect_stream: levels mask " << mask << " of [" << plan_[di].op->name << "] launch on the side stream";
    }
    return Status::kSuccess;
}

// A cached graph exec may still be in flight (ForwardAsync without Sync): wait for the stream before destroying any.
Status EngineImpl::DestroyGraphCache() {
    if (!graph_cache_.empty() && context_ && context_->stream()) si_hip_stream_sync(context_->stream());
    for (auto& g : graph_cache_) si_hip_graph_destroy(g.second);
    graph_cache_.clear();
    return Status::kSuccess;
}

Status EngineImpl::DestroyPipeline() {
    CHECK_STATUS(DestroyGraphCache());
    forward_count_ = 0;
    plan_warm_ = false;
    plan_.clear();
    fused_ops_.clear();
    sibling_ops_.clear();
    dead_operands_.clear();
    aliases_.clear();
    return Status::kSuccess;
}

// conv -> [act] -> [add residual -> [act]]  ==>  one conv launch.  The fused conv runs at the slot of
// the LAST operator of the chain, so a residual produced between the conv and the add is ready.
Status EngineImpl::FuseEpilogues(std::vector<Step>& order) {
    std::map<const pnnx::Operator*, size_t> index;
    for (size_t i = 0; i < order.size(); ++i) index[order[i].op] = i;
    std::vector<bool> removed(order.size(), false);

    auto sole_consumer = [&](const pnnx::Operand* r) -> const pnnx::Operator* {
        if (output_tensor_nodes_.count(r->name)) return nullptr;
        if (r->consumers.size() != 1) return nullptr;
        const pnnx::Operator* c = r->consumers[0];
        if (!index.count(c) || removed[index[c]]) return nullptr;  // not a layer, or already absorbed
        return c;
    };

    for (size_t i = 0; i < order.size(); ++i) {
        if (removed[i]) continue;
        Conv2d* conv = dynamic_cast<Conv2d*>(order[i].layer);
        if (!conv || order[i].op->type != "nn.Conv2d" || order[i].op->outputs.size() != 1) continue;

        const pnnx::Operand* cur = order[i].op->outputs[0];
        size_t last = i;
        int act1 = SI_ACT_NONE, act2 = SI_ACT_NONE;
        float act_param = 0.f;
        TensorNode* residual = nullptr;
        std::vector<const pnnx::Operator*> absorbed;
        std::vector<const pnnx::Operand*> dead;

        auto try_act = [&](int& slot) {
            const pnnx::Operator* c = sole_consumer(cur);
            if (!c) return;
            ActivationLayer* a = dynamic_cast<ActivationLayer*>(order[index[c]].layer);
            if (!a || c->inputs.size() != 1 || c->outputs.size() != 1) return;
            if (slot != SI_ACT_NONE) return;
            if (a->ActCode() == SI_ACT_LEAKYRELU && (act1 == SI_ACT_LEAKYRELU) && act_param != a->ActParam()) return;
            slot = a->ActCode();
            if (a->ActCode() == SI_ACT_LEAKYRELU) act_param = a->ActParam();
            absorbed.push_back(c);
            dead.push_back(cur);
            cur = c->outputs[0];
            last = index[c];
        };

        try_act(act1);
        {
            const pnnx::Operator* c = sole_consumer(cur);
            BinaryOp* b = c ? dynamic_cast<BinaryOp*>(order[index[c]].layer) : nullptr;
            if (b && b->binary_op_type_ == BinaryOp::BinaryOpType::kAdd && c->inputs.size() == 2 && c->outputs.size() == 1 &&
                c->inputs[0] != c->inputs[1]) {
                const pnnx::Operand* other = c->inputs[0] == cur ? c->inputs[1] : c->inputs[0];
                const std::vector<int>& so = tensor_nodes_[c->outputs[0]->name]->tensor.Shape();
                const bool same = IsSameShape(tensor_nodes_[other->name]->tensor.Shape(), so) &&
                                  IsSameShape(tensor_nodes_[cur->name]->tensor.Shape(), so);
                if (same) {
                    residual = tensor_nodes_[other->name];
                    absorbed.push_back(c);
                    dead.push_back(cur);
                    cur = c->outputs[0];
                    last = index[c];
                    try_act(act2);
                }
            }
        }
        if (absorbed.empty()) continue;

        conv->SetFusion(act1, residual, act2, act_param);
        conv->SetOutputNodes({tensor_nodes_[cur->name]});
        for (const pnnx::Operator* c : absorbed) {
            removed[index[c]] = true;
            fused_ops_.insert(c->name);
        }
        for (const pnnx::Operand* r : dead) dead_operands_.insert(r->name);
        if (last != i) {
            order[last] = order[i];
            removed[last] = false;
            removed[i] = true;
        }
    }

    std::vector<Step> out;
    for (size_t i = 0; i < order.size(); ++i)
        if (!removed[i]) out.push_back(order[i]);
    order.swap(out);
    return Status::kSuccess;
}

// Two 1x1 convs reading the same operand with the same geometry (YOLOv5 C3: cv1 and cv2) become one launch with
// twice the output channels and a split destination: the input is read once and the launch has twice the tiles.
Status EngineImpl::FuseSiblingConvs(std::vector<Step>& order) {
    std::vector<bool> removed(order.size(), false);
    for (size_t i = 0; i < order.size(); ++i) {
        if (removed[i]) continue;
        Conv2d* a = dynamic_cast<Conv2d*>(order[i].layer);
        if (!a || order[i].op->type != "nn.Conv2d" || a->InputNodes().size() != 1 || a->OutputNodes().size() != 1) continue;
        for (size_t j = i + 1; j < order.size(); ++j) {
            if (removed[j]) continue;
            Conv2d* b = dynamic_cast<Conv2d*>(order[j].layer);
            if (!b || order[j].op->type != "nn.Conv2d" || b->InputNodes().size() != 1 || b->OutputNodes().size() != 1) continue;
            if (a->InputNodes()[0] != b->InputNodes()[0] || !a->CanFuseSibling(*b)) continue;
            const std::vector<int>& sa = a->OutputNodes()[0]->tensor.Shape();
            const std::vector<int>& sb = b->OutputNodes()[0]->tensor.Shape();
            if (sa.size() != 4 || sb.size() != 4 || sa[0] != sb[0] || sa[1] != sb[1] || sa[2] != sb[2]) continue;
            a->SetSibling(b);
            a->SetOutputNodes({a->OutputNodes()[0], b->OutputNodes()[0]});
            removed[j] = true;
            sibling_ops_.insert(order[j].op->name);
            break;
        }
    }
    std::vector<Step> out;
    for (size_t i = 0; i < order.size(); ++i)
        if (!removed[i]) out.push_back(order[i]);
    order.swap(out);
    return Status::kSuccess;
}

// SPPF: maxpool5 -> maxpool5 -> maxpool5 (each fed by the previous one, every intermediate also read by the concat)
// ==> one launch at the first pool's slot that reads the input once and writes all three operands.
Status EngineImpl::FusePoolChains(std::vector<Step>& order) {
    std::vector<bool> removed(order.size(), false);
    auto pool_at = [&](size_t i) -> MaxPool2d* {
        if (removed[i]) return nullptr;
        MaxPool2d* p = dynamic_cast<MaxPool2d*>(order[i].layer);
        return (p && p->InputNodes().size() == 1 && p->OutputNodes().size() == 1 && p->chain_.empty()) ? p : nullptr;
    };
    auto follower = [&](size_t from, MaxPool2d* head) -> size_t {
        for (size_t j = from + 1; j < order.size(); ++j) {
            MaxPool2d* p = pool_at(j);
            if (p && p->InputNodes()[0] == head->OutputNodes()[0] && head->ChainHead(*p) &&
                IsSameShape(p->OutputNodes()[0]->tensor.Shape(), head->OutputNodes()[0]->tensor.Shape()))
                return j;
        }
        return 0;
    };
    for (size_t i = 0; i < order.size(); ++i) {
        MaxPool2d* a = pool_at(i);
        if (!a || !IsSameShape(a->InputNodes()[0]->tensor.Shape(), a->OutputNodes()[0]->tensor.Shape())) continue;
        const size_t j = follower(i, a);
        if (j == 0) continue;
        MaxPool2d* b = pool_at(j);
        const size_t k = follower(j, b);
        if (k == 0) continue;
        MaxPool2d* c = pool_at(k);
        a->SetChain(b, c);
        a->SetOutputNodes({a->OutputNodes()[0], b->OutputNodes()[0], c->OutputNodes()[0]});
        removed[j] = removed[k] = true;
        fused_ops_.insert(order[j].op->name);
        fused_ops_.insert(order[k].op->name);
    }
    std::vector<Step> out;
    for (size_t i = 0; i < order.size(); ++i)
        if (!removed[i]) out.push_back(order[i]);
    order.swap(out);
    return Status::kSuccess;
}

// nn.Upsample(nearest) -> torch.cat(dim = channels) -> 1x1 convs only  ==>  the upsample launch disappears: every consumer conv
// reads the upsampled channel range from the LOW-RESOLUTION tensor at the nearest-neighbour source pixel (dual-source A rows,
// si_hip_conv2d_upcat_f32), with the reference's index rule (src/layer/upsample.cpp:85-92), so the results are bit-identical to
// the unfused schedule.  YOLOv5s: both upsamples of the PAN top-down path (20x20x256 -> 40x40, 40x40x128 -> 80x80): 131 MB per
// batch-32 forward that are neither written nor read back.  Requires concat aliasing (the cat then has nothing to copy for that
// input) and fp32 storage.
Status EngineImpl::FuseUpsampleIntoConvs(std::vector<Step>& order) {
    std::map<const pnnx::Operator*, size_t> index;
    for (size_t i = 0; i < order.size(); ++i) index[order[i].op] = i;
    std::vector<bool> removed(order.size(), false);
    for (size_t i = 0; i < order.size(); ++i) {
        Upsample* up = dynamic_cast<Upsample*>(order[i].layer);
        if (!up || order[i].op->type != "nn.Upsample" || up->InputNodes().size() != 1 || up->OutputNodes().size() != 1) continue;
        const pnnx::Operand* u = up->OutputNodes()[0]->operand;
        if (!u || u->consumers.size() != 1 || output_tensor_nodes_.count(u->name)) continue;
        const pnnx::Operator* cat = u->consumers[0];
        if (!cat || cat->type != "torch.cat" || cat->outputs.size() != 1 || !index.count(cat)) continue;
        Cat* cat_layer = dynamic_cast<Cat*>(order[index[cat]].layer);
        if (!cat_layer || cat_layer->NhwcAxis() != 3) continue;
        const std::vector<int>& us = up->OutputNodes()[0]->tensor.Shape();
        if (us.size() != 4) continue;
        // channel offset of the upsampled tensor inside the concat
        int c0 = 0;
        bool found = false;
        for (const pnnx::Operand* r : cat->inputs) {
            if (r == u) { found = true; break; }
            const std::vector<int>& rs = tensor_nodes_[r->name]->tensor.Shape();
            if (rs.size() != 4) { found = false; break; }
            c0 += rs[3];
        }
        if (!found) continue;
        const pnnx::Operand* co = cat->outputs[0];
        if (output_tensor_nodes_.count(co->name) || co->consumers.empty()) continue;
        // every reader of the concat must be a pointwise conv that can take the dual-source form (a sibling-fused secondary has
        // left the order: its primary reads the same operand)
        std::vector<Conv2d*> readers;
        bool ok = true;
        for (const pnnx::Operator* c : co->consumers) {
            if (!c || c->type != "nn.Conv2d") { ok = false; break; }
            if (sibling_ops_.count(c->name)) continue;
            Conv2d* conv = index.count(c) ? dynamic_cast<Conv2d*>(order[index[c]].layer) : nullptr;
            if (!conv || conv->UpsampledSource() || !conv->CanReadUpsampledFrom(up->InputNodes()[0], c0, up->scale_factor_h_, up->scale_factor_w_)) { ok = false; break; }
            readers.push_back(conv);
        }
        // ... and every sibling-fused secondary must have its primary among them
        for (const pnnx::Operator* c : co->consumers) {
            if (!ok || !c || !sibling_ops_.count(c->name)) continue;
            bool has_primary = false;
            for (Conv2d* r : readers) has_primary = has_primary || (r->Sibling() && r->Sibling()->GetOp() == c);
            ok = has_primary;
        }
        if (!ok || readers.empty()) continue;
        for (Conv2d* r : readers) r->SetUpsampledSource(up->InputNodes()[0], c0, up->scale_factor_h_, up->scale_factor_w_);
        removed[i] = true;
        fused_ops_.insert(order[i].op->name);
    }
    std::vector<Step> out;
    for (size_t i = 0; i < order.size(); ++i)
        if (!removed[i]) out.push_back(order[i]);
    order.swap(out);
    return Status::kSuccess;
}

// fp16 storage (round 4): the RGB stem conv (fp32 image in, 32 half channels out) whose ONLY reader is a 3x3 stride-2 conv over those
// 32 channels -- YOLOv5's conv_0 -> conv_1 -- becomes one launch at the second conv's slot (si_hip_conv2d_stem_s2c32_f16): the
// intermediate (210 MB at batch 32) is computed tile by tile in LDS and never written.  The stem's step leaves the order, its
// operand is never allocated; the kernel is asked NOW, with the bound shapes, whether it takes the pair.
Status EngineImpl::FuseStemPairs(std::vector<Step>& order) {
    std::map<const pnnx::Operator*, size_t> index;
    for (size_t i = 0; i < order.size(); ++i) index[order[i].op] = i;
    std::vector<bool> removed(order.size(), false);
    for (size_t i = 0; i < order.size(); ++i) {
        Conv2d* stem = dynamic_cast<Conv2d*>(order[i].layer);
        if (!stem || order[i].op->type != "nn.Conv2d" || stem->InputNodes().size() != 1 || stem->OutputNodes().size() != 1) continue;
        const pnnx::Operand* img = stem->InputNodes()[0]->operand;
        const pnnx::Operand* mid = stem->OutputNodes()[0]->operand;
        if (!img || !mid || !input_tensor_nodes_.count(img->name) || output_tensor_nodes_.count(mid->name) || mid->consumers.size() != 1) continue;
        const pnnx::Operator* c = mid->consumers[0];
        if (!c || c->type != "nn.Conv2d" || !index.count(c) || sibling_ops_.count(c->name)) continue;
        Conv2d* conv = dynamic_cast<Conv2d*>(order[index[c]].layer);
        if (!conv || !conv->CanFuseStemProducer(*stem)) continue;
        conv->SetStemProducer(stem);
        dead_operands_.insert(mid->name);
        removed[i] = true;
        fused_ops_.insert(order[i].op->name);
    }
    std::vector<Step> out;
    for (size_t i = 0; i < order.size(); ++i)
        if (!removed[i]) out.push_back(order[i]);
    order.swap(out);
    return Status::kSuccess;
}

// fp16 storage (round 6): the conv that took the stem (FuseStemPairs) and the 1x1 conv over its 64 channels that is its only reader -- with the
// sibling that conv computes as well: YOLOv5's first C3 reads the tensor twice, cv1 and cv2, which FuseSiblingConvs has made ONE conv -- become
// one launch at the 1x1 conv's slot (si_hip_conv2d_stem_s2c32_pw_f16); the 64-channel tensor between them is never allocated.
Status EngineImpl::FuseStemTriples(std::vector<Step>& order) {
    std::map<const pnnx::Operator*, size_t> index;
    for (size_t i = 0; i < order.size(); ++i) index[order[i].op] = i;
    std::vector<bool> removed(order.size(), false);
    for (size_t i = 0; i < order.size(); ++i) {
        Conv2d* c1 = dynamic_cast<Conv2d*>(order[i].layer);
        if (!c1 || order[i].op->type != "nn.Conv2d" || !c1->StemProducer() || c1->OutputNodes().size() != 1) continue;
        const pnnx::Operand* mid = c1->OutputNodes()[0]->operand;
        if (!mid || output_tensor_nodes_.count(mid->name) || mid->consumers.empty()) continue;
        // every reader of `mid` is the same scheduled 1x1 conv or the sibling it computes
        Conv2d* pw = nullptr;
        bool ok = true;
        for (const pnnx::Operator* c : mid->consumers) {
            if (!c || c->type != "nn.Conv2d") { ok = false; break; }
            if (sibling_ops_.count(c->name)) continue;
            if (!index.count(c) || pw) { ok = false; break; }
            pw = dynamic_cast<Conv2d*>(order[index[c]].layer);
        }
        if (!ok || !pw) continue;
        for (const pnnx::Operator* c : mid->consumers)
            if (sibling_ops_.count(c->name) && (!pw->Sibling() || pw->Sibling()->GetOp() != c)) ok = false;
        if (!ok || !pw->CanFuseStemPairProducer(*c1)) continue;
        pw->SetStemPairProducer(c1);
        dead_operands_.insert(mid->name);
        removed[i] = true;
        fused_ops_.insert(order[i].op->name);
    }
    std::vector<Step> out;
    for (size_t i = 0; i < order.size(); ++i)
        if (!removed[i]) out.push_back(order[i]);
    order.swap(out);
    return Status::kSuccess;
}

// fp16 storage (round 5): conv A (1x1, c -> c, SiLU) whose ONLY consumer is conv B (3x3 s1 p1 over c channels that the slab kernel
// serves, SiLU, optional shortcut) -- the C3 bottleneck's pair -- becomes one launch at B's slot; A's output is never allocated.
Status EngineImpl::FuseBottleneckPairs(std::vector<Step>& order) {
    std::map<const pnnx::Operator*, size_t> index;
    for (size_t i = 0; i < order.size(); ++i) index[order[i].op] = i;
    std::vector<bool> removed(order.size(), false);
    for (size_t i = 0; i < order.size(); ++i) {
        Conv2d* pw = dynamic_cast<Conv2d*>(order[i].layer);
        if (!pw || order[i].op->type != "nn.Conv2d" || pw->InputNodes().size() != 1 || pw->OutputNodes().size() != 1) continue;
        const pnnx::Operand* mid = pw->OutputNodes()[0]->operand;
        if (!mid || output_tensor_nodes_.count(mid->name) || mid->consumers.size() != 1 || sibling_ops_.count(order[i].op->name)) continue;
        const pnnx::Operator* c = mid->consumers[0];
        if (!c || c->type != "nn.Conv2d" || !index.count(c) || sibling_ops_.count(c->name)) continue;
        Conv2d* conv = dynamic_cast<Conv2d*>(order[index[c]].layer);
        if (!conv || !conv->CanFusePointwiseProducer(*pw)) continue;
        conv->SetPointwiseProducer(pw);
        dead_operands_.insert(mid->name);
        removed[i] = true;
        fused_ops_.insert(order[i].op->name);
    }
    std::vector<Step> out;
    for (size_t i = 0; i < order.size(); ++i)
        if (!removed[i]) out.push_back(order[i]);
    order.swap(out);
    return Status::kSuccess;
}

// fp16 storage (round 6): torch.cat([y, z], channels) -> 1x1 conv (a YOLOv5 C3's closing cv3) where y is the output of a fused bottleneck pair
// (FuseBottleneckPairs) that only the concat reads: the pair, the concat and the conv become ONE launch at the conv's slot
// (si_hip_conv2d_pw_cv3_f16); y and the concat operand are never allocated, z keeps a buffer of its own (nothing aliases into a concat
// that no longer exists).
Status EngineImpl::FuseCv3IntoPairs(std::vector<Step>& order) {
    std::map<const pnnx::Operator*, size_t> index;
    for (size_t i = 0; i < order.size(); ++i) index[order[i].op] = i;
    std::vector<bool> removed(order.size(), false);
    for (size_t i = 0; i < order.size(); ++i) {
        Cat* cat = dynamic_cast<Cat*>(order[i].layer);
        if (!cat || order[i].op->type != "torch.cat" || cat->NhwcAxis() != 3 || order[i].op->inputs.size() != 2 || order[i].op->outputs.size() != 1) continue;
        const pnnx::Operand* y = order[i].op->inputs[0];
        const pnnx::Operand* z = order[i].op->inputs[1];
        const pnnx::Operand* cc = order[i].op->outputs[0];
        if (!y || !z || !cc || y == z || output_tensor_nodes_.count(y->name) || output_tensor_nodes_.count(cc->name)) continue;
        if (y->consumers.size() != 1 || cc->consumers.size() != 1) continue;
        // the scheduled step that WRITES y: after the epilogue fusion that is the conv whose output node y's is (y's pnnx producer is then the
        // folded SiLU / add operator, which no longer has a step)
        size_t pi = order.size();
        for (size_t j = 0; j < i && pi == order.size(); ++j)
            if (!removed[j])
                for (TensorNode* n : order[j].layer->OutputNodes())
                    if (n && n->operand == y) pi = j;
        if (pi == order.size()) continue;
        const pnnx::Operator* c3 = cc->consumers[0];
        if (!c3 || c3->type != "nn.Conv2d" || !index.count(c3) || sibling_ops_.count(c3->name)) continue;
        Conv2d* pair = dynamic_cast<Conv2d*>(order[pi].layer);
        Conv2d* conv = dynamic_cast<Conv2d*>(order[index[c3]].layer);
        if (!pair || !conv || !pair->PointwiseProducer() || pair->OutputNodes().size() != 1) continue;
        TensorNode* zn = tensor_nodes_[z->name];
        if (!conv->CanFuseCv3Pair(*pair, zn)) continue;
        conv->SetCv3Pair(pair, zn);
        dead_operands_.insert(y->name);
        dead_operands_.insert(cc->name);
        removed[pi] = true;
        removed[i] = true;
        fused_ops_.insert(order[pi].op->name);
        fused_ops_.insert(order[i].op->name);
    }
    std::vector<Step> out;
    for (size_t i = 0; i < order.size(); ++i)
        if (!removed[i]) out.push_back(order[i]);
    order.swap(out);
    return Status::kSuccess;
}

// fp16 storage: graph outputs keep the file's fp32 type.  Conv2d / Linear / Detect write fp32 from their own epilogue;
// any other producer fed by half operands writes a half staging operand instead, and a convert step follows it.
Status EngineImpl::InsertOutputCasts(std::vector<Step>& order) {
    for (auto& kv : output_tensor_nodes_) {
        TensorNode* out = kv.second;
        if (out->tensor.GetDataType() != DataType::kFloat32) continue;
        for (size_t i = 0; i < order.size(); ++i) {
            Layer* layer = order[i].layer;
            std::vector<TensorNode*> outs = layer->OutputNodes();
            auto slot = std::find(outs.begin(), outs.end(), out);
            if (slot == outs.end()) continue;
            if (dynamic_cast<Conv2d*>(layer) || dynamic_cast<Linear*>(layer) || dynamic_cast<YoloDetect*>(layer)) break;
            bool half_in = false;
            for (const TensorNode* in : layer->InputNodes()) half_in = half_in || in->tensor.GetDataType() == DataType::kFloat16;
            if (!half_in) break;

            const std::string staging_name = kv.first + "#f16";
            TensorNode* staging = new TensorNode;
            staging->operand = out->operand;
            staging->tensor = Tensor(DataType::kFloat16, out->tensor.Shape(), MemoryType::kDevice, false);
            tensor_nodes_[staging_name] = staging;
            *slot = staging;
            layer->SetOutputNodes(outs);

            OutputCast* cast = new OutputCast(order[i].op->name);
            layers_[cast->GetOp()->name] = cast;
            cast->SetContext(context_);
            cast->SetInputNodes({staging});
            cast->SetOutputNodes({out});
            CHECK_STATUS(cast->Validate());
            Step s;
            s.layer = cast;
            s.op = cast->GetOp();
            order.insert(order.begin() + i + 1, s);
            break;
        }
    }
    return Status::kSuccess;
}

// fp16 storage, a layer without an fp16 kernel: its half operands get fp32 SHADOW tensors -- a cast step in front of the layer
// for every half tensor it reads (inputs, a fused residual), one behind it for every half tensor it writes -- and the layer runs
// the kernel it has.  What the reference computes in fp32 (src/layer/conv_2d.cpp:94-101 rejects anything else) is then computed
// in fp32 here too; the neighbours keep their fp16 storage.
Status EngineImpl::InsertFp32Fallbacks(std::vector<Step>& order) {
    int serial = 0;
    for (size_t i = 0; i < order.size(); ++i) {
        Layer* layer = order[i].layer;
        std::string why;
        if (layer->HalfStorageOk(why)) continue;
        const std::string lname = order[i].op->name;
        auto refuse = [&](const std::string& more) {
            LOG(ERROR) << "fp16 storage: layer [" << lname << "] (" << order[i].op->type << ") cannot run: " << why << more
                       << "; load the model without the fp16 option";
            return Status::kUnsupport;
        };
        auto shadow_of = [&](TensorNode* n, const char* tag) {
            TensorNode* sh = new TensorNode;
            sh->operand = n->operand;
            sh->tensor = Tensor(DataType::kFloat32, n->tensor.Shape(), MemoryType::kDevice, false);
            tensor_nodes_[(n->operand ? n->operand->name : lname) + "#" + tag + std::to_string(serial++)] = sh;
            return sh;
        };
        std::vector<Step> before, after;
        auto cast_step = [&](TensorNode* from, TensorNode* to, const char* suffix, std::vector<Step>& where) -> Status {
            OutputCast* cast = new OutputCast(lname, (std::string(suffix) + std::to_string(serial++)).c_str());
            layers_[cast->GetOp()->name] = cast;
            cast->SetContext(context_);
            cast->SetInputNodes({from});
            cast->SetOutputNodes({to});
            CHECK_STATUS(cast->Validate());
            Step s;
            s.layer = cast;
            s.op = cast->GetOp();
            where.push_back(s);
            return Status::kSuccess;
        };
        std::map<TensorNode*, TensorNode*> in_shadow;
        std::vector<TensorNode*> ins = layer->InputNodes(), outs = layer->OutputNodes(), extra;
        layer->ExtraReads(extra);
        for (TensorNode*& n : ins) {
            if (n->tensor.GetDataType() != DataType::kFloat16) continue;
            if (!in_shadow.count(n)) {
                in_shadow[n] = shadow_of(n, "f32in");
                CHECK_STATUS(cast_step(n, in_shadow[n], ".in_to_f32.", before));
            }
            n = in_shadow[n];
        }
        for (TensorNode* n : extra) {
            if (n->tensor.GetDataType() != DataType::kFloat16) continue;
            if (!in_shadow.count(n)) {
                in_shadow[n] = shadow_of(n, "f32in");
                CHECK_STATUS(cast_step(n, in_shadow[n], ".in_to_f32.", before));
            }
            if (!layer->ReplaceExtraRead(n, in_shadow[n])) return refuse(" (and its fused extra operand cannot be rebound to an fp32 copy)");
        }
        for (TensorNode*& n : outs) {
            if (n->tensor.GetDataType() != DataType::kFloat16) continue;
            TensorNode* sh = shadow_of(n, "f32out");
            CHECK_STATUS(cast_step(sh, n, ".out_to_f16.", after));
            n = sh;
        }
        layer->SetInputNodes(ins);
        layer->SetOutputNodes(outs);
        std::string still;
        if (Status::kSuccess != layer->Validate() || !layer->HalfStorageOk(still)) return refuse(still.empty() ? "" : " / with fp32 operands: " + still);
        LOG(INFO) << "fp16 storage: layer [" << lname << "] has no fp16 kernel (" << why << "): it runs in fp32 between " << before.size()
                  << " + " << after.size() << " casts";
        order.insert(order.begin() + i + 1, after.begin(), after.end());
        order.insert(order.begin() + i, before.begin(), before.end());
        i += before.size() + after.size();
    }
    return Status::kSuccess;
}

// torch.cat on the channel axis: every eligible input operand becomes a view into the concat output
// (same pixel grid, pixel stride = total channels), so its producer writes in place and Cat::Forward
// finds nothing left to copy.
Status EngineImpl::AliasConcats() {
    for (const Step& s : plan_) {
        Cat* cat = dynamic_cast<Cat*>(s.layer);
        if (!cat || s.op->type != "torch.cat" || cat->NhwcAxis() != 3 || s.op->outputs.size() != 1) continue;
        TensorNode* out = cat->OutputNodes()[0];  // the staging operand when an output cast follows
        if (out->tensor.Shape().size() != 4) continue;
        int offset = 0;
        std::set<const pnnx::Operand*> seen;
        for (const pnnx::Operand* r : s.op->inputs) {
            const std::vector<int>& rs = tensor_nodes_[r->name]->tensor.Shape();
            const int c = rs.empty() ? 0 : rs.back();
            bool ok = rs.size() == 4 && !seen.count(r) && !aliases_.count(r->name) &&
                      !input_tensor_nodes_.count(r->name) && !output_tensor_nodes_.count(r->name) && r->producer &&
                      r->producer->type != "torch.cat" && HonoursPixelStride(r->producer->type) &&
                      tensor_nodes_[r->name]->tensor.GetDataType() == out->tensor.GetDataType() &&
                      (offset * ElementSize(out->tensor.GetDataType()) % 16 == 0);
            for (const pnnx::Operator* c2 : r->consumers) ok = ok && c2 && HonoursPixelStride(c2->type);
            // flatten writes dense NCHW and Detect writes rank-3 rows: they never feed a rank-4 cat
            if (ok && (r->producer->type == "torch.flatten" || r->producer->type == "models.yolo.Detect")) ok = false;
            seen.insert(r);
            if (ok) {
                Alias a;
                a.parent = out;
                a.channel_offset = offset;
                aliases_[r->name] = a;
            }
            offset += c;
        }
    }
    return Status::kSuccess;
}

// ---- memory --------------------------------------------------------------------------------------
Status EngineImpl::AllocateTensorMemory() {
    // 1. Every live operand that is not an alias gets HBM.  The reference mallocs each operand separately and never reuses
    //    (src/engine_impl.cpp:465-482, "TODO" at :466).  Here graph inputs / outputs keep their own buffers (they outlive the
    //    forward), and the intermediates are packed into ONE arena: two buffers may overlap in memory iff no launch of the plan
    //    needs both -- a buffer lives from the first step that writes it (or any alias into it) to the last step that reads it.
    struct Buf {
        TensorNode* node;
        size_t bytes, offset;
        int first, last;
    };
    std::map<TensorNode*, size_t> buf_of;   // root buffer node -> index in bufs
    std::vector<Buf> bufs;
    auto root_of = [&](TensorNode* n) -> TensorNode* {
        if (!n || !n->operand) return n;
        auto al = aliases_.find(n->operand->name);
        return al != aliases_.end() ? al->second.parent : n;
    };
    unshared_bytes_ = arena_bytes_ = 0;
    for (auto& kv : tensor_nodes_) {
        const std::string& name = kv.first;
        Tensor& t = kv.second->tensor;
        if (dead_operands_.count(name) || aliases_.count(name)) continue;
        if (!lanes_.empty() && !input_tensor_nodes_.count(name) && !output_tensor_nodes_.count(name)) continue;   // the lanes own the intermediates
        const size_t bytes = t.ByteSize();
        if (bytes == 0) {
            LOG(ERROR) << "operand [" << name << "] has no static shape";
            return Status::kErrorShape;
        }
        const bool io = input_tensor_nodes_.count(name) || output_tensor_nodes_.count(name);
        if (io || !opt_arena_) {
            void* p = nullptr;
            SI_TRY_HIP(si_hip_malloc(&p, bytes), "hipMalloc operand");
            device_allocs_.push_back(p);
            if (input_tensor_nodes_.count(name)) input_buffers_[name] = p;  // staging for host inputs
            if (output_tensor_nodes_.count(name)) own_output_ptrs_[name] = p;
            t.SetView(p, MemoryType::kDevice, 0);
            if (!io) unshared_bytes_ += bytes, arena_bytes_ += bytes;
            continue;
        }
        buf_of[kv.second] = bufs.size();
        bufs.push_back(Buf{kv.second, (bytes + 255) & ~size_t(255), 0, -1, -1});
        unshared_bytes_ += bytes;
    }
    if (!bufs.empty()) {
        auto touch = [&](TensorNode* n, int step) {
            auto it = buf_of.find(root_of(n));
            if (it == buf_of.end()) return;
            Buf& b = bufs[it->second];
            if (b.first < 0) b.first = step;
            b.last = step;
        };
        for (size_t i = 0; i < plan_.size(); ++i) {
            Layer* L = plan_[i].layer;
            std::vector<TensorNode*> extra;
            L->ExtraReads(extra);
            for (TensorNode* n : L->InputNodes()) touch(n, (int)i);
            for (TensorNode* n : extra) touch(n, (int)i);
            for (TensorNode* n : L->OutputNodes()) touch(n, (int)i);
        }
        // buffers no step of the plan touches (operands only a fused-away operator produced) live for the whole forward
        for (Buf& b : bufs)
            if (b.first < 0) { b.first = 0; b.last = (int)plan_.size(); }
        // greedy by size: place each buffer at the lowest offset that is free during its whole life
        std::vector<size_t> order(bufs.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return bufs[x].bytes != bufs[y].bytes ? bufs[x].bytes > bufs[y].bytes : x < y; });
        std::vector<size_t> placed;
        size_t total = 0;
        for (size_t bi : order) {
            Buf& b = bufs[bi];
            std::vector<std::pair<size_t, size_t>> busy;   // [offset, end) of placed buffers alive at the same time
            for (size_t pj : placed) {
                const Buf& o = bufs[pj];
                if (o.first <= b.last && b.first <= o.last) busy.push_back(std::make_pair(o.offset, o.offset + o.bytes));
            }
            std::sort(busy.begin(), busy.end());
            size_t off = 0;
            for (auto& iv : busy) {
                if (off + b.bytes <= iv.first) break;
                if (iv.second > off) off = iv.second;
            }
            b.offset = off;
            if (off + b.bytes > total) total = off + b.bytes;
            placed.push_back(bi);
        }
        arena_bytes_ = total;
        arena_plan_.clear();
        for (Buf& b : bufs) arena_plan_.push_back(ArenaSlot{b.node, b.offset});
        arena_plan_bytes_ = total;
        arena_pending_ = true;
        LOG(INFO) << "activation arena: " << total << " bytes for " << bufs.size() << " operands (" << unshared_bytes_ << " without sharing)";
    }
    // An engine whose forwards may all be served by the sliced host pipeline (host outputs, slicing not switched off, not a lane)
    // leaves the arena to the first forward that runs its OWN plan: the slicer has an arena of its own, and this one would be a
    // second, never-touched allocation.  Everyone else allocates now, so LoadModel reports an arena that does not fit.
    const bool defer = arena_pending_ && !is_lane_ && opt_outputs_to_host_ && opt_host_slices_ != 1 && lanes_.empty();
    if (!defer) CHECK_STATUS(EnsureArena());
    // 3. pinned host mirrors for outputs
    if (opt_outputs_to_host_) {
        for (auto& kv : output_tensor_nodes_) {
            void* h = nullptr;
            SI_TRY_HIP(si_hip_host_alloc(&h, kv.second->tensor.ByteSize()), "hipHostMalloc output");
            host_outputs_[kv.first] = h;
        }
    }
    return Status::kSuccess;
}

// allocate the planned arena (once) and point the operands -- and the concat aliases, which hang off them -- into it
Status EngineImpl::EnsureArena() {
    if (arena_pending_) {
        void* arena = nullptr;
        SI_TRY_HIP(si_hip_malloc(&arena, arena_plan_bytes_), "hipMalloc activation arena");
        device_allocs_.push_back(arena);
        for (const ArenaSlot& b : arena_plan_) b.node->tensor.SetView(static_cast<char*>(arena) + b.offset, MemoryType::kDevice, 0);
        arena_pending_ = false;
    }
    if (!aliases_bound_) {
        for (auto& kv : aliases_) {
            Tensor& t = tensor_nodes_[kv.first]->tensor;
            Tensor& parent = kv.second.parent->tensor;
            if (parent.RawData() == nullptr) continue;
            t.SetView(static_cast<char*>(parent.RawData()) + (size_t)kv.second.channel_offset * ElementSize(parent.GetDataType()),
                      MemoryType::kDevice, parent.Shape().back());
        }
        aliases_bound_ = true;
    }
    return Status::kSuccess;
}

Status EngineImpl::DeallocateTensorMemory() {
    arena_plan_.clear();
    arena_plan_bytes_ = 0;
    arena_pending_ = false;
    aliases_bound_ = false;
    for (void* p : device_allocs_) si_hip_free(p);
    device_allocs_.clear();
    input_buffers_.clear();
    own_output_ptrs_.clear();
    for (auto& kv : host_outputs_) si_hip_host_free(kv.second);
    host_outputs_.clear();
    for (auto& kv : tensor_nodes_) kv.second->tensor.SetView(nullptr, MemoryType::kDevice, 0);
    return Status::kSuccess;
}


// ---- lanes (option "streams") ---------------------------------------------------------------------
// Two half-batch engines on two streams instead of one full-batch engine on one: while one lane is in a layer's tail (the last,
// partial round of workgroups, the epilogue stores) the other lane's workgroups fill the idle CUs.  Measured on MI355X,
// YOLOv5s batch 32, same box, hipGraph replay: +4.9 % (tools/two_stream_exp.py).  Every operator of the
// path is per-image, so a lane is simply this model re-batched to N / 2 (the "batch" option's rule) that reads
// and writes slab views of this engine's input and output buffers; results are bit-identical to the one-stream schedule because
// no kernel's per-element arithmetic depends on the batch (tests/test_gpu_tiles.py, tests/test_gpu_engine.py).
// Can the batch be cut into contiguous slabs?  Every graph input and output must be batch-major (dimension 0 = the batch, rank >= 2);
// every operator of the path is per-image, so nothing else is needed.
bool EngineImpl::BatchSplittable(int& batch) const {
    batch = 0;
    bool ok = !input_tensor_nodes_.empty() && !output_tensor_nodes_.empty();
    for (auto& kv : input_tensor_nodes_) {
        const std::vector<int>& sh = kv.second->tensor.Shape();
        if (sh.size() < 2 || (batch != 0 && sh[0] != batch)) ok = false;
        else batch = sh[0];
    }
    for (auto& kv : output_tensor_nodes_) {
        const std::vector<int>& sh = kv.second->tensor.Shape();
        if (sh.size() < 2 || sh[0] != batch) ok = false;
    }
    return ok && batch >= 2;
}

Status EngineImpl::PlanLanes(int& lanes) {
    lanes = 1;
    if (is_lane_ || opt_streams_ < 2) return Status::kSuccess;
    int batch = 0;
    const bool ok = BatchSplittable(batch) && batch % 2 == 0;
    if (opt_streams_ >= 2) {
        if (!ok) {
            LOG(ERROR) << "streams=2 needs an even batch that is dimension 0 of every graph input and output";
            return Status::kUnsupport;
        }
        lanes = 2;
    }
    return Status::kSuccess;
}

Status EngineImpl::LoadLanes(int lanes) {
    int batch = input_tensor_nodes_.begin()->second->tensor.Shape()[0];
    for (int l = 0; l < lanes; ++l) {
        EngineImpl* lane = new EngineImpl;
        lanes_.push_back(lane);
        lane->is_lane_ = true;
        lane->opt_device_ = context_->device();
        lane->opt_fuse_ = opt_fuse_;
        lane->opt_alias_cat_ = opt_alias_cat_;
        lane->opt_fuse_upsample_ = opt_fuse_upsample_;
        lane->opt_fuse_stem_ = opt_fuse_stem_;
        lane->opt_fuse_pw_ = opt_fuse_pw_;
        lane->opt_arena_ = opt_arena_;
        lane->opt_winograd_ = opt_winograd_;
        lane->opt_f32_split_ = opt_f32_split_;
        lane->opt_f32_split_policy_ = opt_f32_split_policy_;
        lane->opt_plan_ = opt_plan_;
        lane->opt_plan_set_ = opt_plan_set_;
        lane->opt_detect_stream_ = opt_detect_stream_;
        lane->opt_detect_priority_ = opt_detect_priority_;
        lane->opt_fp16_ = opt_fp16_;
        lane->opt_graph_ = opt_graph_;        // every lane replays its OWN captured graph on its own stream; this engine only forks / joins
                                              // (one graph holding both branches measured 10 % SLOWER than one stream: profiles/r03_ab_streams.txt)
        lane->opt_outputs_to_host_ = false;   // ... and owns the host mirrors
        lane->opt_streams_ = 1;
        lane->opt_batch_ = batch / lanes;
        CHECK_STATUS(lane->LoadModel(param_path_, bin_path_));
        si_event_t ev = nullptr;
        SI_TRY_HIP(si_hip_event_create(&ev), "event create");
        lane_done_.push_back(ev);
    }
    if (!ev_fork_) SI_TRY_HIP(si_hip_event_create(&ev_fork_), "event create");
    LOG(INFO) << "streams: " << lanes << " lanes of batch " << batch / lanes;
    return Status::kSuccess;
}

Status EngineImpl::DestroyLanes() {
    for (EngineImpl* lane : lanes_) delete lane;
    lanes_.clear();
    for (si_event_t ev : lane_done_) si_hip_event_destroy(ev);
    lane_done_.clear();
    return Status::kSuccess;
}

namespace {
// lane `l` of `n`: the contiguous slab of images [l * N / n, (l + 1) * N / n) of a batch-major tensor, as a non-owning view
Tensor SlabView(const Tensor& t, int l, int n) {
    std::vector<int> shape = t.Shape();
    const size_t rows = t.NumElements() / (size_t)shape.back();   // pixels (rank 4) / rows of the last dimension
    const size_t slab_bytes = rows / (size_t)n * (size_t)t.PixelStride() * ElementSize(t.GetDataType());
    shape[0] /= n;
    Tensor v(t.GetDataType(), shape, MemoryType::kDevice, false);
    v.SetView(static_cast<char*>(t.RawData()) + (size_t)l * slab_bytes, MemoryType::kDevice, t.PixelStride() == t.Shape().back() ? 0 : t.PixelStride());
    return v;
}
}  // namespace

// hand every lane its slab of the (already resolved) input and output tensors of this engine
Status EngineImpl::BindLanes() {
    const int n = (int)lanes_.size();
    for (int l = 0; l < n; ++l) {
        for (auto& kv : input_tensor_nodes_) CHECK_STATUS(lanes_[l]->Input(kv.first, SlabView(kv.second->tensor, l, n)));
        for (auto& kv : output_tensor_nodes_) CHECK_STATUS(lanes_[l]->Output(kv.first, SlabView(kv.second->tensor, l, n)));
    }
    return Status::kSuccess;
}

// fork: every lane's stream waits for this engine's stream (the input upload); join: this stream waits for every lane.  With
// option "graph" every lane replays its own captured graph between the two.
Status EngineImpl::LaunchLanes() {
    si_stream_t stream = context_->stream();
    SI_TRY_HIP(si_hip_event_record(ev_fork_, stream), "event record");
    for (size_t l = 0; l < lanes_.size(); ++l) {
        SI_TRY_HIP(si_hip_stream_wait_event(lanes_[l]->context_->stream(), ev_fork_), "stream wait");
        CHECK_STATUS(lanes_[l]->ForwardAsync());
        lanes_[l]->forward_pending_ = false;   // (the lane's own start / stop events may sit inside a capture: never read)
        SI_TRY_HIP(si_hip_event_record(lane_done_[l], lanes_[l]->context_->stream()), "event record");
    }
    for (size_t l = 0; l < lanes_.size(); ++l) SI_TRY_HIP(si_hip_stream_wait_event(stream, lane_done_[l]), "stream wait");
    return Status::kSuccess;
}

// ---- the reference's host-tensor contract at speed (option "host_slices") -------------------------------------------------
// Engine::Input borrows a HOST tensor that is read at Forward() time and Extract hands back host-visible memory (reference
// src/engine_impl.cpp:522-555, bench/bench_yolo.cpp:20-28).  Done naively that is upload -> compute -> download in series:
// YOLOv5s batch 32 moves 157 MB up and 274 MB down around 4.3 ms of kernels, 12.8 ms per Forward.  Here one synchronous
// Forward() is a three-stage pipeline over G contiguous batch slices: slice g's images go up on an upload stream while slice
// g-1 computes (ONE child engine of batch N / G runs the slices back to back on its own stream: same weights, same arena) and
// slice g-2's output rows come down on a download stream into the pinned mirror Extract() returns.  What is left outside the
// overlap is the first slice's upload and the last slice's download.  A borrowed input buffer that is handed over for a second
// Forward is pinned in place (hipHostRegister) so its uploads run asynchronously at the link rate.  Bit-identical to the
// unsliced schedule (per-image arithmetic; tests/test_gpu_engine.py::test_host_tensor_pipeline_*).
int EngineImpl::PlanSlices() const {
    if (is_lane_ || opt_host_slices_ == 1 || !opt_outputs_to_host_ || slicer_failed_) return 1;
    if (!lanes_.empty()) return 1;   // "streams" = 2 was asked for: the lanes serve the engine
    int batch = 0;
    if (!BatchSplittable(batch)) return 1;
    for (auto& kv : input_tensor_nodes_) {
        auto u = user_inputs_.find(kv.first);
        if (u == user_inputs_.end() || u->second.RawData() == nullptr || u->second.GetMemoryType() == MemoryType::kDevice) return 1;
    }
    if (!user_outputs_.empty()) return 1;   // caller-owned device outputs: not the host contract
    if (opt_host_slices_ > 1) return batch % opt_host_slices_ == 0 ? opt_host_slices_ : 1;
    // auto re-batches the model to N / G: every built-in operator is per-image, a layer registered at run time
    // (RegisterLayer) may not be -- such a graph is sliced only on an explicit host_slices > 1
    if (has_user_layers_) return 1;
    // auto (MI355X, YOLOv5s 640x640, profiles/r03_host_slices.txt): slices of 8 images from batch 32 on (a batch-8 forward is still
    // at 0.6 of the MFMA ceiling, and 8 images are 39 MB up / 69 MB down -- 0.7 / 1.3 ms of PCIe outside the overlap), slices of 4
    // for batches 8 .. 31 (batch 8: 3.49 -> 2.95 ms per Forward with 2 slices; batch 16: 6.61 -> 4.95 with 4); at most 8 slices
    // (one captured graph per slice under "graph")
    const int per = batch >= 32 ? 8 : 4;
    if (batch < 8 || batch % per != 0) return 1;
    int g = batch / per;
    while (g > 8 && g % 2 == 0) g /= 2;
    return g <= 8 ? g : 1;
}

Status EngineImpl::DestroySlicer() {
    delete slicer_;
    slicer_ = nullptr;
    slices_ = 0;
    for (si_event_t e : ev_up_) si_hip_event_destroy(e);
    for (si_event_t e : ev_done_) si_hip_event_destroy(e);
    ev_up_.clear();
    ev_done_.clear();
    if (ev_down_all_) si_hip_event_destroy(ev_down_all_);
    ev_down_all_ = nullptr;
    delete up_context_;
    delete down_context_;
    up_context_ = down_context_ = nullptr;
    return Status::kSuccess;
}

void EngineImpl::UnpinInputs() {
    for (auto& kv : pinned_inputs_)
        if (kv.second.registered) si_hip_host_unregister(const_cast<void*>(kv.second.ptr));
    pinned_inputs_.clear();
}

// everything the sliced pipeline needs; slices_ is set LAST, so a failure half way leaves nothing that looks usable
Status EngineImpl::SetupSlicer(int slices) {
    CHECK_STATUS(DestroySlicer());
    int batch = 0;
    CHECK_BOOL(BatchSplittable(batch) && batch % slices == 0);
    slicer_ = new EngineImpl;
    slicer_->is_lane_ = true;
    slicer_->opt_device_ = context_->device();
    slicer_->opt_fuse_ = opt_fuse_;
    slicer_->opt_alias_cat_ = opt_alias_cat_;
    slicer_->opt_fuse_upsample_ = opt_fuse_upsample_;
    slicer_->opt_fuse_stem_ = opt_fuse_stem_;
    slicer_->opt_fuse_pw_ = opt_fuse_pw_;
    slicer_->opt_arena_ = opt_arena_;
    slicer_->opt_winograd_ = opt_winograd_;
    slicer_->opt_f32_split_ = opt_f32_split_;
    slicer_->opt_f32_split_policy_ = opt_f32_split_policy_;
    slicer_->opt_plan_ = opt_plan_;
    slicer_->opt_plan_set_ = opt_plan_set_;
    slicer_->opt_detect_stream_ = opt_detect_stream_;
    slicer_->opt_detect_priority_ = opt_detect_priority_;
    slicer_->opt_fp16_ = opt_fp16_;
    slicer_->opt_graph_ = opt_graph_;          // one captured graph per slice (its I/O pointers key the cache) ...
    slicer_->max_graphs_ = std::max<size_t>(max_graphs_, (size_t)slices);   // ... so the cache holds at least one per slice
    slicer_->opt_outputs_to_host_ = false;
    slicer_->opt_streams_ = 1;
    slicer_->opt_batch_ = batch / slices;
    CHECK_STATUS(slicer_->LoadModel(param_path_, bin_path_));
    up_context_ = new Context;
    CHECK_STATUS(up_context_->Init(context_->device()));
    if (debug_fail_slicer_) return Status::kFail;
    down_context_ = new Context;
    CHECK_STATUS(down_context_->Init(context_->device()));
    ev_up_.assign((size_t)slices, nullptr);
    ev_done_.assign((size_t)slices, nullptr);
    for (auto& e : ev_up_) SI_TRY_HIP(si_hip_event_create(&e), "event create");
    for (auto& e : ev_done_) SI_TRY_HIP(si_hip_event_create(&e), "event create");
    SI_TRY_HIP(si_hip_event_create(&ev_down_all_), "event create");
    if (!ev_fork_) SI_TRY_HIP(si_hip_event_create(&ev_fork_), "event create");
    slices_ = slices;
    LOG(INFO) << "host_slices: " << slices << " pipelined slices of batch " << batch / slices;
    return Status::kSuccess;
}

Status EngineImpl::ForwardSliced(int slices) {
    si_stream_t stream = context_->stream();
    CHECK_BOOL(slicer_ != nullptr && slices_ == slices);
    // the engine's device-side input staging and its own output buffers, whole batch; the slices are views
    for (auto& kv : input_tensor_nodes_) kv.second->tensor.SetView(input_buffers_[kv.first], MemoryType::kDevice, 0);
    CHECK_STATUS(BindOutputs());
    // a borrowed input seen for the second Forward in a row is pinned in place; a different pointer drops the old pin
    for (auto& kv : input_tensor_nodes_) {
        const Tensor& host = user_inputs_[kv.first];
        Pinned& p = pinned_inputs_[kv.first];
        if (p.ptr != host.RawData() || p.bytes != host.ByteSize()) {
            if (p.registered) si_hip_host_unregister(const_cast<void*>(p.ptr));
            p = Pinned();
            p.ptr = host.RawData();
            p.bytes = host.ByteSize();
        }
        // (opt-in, "pin_inputs": a registration outlives this call, so the caller has to keep the buffer mapped until the next
        // Input() / Release -- freeing it after Forward(), which the reference's borrow allows, would leave a stale registration)
        if (opt_pin_inputs_ && ++p.seen == 2 && !p.registered) {
            p.registered = si_hip_host_register(const_cast<void*>(p.ptr), p.bytes) == 0;   // (failure: the uploads stay staged copies)
            if (!p.registered) LOG(INFO) << "host_slices: could not pin the input buffer of [" << kv.first << "]; uploads stay staged";
        }
    }
    SI_TRY_HIP(si_hip_event_record(ev_start_, stream), "event record");
    // the upload stream starts behind whatever this engine's stream still holds (a previous ForwardAsync)
    SI_TRY_HIP(si_hip_event_record(ev_fork_, stream), "event record");
    SI_TRY_HIP(si_hip_stream_wait_event(up_context_->stream(), ev_fork_), "stream wait");
    SI_TRY_HIP(si_hip_stream_wait_event(slicer_->context_->stream(), ev_fork_), "stream wait");
    SI_TRY_HIP(si_hip_stream_wait_event(down_context_->stream(), ev_fork_), "stream wait");
    for (int g = 0; g < slices; ++g) {
        for (auto& kv : input_tensor_nodes_) {
            const Tensor dst = SlabView(kv.second->tensor, g, slices);
            const size_t bytes = dst.ByteSize();
            const char* src = static_cast<const char*>(user_inputs_[kv.first].RawData()) + (size_t)g * bytes;
            SI_TRY_HIP(si_hip_memcpy_h2d(dst.RawData(), src, bytes, up_context_->stream()), "input slice h2d");
            CHECK_STATUS(slicer_->Input(kv.first, dst));
        }
        SI_TRY_HIP(si_hip_event_record(ev_up_[(size_t)g], up_context_->stream()), "event record");
        SI_TRY_HIP(si_hip_stream_wait_event(slicer_->context_->stream(), ev_up_[(size_t)g]), "stream wait");
        for (auto& kv : output_tensor_nodes_) CHECK_STATUS(slicer_->Output(kv.first, SlabView(kv.second->tensor, g, slices)));
        CHECK_STATUS(slicer_->ForwardAsync());
        slicer_->forward_pending_ = false;
        SI_TRY_HIP(si_hip_event_record(ev_done_[(size_t)g], slicer_->context_->stream()), "event record");
        SI_TRY_HIP(si_hip_stream_wait_event(down_context_->stream(), ev_done_[(size_t)g]), "stream wait");
        for (auto& kv : output_tensor_nodes_) {
            const Tensor src = SlabView(kv.second->tensor, g, slices);
            const size_t bytes = src.ByteSize();
            SI_TRY_HIP(si_hip_memcpy_d2h(static_cast<char*>(host_outputs_[kv.first]) + (size_t)g * bytes, src.RawData(), bytes,
                                         down_context_->stream()), "output slice d2h");
        }
    }
    SI_TRY_HIP(si_hip_event_record(ev_down_all_, down_context_->stream()), "event record");
    SI_TRY_HIP(si_hip_stream_wait_event(stream, ev_down_all_), "stream wait");   // Sync() on this engine's stream covers the pipeline
    SI_TRY_HIP(si_hip_event_record(ev_stop_, stream), "event record");
    forward_pending_ = true;
    ++forward_count_;
    return Status::kSuccess;
}

// ---- I/O -----------------------------------------------------------------------------------------
const std::vector<std::string> EngineImpl::InputNames() {
    std::vector<std::string> ret;
    for (auto& kv : input_tensor_nodes_) ret.push_back(kv.first);
    return ret;
}

const std::vector<std::string> EngineImpl::OutputNames() {
    std::vector<std::string> ret;
    for (auto& kv : output_tensor_nodes_) ret.push_back(kv.first);
    return ret;
}

Status EngineImpl::OperandShape(const std::string& name, std::vector<int>& shape) {
    auto it = tensor_nodes_.find(name);
    if (it == tensor_nodes_.end()) return Status::kFail;
    shape = it->second->tensor.Shape();
    return Status::kSuccess;
}

Status EngineImpl::Input(const std::string& name, const Tensor& input) {
    auto it = input_tensor_nodes_.find(name);
    if (it == input_tensor_nodes_.end()) {
        LOG(ERROR) << "tensor [" << name << "] is not an input tensor";
        return Status::kFail;
    }
    if (input.NumElements() != it->second->tensor.NumElements() || input.GetDataType() != it->second->tensor.GetDataType()) {
        LOG(ERROR) << "tensor [" << name << "] does not match the model's input shape / type";
        return Status::kErrorShape;
    }
    // a previously borrowed buffer that was pinned in place (host_slices) is released to its owner NOW: the caller may free it
    // as soon as this returns (reference contract: borrowed until the next Input() / Release, engine_impl.cpp:466-471,528)
    auto pin = pinned_inputs_.find(name);
    if (pin != pinned_inputs_.end() && pin->second.ptr != input.RawData()) {
        if (pin->second.registered) {
            if (up_context_) si_hip_stream_sync(up_context_->stream());
            si_hip_host_unregister(const_cast<void*>(pin->second.ptr));
        }
        pinned_inputs_.erase(pin);
    }
    user_inputs_[name] = input;  // alias: the caller keeps ownership (reference engine_impl.cpp:528)
    return Status::kSuccess;
}

// Output(): the mirror of a device-resident Input() -- the caller provides the HBM buffer an output operand is written
// to (e.g. one of two tensors that are handed to an asynchronous collective while the next Forward runs).  Borrowed
// until the next Output() / Release; a null tensor restores the engine's own buffer.
Status EngineImpl::Output(const std::string& name, const Tensor& output) {
    auto it = output_tensor_nodes_.find(name);
    if (it == output_tensor_nodes_.end()) {
        LOG(ERROR) << "tensor [" << name << "] is not an output tensor";
        return Status::kFail;
    }
    if (nullptr == output.RawData()) {
        user_outputs_.erase(name);
        return Status::kSuccess;
    }
    if (output.GetMemoryType() != MemoryType::kDevice) {
        LOG(ERROR) << "Output(" << name << "): a caller-owned output buffer must be device memory";
        return Status::kUnsupport;
    }
    if (output.NumElements() != it->second->tensor.NumElements() || output.GetDataType() != it->second->tensor.GetDataType()) {
        LOG(ERROR) << "tensor [" << name << "] does not match the model's output shape / type";
        return Status::kErrorShape;
    }
    user_outputs_[name] = output;
    return Status::kSuccess;
}

Status EngineImpl::BindOutputs() {
    for (auto& kv : output_tensor_nodes_) {
        auto u = user_outputs_.find(kv.first);
        void* want = u != user_outputs_.end() ? u->second.RawData() : own_output_ptrs_[kv.first];
        if (want && kv.second->tensor.RawData() != want) kv.second->tensor.SetView(want, MemoryType::kDevice, 0);
    }
    return Status::kSuccess;
}

Status EngineImpl::UploadInputs() {
    for (auto& kv : input_tensor_nodes_) {
        auto u = user_inputs_.find(kv.first);
        if (u == user_inputs_.end() || nullptr == u->second.RawData()) {
            LOG(ERROR) << "input [" << kv.first << "] was not provided";
            return Status::kEmpty;
        }
        Tensor& node_tensor = kv.second->tensor;
        if (u->second.GetMemoryType() == MemoryType::kDevice) {
            // device-resident input: read it in place
            node_tensor.SetView(u->second.RawData(), MemoryType::kDevice, u->second.PixelStride());
        } else {
            void* buf = input_buffers_[kv.first];
            node_tensor.SetView(buf, MemoryType::kDevice, 0);
            SI_TRY_HIP(si_hip_memcpy_h2d(buf, u->second.RawData(), node_tensor.ByteSize(), context_->stream()), "input h2d");
        }
    }
    return Status::kSuccess;
}

Status EngineImpl::LaunchAll() {
    if (!lanes_.empty()) return LaunchLanes();
    YoloDetect* early_detect = nullptr;   // the Detect layer with levels in flight on the side stream
    for (size_t i = 0; i < plan_.size(); ++i) {
        const Step& s = plan_[i];
        Status ret = s.layer->Forward();
        if (Status::kSuccess != ret) {
            LOG(ERROR) << "layer [" << s.op->name << "] forward fail";
            return ret;
        }
        if (side_context_ && early_detect && s.layer == early_detect) {
            // join: everything after Detect (and the end of the forward) waits for the side stream's levels
            SI_TRY_HIP(si_hip_event_record(ev_join_, side_context_->stream()), "event record");
            SI_TRY_HIP(si_hip_stream_wait_event(context_->stream(), ev_join_), "stream wait");
            early_detect = nullptr;
        }
        if (side_context_ && !s.detect_levels.empty()) {
            // fork: the side stream may start once this step is done
            SI_TRY_HIP(si_hip_event_record(ev_fork_, context_->stream()), "event record");
            SI_TRY_HIP(si_hip_stream_wait_event(side_context_->stream(), ev_fork_), "stream wait");
            for (size_t j = i + 1; j < plan_.size() && !early_detect; ++j) {
                YoloDetect* d = dynamic_cast<YoloDetect*>(plan_[j].layer);
                if (d && d->EarlyLevels()) early_detect = d;
            }
            if (!early_detect) return Status::kFail;
            for (int level : s.detect_levels) CHECK_STATUS(early_detect->ForwardLevel(level));
        }
    }
    return Status::kSuccess;
}

Status EngineImpl::Forward() {
    CHECK_STATUS(ForwardAsync());
    return Sync();
}

// Forward() is synchronous by contract (reference engine_impl.cpp:533-544)
Status EngineImpl::Sync() {
    if (nullptr == context_) return Status::kFail;
    SI_TRY_HIP(si_hip_stream_sync(context_->stream()), "stream sync");
    // f32_split range guard: a split kernel that saw an operand leave fp16's range has set its layer's flag (the step's results are then
    // Inf / NaN in that layer's outputs).  The layer goes back to the true-fp32 kernels for this engine's lifetime and the step is re-run, here,
    // on the same inputs -- the caller of Forward() never sees the overflowed step.  (A caller that queued several ForwardAsync() steps gets
    // the LAST one re-run: the guard acts where the engine synchronises.)  At most one round per layer, bounded.
    if (forward_pending_ && !is_lane_) {
        for (int round = 0; round < 8 && TakeSplitTrips(); ++round) {
            ++split_reruns_;
            CHECK_STATUS(ForwardAsync());
            --forward_count_;
            SI_TRY_HIP(si_hip_stream_sync(context_->stream()), "stream sync");
        }
    }
    if (forward_pending_) {
        si_hip_event_elapsed_ms(ev_start_, ev_stop_, &last_forward_ms_);
        forward_pending_ = false;
    }
    return Status::kSuccess;
}

bool EngineImpl::TakeSplitTrips() {
    bool any = false;
    for (size_t i = 0; i < split_convs_.size(); ++i) {
        if (split_flags_ && reinterpret_cast<volatile unsigned*>(split_flags_)[i]) {   // (written by the device into pinned host memory)
            split_flags_[i] = 0u;
            Conv2d* cv = split_convs_[i];
            if (cv->f32_split_) {
                const pnnx::Operator* op = cv->GetOp();
                LOG(WARNING) << "f32_split: an operand of conv [" << (op ? op->name : std::string("(Detect level)")) << "] left fp16's range (|x| >= 65520, or a non-finite value); "
                             << "the layer runs on the true-fp32 kernels from here on and the step is re-run";
                cv->DemoteSplit();
                any = true;
            }
        }
    }
    for (EngineImpl* lane : lanes_) any = lane->TakeSplitTrips() || any;
    if (slicer_) any = slicer_->TakeSplitTrips() || any;
    if (any) {
        // captured graphs hold the split launches, and the demoted layers re-pack their weights at their next launch: eager once
        DestroyGraphCache();
        plan_warm_ = false;
    }
    return any;
}

std::vector<std::string> EngineImpl::SplitDemoted() const {
    std::vector<std::string> out;
    for (Conv2d* cv : split_convs_)
        if (cv->split_demoted_) out.push_back(cv->GetOp() ? cv->GetOp()->name : std::string("detect_level"));
    for (const EngineImpl* lane : lanes_)
        for (auto& n : lane->SplitDemoted()) out.push_back(n);
    if (slicer_)
        for (auto& n : slicer_->SplitDemoted()) out.push_back(n);
    return out;
}

Status EngineImpl::ForwardAsync() {
    if (nullptr == context_ || (plan_.empty() && lanes_.empty())) {
        LOG(ERROR) << "Forward before a successful LoadModel";
        return Status::kFail;
    }
    si_hip_set_device(context_->device());
    si_stream_t stream = context_->stream();
    if (const int slices = PlanSlices(); slices > 1) {
        if (slicer_ != nullptr && slices_ == slices) return ForwardSliced(slices);
        if (Status::kSuccess == SetupSlicer(slices)) return ForwardSliced(slices);
        // (a second arena that does not fit, an operator the re-batch rule cannot serve ...): the unsliced schedule below worked
        // before there was a pipeline and still does; remembered, so the setup is not retried on every Forward
        LOG(ERROR) << "host_slices: could not set up " << slices << " pipelined slices; serving this engine unsliced";
        DestroySlicer();
        slicer_failed_ = true;
    }
    CHECK_STATUS(EnsureArena());
    CHECK_STATUS(UploadInputs());
    CHECK_STATUS(BindOutputs());
    if (!lanes_.empty()) CHECK_STATUS(BindLanes());

    SI_TRY_HIP(si_hip_event_record(ev_start_, stream), "event record");
    if (opt_graph_ && plan_warm_ && lanes_.empty()) {
        // a captured graph bakes pointers in: key the cache on where the device-resident inputs and the outputs are now
        std::vector<void*> key;
        for (auto& kv : input_tensor_nodes_) key.push_back(kv.second->tensor.RawData());
        for (auto& kv : output_tensor_nodes_) key.push_back(kv.second->tensor.RawData());
        si_graph_t exec = nullptr;
        for (size_t i = 0; i < graph_cache_.size(); ++i) {
            if (graph_cache_[i].first == key) {
                exec = graph_cache_[i].second;
                if (i != 0) std::rotate(graph_cache_.begin(), graph_cache_.begin() + i, graph_cache_.begin() + i + 1);  // most recent first
                break;
            }
        }
        if (!exec) {
            SI_TRY_HIP(si_hip_graph_begin_capture(stream), "begin capture");
            Status ret = LaunchAll();
            const int rc = si_hip_graph_end_capture(stream, &exec);
            if (Status::kSuccess != ret || rc != 0) {
                // a launch that could not be captured (a layer fell back to another kernel family at this view and had to re-pack and upload
                // its weights, which a capture forbids -- ADVICE r05): the capture is discarded and this step runs eagerly; the next one captures
                if (exec) si_hip_graph_destroy(exec);
                LOG(WARNING) << "graph: the capture of this step failed; running it eagerly";
                CHECK_STATUS(LaunchAll());
                SI_TRY_HIP(si_hip_event_record(ev_stop_, stream), "event record");
                if (opt_outputs_to_host_) {
                    for (auto& kv : output_tensor_nodes_) {
                        const Tensor& t = kv.second->tensor;
                        SI_TRY_HIP(si_hip_memcpy_d2h(host_outputs_[kv.first], t.RawData(), t.ByteSize(), stream), "output d2h");
                    }
                }
                forward_pending_ = true;
                ++forward_count_;
                return Status::kSuccess;
            }
            if (graph_cache_.size() >= max_graphs_) {
                // the evicted exec may be the one a not-yet-synchronised ForwardAsync() launched: let the stream drain first
                SI_TRY_HIP(si_hip_stream_sync(stream), "stream sync");
                si_hip_graph_destroy(graph_cache_.back().second);
                graph_cache_.pop_back();
            }
            graph_cache_.insert(graph_cache_.begin(), std::make_pair(key, exec));
        }
        SI_TRY_HIP(si_hip_graph_launch(exec, stream), "graph launch");
    } else {
        // the first pass over this engine's own plan is always eager: it uploads weights and sizes scratch buffers (a Forward
        // served by the sliced host pipeline does not count: that ran the slicer's plan)
        CHECK_STATUS(LaunchAll());
        plan_warm_ = true;
    }
    SI_TRY_HIP(si_hip_event_record(ev_stop_, stream), "event record");

    if (opt_outputs_to_host_) {
        for (auto& kv : output_tensor_nodes_) {
            const Tensor& t = kv.second->tensor;
            SI_TRY_HIP(si_hip_memcpy_d2h(host_outputs_[kv.first], t.RawData(), t.ByteSize(), stream), "output d2h");
        }
    }
    forward_pending_ = true;
    ++forward_count_;
    return Status::kSuccess;
}

Status EngineImpl::Extract(const std::string& name, Tensor& output) {
    auto it = output_tensor_nodes_.find(name);
    if (it == output_tensor_nodes_.end()) {
        LOG(ERROR) << "tensor [" << name << "] is not an output tensor";
        return Status::kFail;
    }
    const Tensor& t = it->second->tensor;
    if (opt_outputs_to_host_) {
        output = Tensor(t.GetDataType(), t.Shape(), MemoryType::kHost, false);
        output.SetData(host_outputs_[name], MemoryType::kHost);
    } else {
        output = t;  // non-owning device view
    }
    return Status::kSuccess;
}

Status EngineImpl::Profile(std::vector<LayerProfile>& layers) {
    layers.clear();
    if (nullptr == context_ || (plan_.empty() && lanes_.empty())) return Status::kFail;
    si_stream_t stream = context_->stream();
    CHECK_STATUS(EnsureArena());
    CHECK_STATUS(UploadInputs());
    CHECK_STATUS(BindOutputs());
    if (!lanes_.empty()) {
        // per-launch durations are measured WITHOUT the overlap: one lane after the other, each layer between two events (as
        // Detect's side-stream levels are, below).  A layer appears once per lane, with that lane's FLOPs and bytes.
        CHECK_STATUS(BindLanes());
        SI_TRY_HIP(si_hip_stream_sync(stream), "stream sync");
        for (EngineImpl* lane : lanes_) {
            std::vector<LayerProfile> part;
            CHECK_STATUS(lane->Profile(part));
            layers.insert(layers.end(), part.begin(), part.end());
        }
        return Status::kSuccess;
    }
    std::vector<si_event_t> ev(plan_.size() + 1, nullptr);
    for (auto& e : ev) SI_TRY_HIP(si_hip_event_create(&e), "event create");
    Status ret = Status::kSuccess;
    si_hip_event_record(ev[0], stream);
    // (per-layer timing: Detect runs whole on the main stream here, whatever "detect_stream" says)
    std::vector<std::pair<YoloDetect*, unsigned>> early;
    for (const Step& st : plan_)
        if (YoloDetect* d = dynamic_cast<YoloDetect*>(st.layer))
            if (d->EarlyLevels()) { early.push_back(std::make_pair(d, d->EarlyLevels())); d->SetEarlyLevels(nullptr, 0); }
    for (size_t i = 0; i < plan_.size() && ret == Status::kSuccess; ++i) {
        ret = plan_[i].layer->Forward();
        si_hip_event_record(ev[i + 1], stream);
    }
    si_hip_stream_sync(stream);
    for (auto& e : early) e.first->SetEarlyLevels(side_context_, e.second);
    if (ret == Status::kSuccess) {
        for (size_t i = 0; i < plan_.size(); ++i) {
            LayerProfile p;
            p.name = plan_[i].op->name;
            p.type = plan_[i].op->type;
            p.kernel = plan_[i].layer->KernelName();
            si_hip_event_elapsed_ms(ev[i], ev[i + 1], &p.ms);
            p.flops = plan_[i].layer->Flops();
            p.bytes = plan_[i].layer->Bytes();
            layers.push_back(p);
        }
    }
    for (auto& e : ev) si_hip_event_destroy(e);
    return ret;
}

void* EngineImpl::Stream() { return context_ ? context_->stream() : nullptr; }

std::vector<std::string> EngineImpl::ScheduledOps() const {
    if (!lanes_.empty()) return lanes_[0]->ScheduledOps();
    std::vector<std::string> out;
    for (const Step& s : plan_) out.push_back(s.op->name);
    return out;
}

std::vector<std::string> EngineImpl::FusedOps() const {
    if (!lanes_.empty()) return lanes_[0]->FusedOps();
    std::vector<std::string> out(fused_ops_.begin(), fused_ops_.end());
    out.insert(out.end(), sibling_ops_.begin(), sibling_ops_.end());
    return out;
}

std::vector<std::string> EngineImpl::AliasedOperands() const {
    if (!lanes_.empty()) return lanes_[0]->AliasedOperands();
    std::vector<std::string> out;
    for (auto& kv : aliases_) out.push_back(kv.first);
    return out;
}

}  // namespace SimpleInfer
