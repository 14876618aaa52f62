// engine_impl.h -- the runtime behind SimpleInfer::Engine.
//
// LoadModel keeps the reference's stages (src/engine_impl.cpp:16-75): context -> graph (+ expression
// lowering) -> tensor nodes (NCHW file shapes become NHWC, :182-189) -> layers through the registry
// (:232-310) -> schedule -> memory.  What changes is everything below the Layer interface:
//   * the CGraph thread-pool pipeline (:336-437) becomes a topologically ordered launch list on one
//     HIP stream, optionally captured once and replayed as a hipGraph;
//   * per-operand malloc (:465-482) becomes HBM buffers, with torch.cat inputs aliased into the
//     concat buffer (producers write their channel slice in place);
//   * conv -> activation -> residual-add -> activation chains are folded into the conv kernel's
//     epilogue, so those layers launch nothing.
#ifndef SIMPLE_INFER_SRC_ENGINE_IMPL_H_
#define SIMPLE_INFER_SRC_ENGINE_IMPL_H_

#include <map>
#include <set>
#include <string>
#include <vector>

#include "context.h"
#include "engine.h"
#include "layer.h"
#include "pnnx/pnnx_helper.h"
#include "tensor.h"
#include "si_hip.h"
#include "tensor_node.h"
#include "types.h"

namespace SimpleInfer {

class Conv2d;

class EngineImpl {
public:
    EngineImpl();
    ~EngineImpl();

public:
    Status LoadModel(const std::string& parampath, const std::string& binpath);
    Status Release();

    Status CreateContext();
    Status DestroyContext();

    Status CreateGraph(const std::string& parampath, const std::string& binpath);
    Status DestroyGraph();

    Status CreateTensorNodes();
    Status DestroyTensorNodes();

    Status CreateLayers();
    Status DestroyLayers();

    Status CreatePipeline();   // schedule + fusion + concat aliasing
    Status DestroyPipeline();

    Status AllocateTensorMemory();
    Status DeallocateTensorMemory();

public:
    const std::vector<std::string> InputNames();
    const std::vector<std::string> OutputNames();

    Status Input(const std::string& name, const Tensor& input);
    Status Output(const std::string& name, const Tensor& output);
    Status Forward();
    // extension: enqueue the forward on the engine's stream and return; Sync() waits for it (Forward() = both).  Lets a
    // caller keep several engines (e.g. half-batches, or ranks of a node) busy from one host thread.
    Status ForwardAsync();
    Status Sync();
    Status Extract(const std::string& name, Tensor& output);

    Status SetOption(const std::string& key, int value);
    Status OperandShape(const std::string& name, std::vector<int>& shape);
    Status Profile(std::vector<LayerProfile>& layers);
    void* Stream();
    float LastForwardMs() const { return last_forward_ms_; }

    // schedule introspection (tests): operator names in launch order, and the fused-away ones
    std::vector<std::string> ScheduledOps() const;
    std::vector<std::string> FusedOps() const;
    std::vector<std::string> AliasedOperands() const;
    // bytes of HBM held for intermediate operands: with lifetime sharing, and what one allocation per operand would take
    void ActivationFootprint(size_t& arena_bytes, size_t& per_operand_bytes) const {
        arena_bytes = arena_bytes_;
        per_operand_bytes = unshared_bytes_;
        for (const EngineImpl* lane : lanes_) arena_bytes += lane->arena_bytes_, per_operand_bytes += lane->unshared_bytes_;
    }
    int Lanes() const { return lanes_.empty() ? 1 : (int)lanes_.size(); }

private:
    struct Step {
        Layer* layer = nullptr;
        const pnnx::Operator* op = nullptr;
        std::vector<int> detect_levels;   // Detect levels whose input this step completes (option "detect_stream")
    };
    Status PlanDetectStream();

    Status FuseEpilogues(std::vector<Step>& order);
    Status FuseSiblingConvs(std::vector<Step>& order);
    Status FusePoolChains(std::vector<Step>& order);
    Status FuseUpsampleIntoConvs(std::vector<Step>& order);
    Status FuseStemPairs(std::vector<Step>& order);
    Status FuseStemTriples(std::vector<Step>& order);
    Status FuseCv3IntoPairs(std::vector<Step>& order);
    Status FuseBottleneckPairs(std::vector<Step>& order);
    Status InsertOutputCasts(std::vector<Step>& order);
    Status InsertFp32Fallbacks(std::vector<Step>& order);
    Status AliasConcats();
    Status UploadInputs();
    Status BindOutputs();
    Status LaunchAll();

    // option "streams" = 2: the batch runs as two half-batch LANES on two streams (see LoadLanes)
    Status PlanLanes(int& lanes);
    Status LoadLanes(int lanes);
    Status BindLanes();
    Status LaunchLanes();
    Status DestroyLanes();
    Status DestroyGraphCache();

    // host-tensor contract at speed (option "host_slices"): see ForwardSliced
    bool BatchSplittable(int& batch) const;
    int PlanSlices() const;
    Status ForwardSliced(int slices);
    Status SetupSlicer(int slices);
    Status DestroySlicer();
    Status EnsureArena();
    void UnpinInputs();

private:
    // options
    int opt_device_ = -1;
    bool opt_fuse_ = true;
    bool opt_alias_cat_ = true;
    bool opt_fuse_upsample_ = true;
    int opt_fuse_stem_ = 2;
    int opt_fuse_pw_ = 1;   // (2: ... and the C3's closing conv behind its last 64-channel pair -- built in round 6, bit-identical, 0.74-0.84x of the launches
                            // it replaces: the fused workgroup is bound by its three SiLU epilogues on two waves per SIMD; opt-in, LAB_NOTEBOOK R6.4)
    bool opt_f32_split_ = false;
    int opt_f32_split_policy_ = 4;
    SiConvPlan opt_plan_ = SI_CONV_PLAN_DEFAULT;   // kernel-form choices handed to every conv launch (options f32_tile, f16_slab, ...; all default: the policy)
    bool opt_plan_set_ = false;
    bool opt_arena_ = true;      // intermediate operands share one HBM arena by lifetime (0: one allocation per operand, as the reference)
    bool opt_graph_ = false;
    bool opt_outputs_to_host_ = true;
    int opt_winograd_ = 1;
    int opt_batch_ = 0;          // > 0: serve this batch whatever batch the file was traced with
    int opt_detect_stream_ = 1;        // Detect's early levels on a second stream beside the layers that follow their inputs: 0 never, 1 for levels with enough work, 2 always
    int opt_detect_priority_ = 0;   // si_hip_stream_create_priority level of that stream (engine option detect_priority): the device default.  -1 (lowest: the
                                    // neck on the main stream is the critical path, Detect fills what it leaves) measured +0.7 % fp16 / flat fp32 on ONE engine
                                    // (r05_ab_detect_priority.txt) -- inside the box-to-box spread, and with several engines on a device a lowest-priority
                                    // stream can be starved by the others' default-priority work: opt-in since round 6
    bool opt_fp16_ = false;      // fp16 storage for internal activations and weights (BASELINE.json configs[3])
    int opt_streams_ = 1;        // 2: two half-batch lanes on two streams; 1 (default): one stream
    int opt_host_slices_ = 0;    // host inputs + host outputs: G batch slices pipelined over PCIe inside one Forward(); 1: off; 0 (default): auto
    bool opt_pin_inputs_ = false; // the sliced pipeline may hipHostRegister a borrowed input buffer in place (opt-in: the caller keeps it mapped until the next Input() / Release)

    Context* context_ = nullptr;
    Context* side_context_ = nullptr;        // second stream (option "detect_stream"); created with the first plan that uses it
    si_event_t ev_fork_ = nullptr, ev_join_ = nullptr;
    bool in_profile_ = false;                // Profile() times layers one by one on the main stream: no side launches
    pnnx::Graph* graph_ = nullptr;

    std::map<std::string, Layer*> layers_;
    std::map<std::string, TensorNode*> tensor_nodes_;
    std::map<std::string, TensorNode*> input_tensor_nodes_;
    std::map<std::string, TensorNode*> output_tensor_nodes_;

    std::vector<Step> plan_;

    // lanes: child engines serving one contiguous slab of the batch each, on their own streams, reading / writing slab views of
    // THIS engine's input and output buffers; this engine then has no layers of its own
    std::vector<EngineImpl*> lanes_;
    std::vector<si_event_t> lane_done_;
    bool is_lane_ = false;
    std::string param_path_, bin_path_;

    // the sliced host pipeline: ONE child engine of batch N / slices_ that runs the slices one after the other, an upload and a
    // download stream beside it
    EngineImpl* slicer_ = nullptr;
    int slices_ = 0;                         // set only once every resource of the pipeline exists
    bool debug_fail_slicer_ = false;         // option "_fail_slicer" (tests)
    bool slicer_failed_ = false;             // setting the pipeline up failed once: serve unsliced from then on
    bool has_user_layers_ = false;           // a layer type registered at run time (RegisterLayer): nothing is known about its batch semantics
    size_t max_graphs_ = 8;                  // captured hipGraphs kept (one per distinct set of I/O pointers)
    // the activation arena is planned at LoadModel but allocated at the first forward that runs THIS engine's plan (an engine
    // served by the sliced host pipeline never needs it)
    struct ArenaSlot { TensorNode* node; size_t offset; };
    std::vector<ArenaSlot> arena_plan_;
    size_t arena_plan_bytes_ = 0;
    bool arena_pending_ = false;
    bool aliases_bound_ = false;
    Context* up_context_ = nullptr;
    Context* down_context_ = nullptr;
    std::vector<si_event_t> ev_up_, ev_done_;
    si_event_t ev_down_all_ = nullptr;
    struct Pinned { const void* ptr = nullptr; size_t bytes = 0; int seen = 0; bool registered = false; };
    std::map<std::string, Pinned> pinned_inputs_;   // borrowed host inputs that were pinned in place (hipHostRegister)
    std::set<std::string> fused_ops_;        // operator names folded into a conv epilogue
    std::set<std::string> sibling_ops_;      // convs computed by a sibling conv's launch
    std::set<std::string> dead_operands_;    // operands that no longer exist after fusion
    struct Alias {
        TensorNode* parent = nullptr;
        int channel_offset = 0;
    };
    std::map<std::string, Alias> aliases_;   // operand name -> slice of a concat buffer

    std::vector<void*> device_allocs_;
    size_t arena_bytes_ = 0, unshared_bytes_ = 0;   // footprint with / without lifetime sharing (statistics)
    std::map<std::string, Tensor> user_inputs_;     // what Input() bound (host or device alias)
    std::map<std::string, Tensor> user_outputs_;    // what Output() bound (caller-owned device buffers)
    std::map<std::string, void*> own_output_ptrs_;  // the engine's own buffer of every output operand
    std::map<std::string, void*> input_buffers_;    // engine-owned device staging per input
    std::map<std::string, void*> host_outputs_;     // pinned mirrors per output

    // one captured graph per distinct set of I/O pointers (a capture bakes them in): alternating caller-owned output
    // buffers (ShardedEngine, OverlappedGather) replay instead of re-capturing every Forward.  Small LRU.
    std::vector<std::pair<std::vector<void*>, si_graph_t>> graph_cache_;
    int forward_count_ = 0;
    bool plan_warm_ = false;      // this engine's own plan has run eagerly once (weights uploaded): captures are allowed
    bool forward_pending_ = false;

    si_event_t ev_start_ = nullptr, ev_stop_ = nullptr;
    float last_forward_ms_ = 0.f;

    // f32_split range guard (round 6): one word of pinned host memory per conv that may run on the split kernels; a kernel writes 1 there when
    // an operand left fp16's range (include/si_hip.h).  Read at Sync(): a tripped layer goes back to the true-fp32 kernels for this engine's
    // lifetime and the step is re-run in place.
    unsigned* split_flags_ = nullptr;
    std::vector<Conv2d*> split_convs_;       // index = flag index
    int split_reruns_ = 0;                   // steps re-run because a flag tripped (statistics: Schedule())
    bool TakeSplitTrips();                   // demote the convs whose flag is set (here, in the lanes, in the slicer); true when any was
public:
    int SplitReruns() const { return split_reruns_; }
    std::vector<std::string> SplitDemoted() const;   // names of the convs the guard (or a weight out of range) sent back to true fp32
private:
};

}  // namespace SimpleInfer

#endif
