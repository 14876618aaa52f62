// engine.h -- SimpleInfer::Engine, the drop-in entry point (reference include/engine.h:12-38):
// LoadModel / Release / InputNames / OutputNames / Input / Forward / Extract with the same
// signatures, operand-name keys (std::map order) and ownership rules:
//   * Input() borrows the caller's buffer; it is read at Forward() time
//     (reference src/engine_impl.cpp:522-531), so call Input once, Forward many times.
//   * Forward() is synchronous: outputs are valid when it returns.
//   * Extract() hands out a non-owning view of engine memory, valid until the next Forward /
//     Release (reference src/engine_impl.cpp:546-555).
// Behind it: the pnnx graph is loaded, one Layer is created per operator through the
// LayerRegistry, weights are uploaded to HBM once, and Forward() launches hand-written gfx950
// HIP kernels in topological order on one stream (optionally replayed as a hipGraph).
//
// Extensions (no reference counterpart): SetOption(), device-resident Input / Extract, profiling.
#pragma once

#include <string>
#include <vector>

#include "tensor.h"
#include "types.h"

namespace SimpleInfer {

class EngineImpl;

struct LayerProfile {
    std::string name;   // operator name
    std::string type;   // pnnx type string
    std::string kernel; // kernel family that ran ("conv_igemm", "maxpool", ...)
    float ms = 0.f;     // HIP-event time of the layer's launches
    double flops = 0.0; // algorithmic (direct-convolution) flops, 0 for non-conv layers
    double bytes = 0.0; // algorithmic HBM bytes (inputs + outputs + weights, each once)
};

class Engine {
public:
    // ---- the reference's surface (include/engine.h:12-38), same names, argument meaning and Status codes ------------
    Engine();   // pimpl, no device work until LoadModel
    ~Engine();  // Release()s

    // parse .pnnx.param + stored-zip .pnnx.bin, build one Layer per operator through the registry, plan the launch
    // schedule, allocate HBM; implicitly Release()s a previously loaded model (src/engine_impl.cpp:16-75)
    Status LoadModel(const std::string& parampath, const std::string& binpath);
    Status Release();  // src/engine_impl.cpp:77-127

    const std::vector<std::string> InputNames();   // pnnx operand names, sorted (std::map order, :484-520)
    const std::vector<std::string> OutputNames();

    Status Input(const std::string& name, const Tensor& input);  // borrows; read at Forward() time (:522-531)
    Status Forward();                                             // synchronous (:533-544)
    Status Extract(const std::string& name, Tensor& output);     // non-owning view (:546-555)

    // Extension: have an output operand written into caller-owned device memory (MemoryType::kDevice, same element
    // count and type) from the next Forward() on; borrowed until the next Output() / Release().  A tensor without data
    // restores the engine's own buffer.  Lets a caller alternate output buffers, e.g. to hand one to an asynchronous
    // collective while the next Forward() fills the other.
    Status Output(const std::string& name, const Tensor& output);

public:
    // ---- extensions -------------------------------------------------------------------
    // Must be called before LoadModel.  Keys:
    //   "device"          HIP device ordinal (default: current device)
    //   "fuse"            1/0  fold activation / residual add into the conv epilogue (default 1)
    //   "alias_cat"       1/0  producers write straight into torch.cat outputs (default 1)
    //   "arena"           1/0  intermediate operands share one HBM arena by lifetime (default 1); 0 = one allocation per operand,
    //                          as the reference does (src/engine_impl.cpp:465-482)
    //   "fuse_upsample"   1/0  a 1x1 conv that consumes cat(upsample(x), skip) reads x at the source pixel; the upsample launch
    //                          and its output disappear (default 1; needs "fuse" and "alias_cat")
    //   "winograd"        0/1/2  3x3 stride-1 convs: 1 (default) fused Winograd F(2,3) where it is the faster kernel, as
    //                          the reference does on the CPU; 0 = implicit GEMM everywhere; 2 = fused Winograd F(4,3)
    //   "fp16"            1/0  fp16 storage for internal activations and weights, fp16 MFMA with fp32 accumulation;
    //                          Input / Extract tensors stay fp32 (default 0: the reference's fp32 arithmetic)
    //   "f32_split"       1/0  fp32 tensors everywhere, but the dense convs over multiples of 32 channels contract on the fp16 matrix
    //                          cores: every operand as two fp16 halves (22 significant bits), three exact fp16 products per fp32
    //                          product, fp32 accumulation (Ootomo & Yokota 2022; conv_split3.hip).  Measured closer to the float64
    //                          convolution than the fp32 MFMA chain and 1.8-2.1x faster on the K-heavy layers; another arithmetic
    //                          than the reference's fp32.  Default 0 -- the headline path is true fp32; ignored with "fp16".
    //                          Which layers: "f32_split_policy" 1..4 (4, the default: the K-heavy dense convs, sibling-fused and wide 1x1
    //                          layers, the layers that read upsample + concat at the source, the Winograd layers, the Detect levels and
    //                          the RGB stem; lower levels are the earlier rounds' sets, kept for A/B runs).
    //                          RANGE GUARD (round 6): the reference convolves any finite fp32; a value that rounds to fp16 infinity
    //                          (|x| >= 65520; in the Winograd layers a transformed value, i.e. a sum of four inputs) cannot be split.
    //                          Weights are checked at load (such a layer never leaves the fp32 kernels); activations by the split
    //                          kernels themselves, at no cost while nothing trips: Forward() / Sync() then puts the layer back on the
    //                          true-fp32 kernels for the engine's lifetime and RE-RUNS the step in place -- the caller never sees the
    //                          overflowed step, and gets what the default engine computes for that layer.  (ForwardAsync() steps queued
    //                          without a Sync() between them: the guard acts at the Sync(), on the last one.)  Tensors whose every
    //                          value is tiny (scale ~1e-6) keep 1e-5-class, not fp32-class, relative accuracy (fp16 subnormals).
    //   "f32_tile", "wino23_form", "wino23_ocg", "f16_tile", "f16_detect_tile", "f16_s2c32", "f16_slab", "f16_slab_w2", "f16_pw_patch"
    //                          kernel-form choices handed to every conv launch (SiConvPlan, include/si_hip.h): A/B runs and tests.  Every
    //                          form of a kernel family produces the same bits; the default leaves the choice to the launch-size policy.
    //   "batch"           N>0  serve batch N whatever batch the .param file was traced with (default 0: as in the file)
    //   "host_slices"     G    host tensors in (Input) and out (Extract) -- the reference's calling convention: one synchronous
    //                          Forward() pipelines G batch slices over an upload, a compute and a download stream; 1 = off,
    //                          0 (default) = slices of 8 images from batch 32 on, of 4 for batches 8 .. 31 (graphs of built-in
    //                          layers only: the model is loaded a second time re-batched to N / G, which assumes per-image
    //                          operators; a graph holding a RegisterLayer type is sliced only on an explicit G > 1).  If the
    //                          pipeline cannot be set up the engine logs it and serves unsliced.
    //   "pin_inputs"      1/0  host_slices only: a borrowed input buffer handed over for a second Forward() in a row is pinned IN
    //                          PLACE (hipHostRegister) so its uploads run at link rate.  Default 0, opt-in: the registration
    //                          outlives Forward(), so the caller must keep the buffer mapped -- not free or reallocate it --
    //                          until the next Input() for that name or Release(), which unregister it; and must not hand the
    //                          same buffer to a second engine meanwhile.  Without it the uploads are staged copies.
    //   "streams"         1/2  2: the batch runs as two half-batch lanes on two streams (default 1: no gain measured since the
    //                          tile policy follows the launch size); with 2 the lanes, not host_slices, serve host tensors
    //   "detect_stream"   0/1/2  YOLOv5 Detect's finer levels on a second stream: 0 never, 1 (default) for levels with enough work,
    //                          2 always
    //   "detect_priority" -1/0/1  priority of that second stream: 0 (default) the device's default; -1 the lowest (the neck on the main
    //                          stream is the critical path, Detect fills what it leaves: +0.7 % fp16 on one engine, but starvable when
    //                          several engines share a device -- opt-in since round 6); 1 the highest
    //   "graph"           1/0  replay Forward() as a captured hipGraph (default 0)
    //   "outputs_to_host" 1/0  copy outputs to pinned host memory in Forward() (default 1);
    //                          with 0, Extract() returns device tensors
    Status SetOption(const std::string& key, int value);

    // Forward() in two halves: ForwardAsync() enqueues the launches on the engine's stream and returns, Sync() waits for them
    // (and for the output copies).  One host thread can keep several engines busy this way.
    Status ForwardAsync();
    Status Sync();

    // shape (NHWC / as stored) of any input or output operand
    Status OperandShape(const std::string& name, std::vector<int>& shape);

    // run one instrumented forward: per-layer HIP-event timings on the engine's stream
    Status Profile(std::vector<LayerProfile>& layers);

    // HIP stream the engine launches on (hipStream_t as void*)
    void* Stream();

    // timing of the most recent Forward() measured with HIP events on the engine stream
    float LastForwardMs();

private:
    EngineImpl* impl_ = nullptr;
};

void InitializeContext();

}  // namespace SimpleInfer
