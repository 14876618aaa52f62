// c_api.cpp -- implementation of include/si_engine.h (extern "C" over SimpleInfer::Engine).
#include "si_engine.h"

#include <exception>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "engine.h"
#include "engine_impl.h"
#include "layer_registry.h"
#include "pnnx/expand_expression.h"
#include "pnnx/ir.h"

using namespace SimpleInfer;

struct SiEngine {
    EngineImpl impl;
    std::vector<std::string> in_names, out_names;
    std::vector<LayerProfile> profile;
};

namespace {

int code(Status s) { return static_cast<int>(s); }

void refresh_names(SiEngine* e) {
    e->in_names = e->impl.InputNames();
    e->out_names = e->impl.OutputNames();
}

int copy_out(const std::string& s, char* buf, size_t cap) {
    if (!buf || cap == 0) return (int)s.size();
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(buf, s.data(), n);
    buf[n] = 0;
    return (int)s.size();
}

}  // namespace

extern "C" {

int si_engine_create(SiEngine** engine) {
    if (!engine) return code(Status::kFail);
    InitializeContext();
    *engine = new SiEngine;
    return 0;
}

int si_engine_destroy(SiEngine* engine) {
    delete engine;
    return 0;
}

int si_engine_set_option(SiEngine* e, const char* key, int value) {
    if (!e || !key) return code(Status::kFail);
    return code(e->impl.SetOption(key, value));
}

int si_engine_load_model(SiEngine* e, const char* param_path, const char* bin_path) {
    if (!e || !param_path || !bin_path) return code(Status::kFail);
    const Status s = e->impl.LoadModel(param_path, bin_path);
    refresh_names(e);
    return code(s);
}

int si_engine_release(SiEngine* e) {
    if (!e) return code(Status::kFail);
    const Status s = e->impl.Release();
    refresh_names(e);
    return code(s);
}

int si_engine_num_inputs(SiEngine* e) { return e ? (int)e->in_names.size() : -1; }
int si_engine_num_outputs(SiEngine* e) { return e ? (int)e->out_names.size() : -1; }

const char* si_engine_input_name(SiEngine* e, int i) {
    return (e && i >= 0 && i < (int)e->in_names.size()) ? e->in_names[i].c_str() : nullptr;
}
const char* si_engine_output_name(SiEngine* e, int i) {
    return (e && i >= 0 && i < (int)e->out_names.size()) ? e->out_names[i].c_str() : nullptr;
}

int si_engine_operand_shape(SiEngine* e, const char* name, int* rank, int* dims) {
    if (!e || !name || !rank || !dims) return code(Status::kFail);
    std::vector<int> shape;
    const Status s = e->impl.OperandShape(name, shape);
    if (s != Status::kSuccess) return code(s);
    if (shape.size() > 8) return code(Status::kErrorShape);
    *rank = (int)shape.size();
    for (size_t i = 0; i < shape.size(); ++i) dims[i] = shape[i];
    return 0;
}

int si_engine_input(SiEngine* e, const char* name, const void* data, int on_device) {
    if (!e || !name || !data) return code(Status::kFail);
    std::vector<int> shape;
    if (e->impl.OperandShape(name, shape) != Status::kSuccess) return code(Status::kFail);
    Tensor t(DataType::kFloat32, shape, false);
    t.SetData(const_cast<void*>(data), on_device ? MemoryType::kDevice : MemoryType::kHost);
    return code(e->impl.Input(name, t));
}

int si_engine_bind_output(SiEngine* e, const char* name, void* device_data) {
    if (!e || !name) return code(Status::kFail);
    std::vector<int> shape;
    if (e->impl.OperandShape(name, shape) != Status::kSuccess) return code(Status::kFail);
    Tensor t(DataType::kFloat32, shape, false);
    if (device_data) t.SetData(device_data, MemoryType::kDevice);
    return code(e->impl.Output(name, t));
}

int si_engine_forward(SiEngine* e) { return e ? code(e->impl.Forward()) : code(Status::kFail); }
int si_engine_forward_async(SiEngine* e) { return e ? code(e->impl.ForwardAsync()) : code(Status::kFail); }
int si_engine_sync(SiEngine* e) { return e ? code(e->impl.Sync()) : code(Status::kFail); }

int si_engine_extract(SiEngine* e, const char* name, void** data, int* on_device) {
    if (!e || !name || !data) return code(Status::kFail);
    Tensor t;
    const Status s = e->impl.Extract(name, t);
    if (s != Status::kSuccess) return code(s);
    *data = t.RawData();
    if (on_device) *on_device = t.GetMemoryType() == MemoryType::kDevice ? 1 : 0;
    return 0;
}

void* si_engine_stream(SiEngine* e) { return e ? e->impl.Stream() : nullptr; }
float si_engine_last_forward_ms(SiEngine* e) { return e ? e->impl.LastForwardMs() : -1.f; }

int si_engine_profile(SiEngine* e) {
    if (!e) return -1;
    const Status s = e->impl.Profile(e->profile);
    return s == Status::kSuccess ? (int)e->profile.size() : -code(s);
}

int si_engine_profile_entry(SiEngine* e, int i, const char** op_name, const char** op_type, const char** kernel,
                            float* ms, double* flops, double* bytes) {
    if (!e || i < 0 || i >= (int)e->profile.size()) return code(Status::kFail);
    const LayerProfile& p = e->profile[i];
    if (op_name) *op_name = p.name.c_str();
    if (op_type) *op_type = p.type.c_str();
    if (kernel) *kernel = p.kernel.c_str();
    if (ms) *ms = p.ms;
    if (flops) *flops = p.flops;
    if (bytes) *bytes = p.bytes;
    return 0;
}

int si_engine_schedule(SiEngine* e, char* buf, size_t cap) {
    if (!e) return -1;
    std::ostringstream os;
    for (auto& n : e->impl.ScheduledOps()) os << "run " << n << "\n";
    for (auto& n : e->impl.FusedOps()) os << "fused " << n << "\n";
    for (auto& n : e->impl.AliasedOperands()) os << "alias " << n << "\n";
    size_t arena = 0, unshared = 0;
    e->impl.ActivationFootprint(arena, unshared);
    os << "arena_bytes " << arena << "\n" << "per_operand_bytes " << unshared << "\n";
    os << "lanes " << e->impl.Lanes() << "\n";   // 2: the batch runs as two half-batch lanes on two streams (option "streams")
    // f32_split range guard: steps re-run because an operand left fp16's range, and the convs that went back to the true-fp32 kernels
    os << "split_reruns " << e->impl.SplitReruns() << "\n";
    for (auto& n : e->impl.SplitDemoted()) os << "split_demoted " << n << "\n";
    return copy_out(os.str(), buf, cap);
}

int si_registry_types(char* buf, size_t cap) {
    std::string s;
    for (auto& t : RegisteredLayerTypes()) s += t + "\n";
    return copy_out(s, buf, cap);
}

int si_pnnx_save(const char* param_path, const char* bin_path, int expand, int batch, const char* out_param_path,
                 const char* out_bin_path) {
    if (!param_path || !bin_path || !out_param_path || !out_bin_path) return -1;
    pnnx::Graph g;
    try {
        if (g.load(param_path, bin_path) != 0) return 1;
        if (expand) pnnx::expand_expression(g);
        if (batch > 0) {
            int traced = 0;
            for (const pnnx::Operand* r : g.operands)
                if (r->producer && r->producer->inputs.empty() && !r->shape.empty()) {
                    if (traced != 0 && traced != r->shape[0]) return 5;
                    traced = r->shape[0];
                }
            if (traced <= 0) return 5;
            for (pnnx::Operand* r : g.operands)
                if (!r->shape.empty() && r->shape[0] == traced) r->shape[0] = batch;
        }
        return g.save(out_param_path, out_bin_path) == 0 ? 0 : 2;
    } catch (const std::exception&) {
        return 1;
    }
}

// canonical dump; line formats match oracle/ref_pnnx_dump.cpp (the driver around the reference loader)
int si_pnnx_dump(const char* param_path, const char* bin_path, int expand, const char* out_path) {
    if (!param_path || !bin_path || !out_path) return -1;
    pnnx::Graph g;
    try {
        if (g.load(param_path, bin_path) != 0) return 1;
        if (expand) pnnx::expand_expression(g);
    } catch (const std::exception&) {
        return 1;  // malformed file: the parser's std::stoi / map::at threw
    }
    FILE* f = fopen(out_path, "w");
    if (!f) return 2;
    auto fnv1a = [](const std::vector<char>& d) {
        uint64_t h = 1469598103934665603ull;
        for (char c : d) {
            h ^= (unsigned char)c;
            h *= 1099511628211ull;
        }
        return h;
    };
    fprintf(f, "ops %zu operands %zu\n", g.ops.size(), g.operands.size());
    for (auto* op : g.ops) {
        fprintf(f, "op %s %s in=", op->type.c_str(), op->name.c_str());
        for (auto* r : op->inputs) fprintf(f, "%s,", r->name.c_str());
        fprintf(f, " out=");
        for (auto* r : op->outputs) fprintf(f, "%s,", r->name.c_str());
        fprintf(f, "\n");
        for (auto& kv : op->params) {
            const pnnx::Parameter& p = kv.second;
            fprintf(f, "  param %s type=%d", kv.first.c_str(), p.type);
            switch (p.type) {
                case 1: fprintf(f, " b=%d", p.b ? 1 : 0); break;
                case 2: fprintf(f, " i=%d", p.i); break;
                case 3: fprintf(f, " f=%.9g", p.f); break;
                case 4: fprintf(f, " s=%s", p.s.c_str()); break;
                case 5: fprintf(f, " ai="); for (int v : p.ai) fprintf(f, "%d,", v); break;
                case 6: fprintf(f, " af="); for (float v : p.af) fprintf(f, "%.9g,", v); break;
                case 7: fprintf(f, " as="); for (auto& v : p.as) fprintf(f, "%s,", v.c_str()); break;
                default: break;
            }
            fprintf(f, "\n");
        }
        for (auto& kv : op->attrs) {
            fprintf(f, "  attr %s type=%d shape=", kv.first.c_str(), kv.second.type);
            for (int s : kv.second.shape) fprintf(f, "%d,", s);
            fprintf(f, " bytes=%zu fnv=%016llx\n", kv.second.data.size(), (unsigned long long)fnv1a(kv.second.data));
        }
    }
    for (auto* r : g.operands) {
        fprintf(f, "operand %s type=%d shape=", r->name.c_str(), r->type);
        for (int s : r->shape) fprintf(f, "%d,", s);
        fprintf(f, " producer=%s consumers=", r->producer ? r->producer->name.c_str() : "-");
        for (auto* c : r->consumers) fprintf(f, "%s,", c->name.c_str());
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}

}  // extern "C"
