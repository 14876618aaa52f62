/*
 * si_engine.h -- C-ABI over SimpleInfer::Engine (include/engine.h) for non-C++ hosts (ctypes, cgo,
 * JNI...).  Every call returns a SimpleInfer::Status code as int (0 = kSuccess, 1 kFail, 2 kEmpty,
 * 3 kErrorShape, 4 kErrorContext, 5 kUnsupport; reference include/types.h:24-31) unless noted.
 * Entry points mirror the reference's Engine methods one to one (include/engine.h:12-38), which is
 * also the surface its pybind11 module exposes (python/pybind11_main.cpp:13-68).
 */
#ifndef SI_ENGINE_H_
#define SI_ENGINE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SiEngine SiEngine;

/* Engine::Engine / ~Engine */
int si_engine_create(SiEngine** engine);
int si_engine_destroy(SiEngine* engine);
/* Engine::SetOption -- before load_model.  Keys: device, fuse, alias_cat, fuse_upsample, arena, winograd, fp16, batch, graph, outputs_to_host, streams, host_slices, detect_stream
 * (include/engine.h documents the values) */
int si_engine_set_option(SiEngine* engine, const char* key, int value);
/* Engine::LoadModel / Release (reference src/engine_impl.cpp:16-75, :77-127) */
int si_engine_load_model(SiEngine* engine, const char* param_path, const char* bin_path);
int si_engine_release(SiEngine* engine);
/* Engine::InputNames / OutputNames: counts, then names (pointer valid until the next call) */
int si_engine_num_inputs(SiEngine* engine);
int si_engine_num_outputs(SiEngine* engine);
const char* si_engine_input_name(SiEngine* engine, int index);
const char* si_engine_output_name(SiEngine* engine, int index);
/* NHWC shape of an input/output operand; dims must hold 8 ints */
int si_engine_operand_shape(SiEngine* engine, const char* name, int* rank, int* dims);
/* Engine::Input: borrows `data` (fp32, NHWC, the operand's element count) until the next Input/Release;
 * it is read at forward time.  on_device != 0: `data` is a device pointer, read in place. */
int si_engine_input(SiEngine* engine, const char* name, const void* data, int on_device);
/* Engine::Output (extension): write output operand `name` into caller-owned DEVICE memory of the operand's size from the
 * next forward on; NULL restores the engine's own buffer */
int si_engine_bind_output(SiEngine* engine, const char* name, void* device_data);
/* Engine::Forward: synchronous */
int si_engine_forward(SiEngine* engine);
/* Engine::ForwardAsync / Sync (extension): the two halves of forward */
int si_engine_forward_async(SiEngine* engine);
int si_engine_sync(SiEngine* engine);
/* Engine::Extract: non-owning view of engine memory (host pinned mirror, or device pointer when the
 * engine was created with outputs_to_host = 0) */
int si_engine_extract(SiEngine* engine, const char* name, void** data, int* on_device);
/* hipStream_t the engine launches on; HIP-event time of the last forward's kernels */
void* si_engine_stream(SiEngine* engine);
float si_engine_last_forward_ms(SiEngine* engine);
/* One instrumented forward with a HIP event after every layer.  Returns the number of scheduled
 * layers (<0 on error); entries are then read one by one. */
int si_engine_profile(SiEngine* engine);
int si_engine_profile_entry(SiEngine* engine, int index, const char** op_name, const char** op_type,
                            const char** kernel, float* ms, double* flops, double* bytes);
/* text description of the launch schedule: "run <op>", "fused <op>", "alias <operand>" lines, then "arena_bytes <n>" (HBM held
 * for intermediate operands, shared by lifetime) and "per_operand_bytes <n>" (what one allocation per operand would take) */
int si_engine_schedule(SiEngine* engine, char* buf, size_t cap);

/* Loader check: parse a .pnnx.param/.bin with THIS library's pnnx loader (optionally lowering
 * pnnx.Expression) and write a canonical text dump to out_path -- byte-comparable with the dump the
 * reference's own loader produces through oracle/_ref/ref_pnnx_dump.  Returns 0 on success. */
int si_pnnx_dump(const char* param_path, const char* bin_path, int expand, const char* out_path);

/* Model tooling (reference src/pnnx/ir.cpp:817-1008 Graph::save, src/pnnx/storezip.cpp:242-395):
 * load a model with this library's loader, optionally lower pnnx.Expression (expand != 0) and / or rewrite the traced
 * batch in every operand shape to `batch` (batch > 0; the same rule SetOption("batch") applies at load), and write it
 * back as .pnnx.param + stored-zip .pnnx.bin.  Returns 0 on success, 1 for an unreadable / malformed model,
 * 2 when the output cannot be written, 5 when the graph inputs carry no common static batch. */
int si_pnnx_save(const char* param_path, const char* bin_path, int expand, int batch, const char* out_param_path,
                 const char* out_bin_path);

/* registry introspection: newline-separated pnnx type strings */
int si_registry_types(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif

#endif /* SI_ENGINE_H_ */
