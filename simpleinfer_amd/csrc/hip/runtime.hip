// runtime.hip -- device / memory / stream / event / graph wrappers of the C-ABI (include/si_hip.h).
// These replace the host-side substrate of the reference: malloc'd tensors
// (src/tensor.cpp:47-97), the Eigen thread-pool context (src/context.cpp:9-26) and the CGraph
// pipeline run (src/engine_impl.cpp:533-544) become HBM buffers, a HIP stream and an optional
// captured hipGraph.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "si_hip.h"
#include "si_hip_internal.h"

extern "C" {

const char* si_hip_version(void) { return "simpleinfer_amd-hip 0.1 (gfx950)"; }

const char* si_hip_error_string(int code) {
    if (code == 0) return "success";
    if (code == SI_E_BADARG) return "bad argument";
    if (code == SI_E_UNSUPPORTED) return "unsupported configuration";
    if (code == SI_E_NODEVICE) return "no HIP device";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

int si_hip_device_count(int* count) {
    if (!count) return SI_E_BADARG;
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return (int)e;
    }
    return 0;
}

int si_hip_set_device(int device) { SI_HIP_TRY(hipSetDevice(device)); return 0; }
int si_hip_get_device(int* device) { if (!device) return SI_E_BADARG; SI_HIP_TRY(hipGetDevice(device)); return 0; }

int si_hip_device_info(int device, char* name, int* cus, size_t* hbm_bytes, int* clock_khz) {
    hipDeviceProp_t p;
    SI_HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name) { strncpy(name, p.name, 255); name[255] = 0; }
    if (cus) *cus = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    if (clock_khz) *clock_khz = p.clockRate;
    return 0;
}

int si_hip_malloc(void** ptr, size_t bytes) {
    if (!ptr) return SI_E_BADARG;
    *ptr = nullptr;
    if (bytes == 0) bytes = 16;
    SI_HIP_TRY(hipMalloc(ptr, bytes));
    return 0;
}
int si_hip_free(void* ptr) { if (ptr) SI_HIP_TRY(hipFree(ptr)); return 0; }
int si_hip_host_alloc(void** ptr, size_t bytes) {
    if (!ptr) return SI_E_BADARG;
    *ptr = nullptr;
    SI_HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return 0;
}
int si_hip_host_free(void* ptr) { if (ptr) SI_HIP_TRY(hipHostFree(ptr)); return 0; }
// pin caller-owned host memory in place (a borrowed Engine::Input buffer that is uploaded on every Forward): copies from it
// then run asynchronously at the link rate instead of being staged through the runtime's bounce buffers
int si_hip_host_register(void* ptr, size_t bytes) {
    if (!ptr || bytes == 0) return SI_E_BADARG;
    SI_HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return 0;
}
int si_hip_host_unregister(void* ptr) { if (ptr) SI_HIP_TRY(hipHostUnregister(ptr)); return 0; }

int si_hip_memset_async(void* ptr, int value, size_t bytes, si_stream_t s) {
    SI_HIP_TRY(hipMemsetAsync(ptr, value, bytes, (hipStream_t)s));
    return 0;
}
int si_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, si_stream_t s) {
    SI_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    return 0;
}
int si_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, si_stream_t s) {
    SI_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
    return 0;
}
int si_hip_memcpy_d2d(void* dst, const void* src, size_t bytes, si_stream_t s) {
    SI_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return 0;
}

int si_hip_stream_create(si_stream_t* s) {
    if (!s) return SI_E_BADARG;
    hipStream_t h;
    SI_HIP_TRY(hipStreamCreateWithFlags(&h, hipStreamNonBlocking));
    *s = h;
    return 0;
}
// level: -1 the device's lowest stream priority, +1 its highest, 0 the default (hipDeviceGetStreamPriorityRange: numerically lower = higher)
int si_hip_stream_create_priority(si_stream_t* s, int level) {
    if (!s) return SI_E_BADARG;
    int lo = 0, hi = 0;
    SI_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t h;
    SI_HIP_TRY(hipStreamCreateWithPriority(&h, hipStreamNonBlocking, level < 0 ? lo : (level > 0 ? hi : 0)));
    *s = h;
    return 0;
}
int si_hip_stream_destroy(si_stream_t s) { if (s) SI_HIP_TRY(hipStreamDestroy((hipStream_t)s)); return 0; }
int si_hip_stream_sync(si_stream_t s) { SI_HIP_TRY(hipStreamSynchronize((hipStream_t)s)); return 0; }
int si_hip_device_sync(void) { SI_HIP_TRY(hipDeviceSynchronize()); return 0; }

int si_hip_event_create(si_event_t* ev) {
    if (!ev) return SI_E_BADARG;
    hipEvent_t e;
    SI_HIP_TRY(hipEventCreate(&e));
    *ev = e;
    return 0;
}
int si_hip_event_destroy(si_event_t ev) { if (ev) SI_HIP_TRY(hipEventDestroy((hipEvent_t)ev)); return 0; }
int si_hip_event_record(si_event_t ev, si_stream_t s) { SI_HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)s)); return 0; }
int si_hip_event_sync(si_event_t ev) { SI_HIP_TRY(hipEventSynchronize((hipEvent_t)ev)); return 0; }
int si_hip_event_elapsed_ms(si_event_t a, si_event_t b, float* ms) {
    if (!ms) return SI_E_BADARG;
    SI_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return 0;
}

int si_hip_stream_wait_event(si_stream_t s, si_event_t ev) {
    SI_HIP_TRY(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)ev, 0));
    return 0;
}

// ---- device memory shared between the per-GPU processes of one node (direct output all-gather) ----
static_assert(sizeof(hipIpcMemHandle_t) == SI_IPC_HANDLE_BYTES, "SI_IPC_HANDLE_BYTES must match hipIpcMemHandle_t");

int si_hip_ipc_get_mem_handle(void* dptr, void* handle) {
    if (!dptr || !handle) return SI_E_BADARG;
    hipIpcMemHandle_t h;
    SI_HIP_TRY(hipIpcGetMemHandle(&h, dptr));
    memcpy(handle, &h, sizeof(h));
    return 0;
}
int si_hip_ipc_open_mem_handle(const void* handle, void** dptr) {
    if (!dptr || !handle) return SI_E_BADARG;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    *dptr = nullptr;
    SI_HIP_TRY(hipIpcOpenMemHandle(dptr, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}
int si_hip_ipc_close_mem_handle(void* dptr) { if (dptr) SI_HIP_TRY(hipIpcCloseMemHandle(dptr)); return 0; }
// the device's PCI bus id ("0000:c1:00.0"): what identifies a GPU ACROSS processes whose HIP_VISIBLE_DEVICES differ (a device
// INDEX only means something inside one process)
int si_hip_device_pci_bus_id(int device, char* buf, int len) {
    if (!buf || len < 16) return SI_E_BADARG;
    SI_HIP_TRY(hipDeviceGetPCIBusId(buf, len, device));
    return 0;
}
// index of the visible device with this PCI bus id, or -1 (hidden from this process / unknown)
int si_hip_device_by_pci_bus_id(const char* bus_id) {
    int dev = -1;
    if (!bus_id || hipDeviceGetByPCIBusId(&dev, bus_id) != hipSuccess) return -1;
    return dev;
}
int si_hip_enable_peer_access(int peer_device) {
    int cur = -1;
    SI_HIP_TRY(hipGetDevice(&cur));
    if (cur == peer_device) return 0;
    int can = 0;
    SI_HIP_TRY(hipDeviceCanAccessPeer(&can, cur, peer_device));
    if (!can) return SI_E_UNSUPPORTED;
    hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return 0; }
    return (int)e;
}

int si_hip_graph_begin_capture(si_stream_t s) {
    SI_HIP_TRY(hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeThreadLocal));
    return 0;
}
int si_hip_graph_end_capture(si_stream_t s, si_graph_t* exec) {
    if (!exec) return SI_E_BADARG;
    hipGraph_t g = nullptr;
    SI_HIP_TRY(hipStreamEndCapture((hipStream_t)s, &g));
    hipGraphExec_t e = nullptr;
    hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (rc != hipSuccess) return (int)rc;
    *exec = e;
    return 0;
}
int si_hip_graph_launch(si_graph_t exec, si_stream_t s) {
    SI_HIP_TRY(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)s));
    return 0;
}
int si_hip_graph_destroy(si_graph_t exec) { if (exec) SI_HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)exec)); return 0; }

}  // extern "C"
