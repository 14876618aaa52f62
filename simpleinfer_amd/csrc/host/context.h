// context.h -- per-engine execution context.  The reference's Context owns an Eigen thread pool
// (src/context.h:11-28, src/context.cpp:9-26); here it owns what a HIP launch needs: the device,
// the stream every layer launches on, and the profiling switch.
#ifndef SIMPLE_INFER_SRC_CONTEXT_H_
#define SIMPLE_INFER_SRC_CONTEXT_H_

#include "si_hip.h"
#include "types.h"

namespace SimpleInfer {

class Context {
public:
    Context();
    virtual ~Context();

    // binds the device and creates the stream; kErrorContext when no HIP device is usable
    Status Init(int device, int stream_priority = 0);   // stream_priority: si_hip_stream_create_priority's level

    int device() const { return device_; }
    si_stream_t stream() const { return stream_; }

    // context used by layers constructed outside an engine (unit tests): default device, NULL stream
    static Context* Default();

protected:
    int device_ = -1;
    si_stream_t stream_ = nullptr;
    bool owns_stream_ = false;
};

}  // namespace SimpleInfer

#endif
