#include "context.h"

#include "logger.h"

namespace SimpleInfer {

Context::Context() {}

Context::~Context() {
    if (owns_stream_ && stream_) si_hip_stream_destroy(stream_);
}

Status Context::Init(int device, int stream_priority) {
    int count = 0;
    if (si_hip_device_count(&count) != 0 || count <= 0) {
        LOG(ERROR) << "no HIP device available: the MI355X path has no CPU fallback";
        return Status::kErrorContext;
    }
    if (device < 0) {
        if (si_hip_get_device(&device) != 0) device = 0;
    }
    if (device >= count) {
        LOG(ERROR) << "device " << device << " out of range (" << count << " devices)";
        return Status::kErrorContext;
    }
    int rc = si_hip_set_device(device);
    if (rc != 0) {
        LOG(ERROR) << "hipSetDevice(" << device << "): " << si_hip_error_string(rc);
        return Status::kErrorContext;
    }
    device_ = device;
    rc = stream_priority ? si_hip_stream_create_priority(&stream_, stream_priority) : si_hip_stream_create(&stream_);
    if (rc != 0) {
        LOG(ERROR) << "hipStreamCreate: " << si_hip_error_string(rc);
        return Status::kErrorContext;
    }
    owns_stream_ = true;
    return Status::kSuccess;
}

Context* Context::Default() {
    static Context ctx;  // NULL stream, current device
    return &ctx;
}

}  // namespace SimpleInfer
