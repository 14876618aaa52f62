// sharded_engine.cpp -- SimpleInfer::ShardedEngine (include/shard.h): Engine::Forward() + the direct output
// all-gather of include/si_shard.h, one step in flight behind the next step's compute.
#include "shard.h"

#include "logger.h"
#include "si_hip.h"
#include "si_shard.h"

namespace SimpleInfer {

#define SI_TRY_SHARD(expr, what)                                                     \
    {                                                                                \
        const int _rc = (expr);                                                      \
        if (_rc != 0) {                                                              \
            LOG(ERROR) << what << " failed with code " << _rc;                       \
            return Status::kFail;                                                    \
        }                                                                            \
    }

ShardedEngine::ShardedEngine() {}
ShardedEngine::~ShardedEngine() { Release(); }

int ShardedEngine::Rank() const { return group_ ? si_group_rank(group_) : 0; }
int ShardedEngine::World() const { return group_ ? si_group_world(group_) : 1; }

GatherMode ShardedEngine::Mode() const { return gather_ && si_gather_mode(gather_) == SI_GATHER_RCCL ? GatherMode::kRccl : GatherMode::kDirect; }

Status ShardedEngine::Init(const std::string& group_name, int rank, int world, Engine* engine, const std::string& output_name,
                           int slots, double timeout_s, GatherMode gather) {
    CHECK_BOOL(engine != nullptr && group_ == nullptr);
    Tensor out;
    CHECK_STATUS(engine->Extract(output_name, out));
    if (out.GetMemoryType() != MemoryType::kDevice) {
        LOG(ERROR) << "ShardedEngine: the engine must keep its outputs on the device (SetOption(\"outputs_to_host\", 0))";
        return Status::kUnsupport;
    }
    SI_TRY_SHARD(si_hip_get_device(&device_), "hipGetDevice");
    SI_TRY_SHARD(si_group_create(group_name.c_str(), rank, world, timeout_s, &group_), "si_group_create");
    const int rc = si_gather_create_mode(group_, device_, out.ByteSize(), slots, (int)gather, &gather_);
    if (rc != 0) {
        LOG(ERROR) << "si_gather_create_mode failed with code " << rc;
        si_group_destroy(group_);
        group_ = nullptr;
        return Status::kFail;
    }
    engine_ = engine;
    output_name_ = output_name;
    local_shape_ = out.Shape();
    step_ = 0;
    completed_ = pending_ = -1;
    return Status::kSuccess;
}

Status ShardedEngine::Forward() {
    CHECK_BOOL(engine_ != nullptr && gather_ != nullptr);
    const int slot = (int)(step_ % si_gather_slots(gather_));
    Tensor target(DataType::kFloat32, local_shape_, MemoryType::kDevice, false);
    CHECK_STATUS(target.SetData(si_gather_slab(gather_, slot), MemoryType::kDevice));
    CHECK_STATUS(engine_->Output(output_name_, target));
    CHECK_STATUS(engine_->Forward());  // synchronous: the slab is complete
    SI_TRY_SHARD(si_gather_push(gather_, slot, engine_->Stream()), "si_gather_push");
    ++step_;
    const int prev = pending_;
    pending_ = slot;
    if (si_gather_slots(gather_) == 1) return Flush();  // no overlap possible with a single buffer
    if (prev >= 0) {
        SI_TRY_SHARD(si_gather_complete(gather_, prev), "si_gather_complete");
        completed_ = prev;
    }
    return Status::kSuccess;
}

Status ShardedEngine::Flush() {
    CHECK_BOOL(gather_ != nullptr);
    if (pending_ >= 0) {
        SI_TRY_SHARD(si_gather_complete(gather_, pending_), "si_gather_complete");
        completed_ = pending_;
        pending_ = -1;
    }
    return Status::kSuccess;
}

Status ShardedEngine::Gathered(Tensor& gathered) const {
    if (gather_ == nullptr || completed_ < 0) return Status::kEmpty;
    std::vector<int> shape = local_shape_;
    if (shape.empty()) return Status::kErrorShape;
    shape[0] *= si_group_world(group_);
    gathered = Tensor(DataType::kFloat32, shape, MemoryType::kDevice, false);
    return gathered.SetData(si_gather_buffer(gather_, completed_), MemoryType::kDevice);
}

Status ShardedEngine::Release() {
    Status ret = Status::kSuccess;
    if (gather_) {
        if (engine_) engine_->Output(output_name_, Tensor());  // back to the engine's own buffer before ours is freed
        if (si_gather_destroy(gather_) != 0) ret = Status::kFail;
        gather_ = nullptr;
    }
    if (group_) {
        si_group_destroy(group_);
        group_ = nullptr;
    }
    engine_ = nullptr;
    return ret;
}

}  // namespace SimpleInfer
