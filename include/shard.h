// shard.h -- batch sharding for C++ users of include/engine.h (extension; the reference is single-process,
// SURVEY.md D9).  One process per GPU of one node; every rank loads the same model at its per-rank batch
// (Engine::SetOption("batch", B / G)) and runs an independent Engine on its contiguous slab of the global batch;
// the ONLY exchange is the all-gather of the output slabs [B/G, rows, 85] -> [B, rows, 85] (BASELINE.json
// north_star), done here as a direct fan-out over IPC-shared HBM or as RCCL's ncclAllGather through the C-ABI
// (include/si_shard.h; librccl.so loaded with dlopen): no torch either way.
//
//   ShardedEngine sh;
//   sh.Init("/job42", rank, world, &engine, engine.OutputNames()[0]);   // collective, after engine.LoadModel()
//   for (;;) { engine.Input(...); sh.Forward(); ... }                    // step s computes while step s-1's slabs travel
//   sh.Flush(); sh.Gathered(t);                                          // [B, rows, 85] device tensor of the last step
#pragma once

#include <string>
#include <vector>

#include "engine.h"
#include "tensor.h"
#include "types.h"

struct SiNodeGroup;
struct SiDirectGather;

namespace SimpleInfer {

// how the output slabs travel: the direct IPC fan-out, RCCL's all-gather, or direct with a collective fallback to RCCL when
// the direct path cannot be set up on some rank (hipIpc / peer mapping refused)
enum class GatherMode { kDirect = 0, kRccl = 1, kAuto = 2 };

class ShardedEngine {
public:
    ShardedEngine();
    ~ShardedEngine();  // Release()s
    ShardedEngine(const ShardedEngine&) = delete;
    ShardedEngine& operator=(const ShardedEngine&) = delete;

    // Collective over the node's ranks.  `group_name`: POSIX shm name ("/...") unique to the job, the same on every
    // rank.  `engine` must have a model loaded with device-resident outputs (SetOption("outputs_to_host", 0)); its
    // output operand `output_name` is re-bound into this object's gathered buffers (Engine::Output).
    Status Init(const std::string& group_name, int rank, int world, Engine* engine, const std::string& output_name,
                int slots = 4, double timeout_s = 60.0, GatherMode gather = GatherMode::kAuto);
    GatherMode Mode() const;   // kDirect or kRccl: what Init ended up with
    // engine->Forward() into this step's slot, start the fan-out of the slab to every peer (asynchronous), and complete
    // the PREVIOUS step's gather (wait for its copies + node barrier).
    Status Forward();
    // complete the gather of the last Forward()
    Status Flush();
    // device tensor [world * b, ...] holding every rank's slab of the most recently completed step.  With the default 4
    // slots it stays valid through the NEXT Forward() and is overwritten during the one after (a consumer may read it
    // asynchronously while the next step computes); with 3 slots only until the next Forward() (include/si_shard.h)
    Status Gathered(Tensor& gathered) const;
    Status Release();  // collective

    int Rank() const;
    int World() const;

private:
    Engine* engine_ = nullptr;
    std::string output_name_;
    SiNodeGroup* group_ = nullptr;
    SiDirectGather* gather_ = nullptr;
    std::vector<int> local_shape_;
    long step_ = 0;        // steps issued
    int completed_ = -1;   // slot of the last completed gather
    int pending_ = -1;     // slot pushed but not completed
    int device_ = 0;
};

}  // namespace SimpleInfer
