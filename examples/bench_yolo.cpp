// examples/bench_yolo.cpp -- the reference's only benchmark (bench/bench_yolo.cpp:7-34: load, Input once, one warm-up
// Forward, then time Forward in a loop, Extract) against this library's Engine, without Google Benchmark (an absent
// submodule): std::chrono around N forwards.  Two timings are printed: the reference's usage (host tensor in, host
// view out: the input is re-uploaded and the output copied back on every Forward) and the device-resident usage
// (SetOption("outputs_to_host", 0) + a device input), which is what bench.py reports.
//
//   g++ -std=c++17 -O2 -Iinclude examples/bench_yolo.cpp -Lsimpleinfer_amd -lsimpleinfer_amd -lsi_hip \
//       -Wl,-rpath,$PWD/simpleinfer_amd -o bench_yolo && ./bench_yolo model.pnnx.param model.pnnx.bin [iterations]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "engine.h"
#include "si_hip.h"  // si_hip_memcpy_h2d for the one-time upload of the device-resident input

using namespace SimpleInfer;

static double time_forwards(Engine& engine, int iters) {
    if (engine.Forward() != Status::kSuccess) return -1.0;  // warm-up, as the reference does
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i)
        if (engine.Forward() != Status::kSuccess) return -1.0;
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <model.pnnx.param> <model.pnnx.bin> [iterations]\n", argv[0]);
        return 2;
    }
    const int iters = argc > 3 ? atoi(argv[3]) : 20;
    InitializeContext();

    for (int device_resident = 0; device_resident < 2; ++device_resident) {
        Engine engine;
        if (device_resident) engine.SetOption("outputs_to_host", 0);
        else engine.SetOption("pin_inputs", 1);   // host_input below stays alive and mapped for the whole loop (include/engine.h)
        if (engine.LoadModel(argv[1], argv[2]) != Status::kSuccess) {
            fprintf(stderr, "LoadModel failed\n");
            return 1;
        }
        const std::string in_name = engine.InputNames()[0], out_name = engine.OutputNames()[0];
        std::vector<int> shape;
        engine.OperandShape(in_name, shape);
        Tensor host_input(DataType::kFloat32, shape, true);  // NHWC, like the reference's {8, 640, 640, 3}
        for (size_t i = 0; i < host_input.NumElements(); ++i) host_input.Data<float>()[i] = (float)(i % 255) / 255.0f;
        Tensor device_input(DataType::kFloat32, shape, MemoryType::kDevice, true);
        if (device_resident) {
            // one upload; Forward then reads the tensor in place
            si_hip_memcpy_h2d(device_input.RawData(), host_input.RawData(), host_input.ByteSize(), nullptr);
            si_hip_stream_sync(nullptr);
            engine.Input(in_name, device_input);
        } else {
            engine.Input(in_name, host_input);
        }
        const double ms = time_forwards(engine, iters);
        Tensor output;
        engine.Extract(out_name, output);
        if (ms < 0 || output.NumElements() == 0) {
            fprintf(stderr, "Forward failed\n");
            return 1;
        }
        printf("%-16s batch %d: %.3f ms per Forward, %.1f images/sec (output %zu floats, %s memory)\n",
               device_resident ? "device-resident" : "host tensors", shape[0], ms, shape[0] * 1000.0 / ms, output.NumElements(),
               output.GetMemoryType() == MemoryType::kDevice ? "device" : "host");
    }
    return 0;
}
