// tests/cpp/test_layers.cpp -- the reference's layer tests, re-stated against this repo's C++ plugin surface.
//
// Same method as /root/reference/test/test_layer/*.cpp: construct the Layer object directly, set its PUBLIC fields
// (no pnnx), call Forward(input, output) on host tensors filled with U[0,1) data, and compare element-wise with a
// naive loop written here, at the reference's tolerances (test/common.h:8-11, CHECK_FLOAT_EQ = abs 1e-6).  Host tensors
// are staged through HBM by Layer::RunOnDevice, so the arithmetic under test is the HIP kernels'.
// The Engine section replays test/test_engine/test_engine.cpp / bench/bench_yolo.cpp: LoadModel, Input, Forward,
// Extract -- and, unlike the reference (which asserts nothing), checks the output against the oracle's file.
//
// usage: test_layers [model.pnnx.param model.pnnx.bin input.f32 expected.f32 output_name]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"
#include "layer/activation.h"
#include "layer/adaptive_avg_pool_2d.h"
#include "layer/binary_op.h"
#include "layer/cat.h"
#include "layer/conv_2d.h"
#include "layer/flatten.h"
#include "layer/max_pool_2d.h"
#include "layer/upsample.h"
#include "layer_registry.h"
#include "si_hip.h"
#include "tensor.h"

using namespace SimpleInfer;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        ++g_checks;                                                          \
        if (!(cond)) {                                                       \
            if (++g_fail < 20) fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                                    \
    } while (0)
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_FLOAT_EPS_EQ(a, b, eps) CHECK(std::abs((a) - (b)) < (eps))

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static float urand() {  // deterministic U[0,1) (the reference uses Eigen setRandom with srand(time))
    g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17;
    return (float)((g_rng >> 40) / 16777216.0);
}
static void fill(Tensor& t) {
    float* p = t.Data<float>();
    for (size_t i = 0; i < t.NumElements(); ++i) p[i] = urand();
}

// test/test_layer/test_conv_2d.cpp: conv0 (:8-132), conv1 groups=2 (:134-274), conv2 6x6 s2 (:276-416, at 160x160),
// conv3 1x1 (:418-558); naive loop :100-131; tolerance abs 2e-4
static void TestConv(int h, int w, int ic, int oc, int groups, int k, int s, int p) {
    const int oh = (h + 2 * p - k) / s + 1, ow = (w + 2 * p - k) / s + 1, icg = ic / groups, ocg = oc / groups;
    Tensor input(DataType::kFloat32, {1, h, w, ic}, true), output(DataType::kFloat32, {1, oh, ow, oc}, true);
    fill(input);
    Conv2d conv;
    conv.in_channels_ = ic; conv.out_channels_ = oc; conv.groups_ = groups;
    conv.kernel_h_ = conv.kernel_w_ = k; conv.stride_h_ = conv.stride_w_ = s;
    conv.dilation_h_ = conv.dilation_w_ = 1;
    conv.padding_t_ = conv.padding_b_ = conv.padding_l_ = conv.padding_r_ = p;
    std::vector<float> weight((size_t)oc * icg * k * k), bias(oc);
    for (auto& v : weight) v = urand();
    for (auto& v : bias) v = urand();
    CHECK_EQ(Status::kSuccess, conv.SetWeights(weight, bias));
    CHECK_EQ(Status::kSuccess, conv.Forward(input, output));
    const float* in = input.Data<float>();
    const float* out = output.Data<float>();
    double scale = 1.0;
    for (int j = 0; j < oh; ++j)
        for (int x = 0; x < ow; ++x)
            for (int l = 0; l < oc; ++l) {
                const int g = l / ocg;
                float sum = 0.0f;
                for (int c = 0; c < icg; ++c)
                    for (int a = 0; a < k; ++a)
                        for (int b = 0; b < k; ++b) {
                            const int y = j * s - p + a, xx = x * s - p + b;
                            if (y < 0 || y >= h || xx < 0 || xx >= w) continue;
                            sum += in[((size_t)y * w + xx) * ic + g * icg + c] * weight[(((size_t)l * icg + c) * k + a) * k + b];
                        }
                sum += bias[l];
                scale = std::max(scale, (double)std::abs(sum) / 16.0);
                CHECK_FLOAT_EPS_EQ(out[((size_t)j * ow + x) * oc + l], sum, 2e-4 * scale);
            }
}

// test/test_layer/test_max_pool_2d.cpp:7-73 (k2 s2) and :75-150 (8x20x20x256 k5 s1 p2); exact
static void TestMaxPool(int n, int h, int w, int c, int k, int s, int p) {
    const int oh = (h + 2 * p - k) / s + 1, ow = (w + 2 * p - k) / s + 1;
    Tensor input(DataType::kFloat32, {n, h, w, c}, true), output(DataType::kFloat32, {n, oh, ow, c}, true);
    fill(input);
    MaxPool2d pool;
    pool.kernel_h_ = pool.kernel_w_ = k; pool.stride_h_ = pool.stride_w_ = s;
    pool.padding_t_ = pool.padding_b_ = pool.padding_l_ = pool.padding_r_ = p;
    CHECK_EQ(Status::kSuccess, pool.Forward(input, output));
    const float* in = input.Data<float>();
    const float* out = output.Data<float>();
    for (int b = 0; b < n; ++b)
        for (int j = 0; j < oh; ++j)
            for (int x = 0; x < ow; ++x)
                for (int ch = 0; ch < c; ++ch) {
                    float m = -3.4e38f;
                    for (int a = 0; a < k; ++a)
                        for (int q = 0; q < k; ++q) {
                            const int y = j * s - p + a, xx = x * s - p + q;
                            if (y < 0 || y >= h || xx < 0 || xx >= w) continue;
                            m = std::max(m, in[(((size_t)b * h + y) * w + xx) * c + ch]);
                        }
                    CHECK_EQ(out[(((size_t)b * oh + j) * ow + x) * c + ch], m);
                }
}

// test/test_layer/test_upsample.cpp:8-54, :56-102; exact; index rule upsample.cpp:85-92
static void TestUpsample(int n, int h, int w, int c, float scale) {
    const int oh = (int)(h * scale), ow = (int)(w * scale);
    Tensor input(DataType::kFloat32, {n, h, w, c}, true), output(DataType::kFloat32, {n, oh, ow, c}, true);
    fill(input);
    Upsample up;
    up.scale_factor_h_ = up.scale_factor_w_ = scale;
    CHECK_EQ(Status::kSuccess, up.Forward(input, output));
    const float* in = input.Data<float>();
    const float* out = output.Data<float>();
    const float inv = 1.0f / scale;
    for (int b = 0; b < n; ++b)
        for (int j = 0; j < oh; ++j)
            for (int x = 0; x < ow; ++x) {
                const int y = std::max(0, std::min(h - 1, (int)((float)j * inv)));
                const int xx = std::max(0, std::min(w - 1, (int)((float)x * inv)));
                for (int ch = 0; ch < c; ++ch)
                    CHECK_EQ(out[(((size_t)b * oh + j) * ow + x) * c + ch], in[(((size_t)b * h + y) * w + xx) * c + ch]);
            }
}

// test/test_layer/test_cat.cpp:7-65: three inputs with C = 3, 2, 4 along dim=1 (NCHW) -> channel axis; exact
static void TestCat() {
    const int n = 1, h = 8, w = 8, cs[3] = {3, 2, 4};
    std::vector<Tensor> inputs;
    for (int c : cs) {
        inputs.emplace_back(DataType::kFloat32, std::vector<int>{n, h, w, c}, false);
    }
    std::vector<std::vector<float>> store(3);
    for (int i = 0; i < 3; ++i) {
        store[i].resize((size_t)n * h * w * cs[i]);
        for (auto& v : store[i]) v = urand();
        inputs[i].SetData(store[i].data());
    }
    Tensor output(DataType::kFloat32, {n, h, w, 9}, true);
    Cat cat;
    cat.dim_ = 1;
    CHECK_EQ(Status::kSuccess, cat.Forward(inputs, output));
    const float* out = output.Data<float>();
    for (int px = 0; px < n * h * w; ++px) {
        int off = 0;
        for (int i = 0; i < 3; ++i) {
            for (int c = 0; c < cs[i]; ++c) CHECK_EQ(out[(size_t)px * 9 + off + c], store[i][(size_t)px * cs[i] + c]);
            off += cs[i];
        }
    }
}

// test_silu.cpp / test_relu.cpp / test_sigmoid.cpp / test_hard_*.cpp: 1x128x128x3, abs 1e-6 (relu exact)
template<typename L, typename F>
static void TestActivation(F ref, double eps) {
    Tensor input(DataType::kFloat32, {1, 128, 128, 3}, true), output(DataType::kFloat32, {1, 128, 128, 3}, true);
    float* in = input.Data<float>();
    for (size_t i = 0; i < input.NumElements(); ++i) in[i] = urand() * 8.0f - 4.0f;
    L layer;
    CHECK_EQ(Status::kSuccess, layer.Forward(input, output));
    const float* out = output.Data<float>();
    for (size_t i = 0; i < input.NumElements(); ++i) CHECK_FLOAT_EPS_EQ(out[i], ref(in[i]), eps);
}

// test/test_layer/test_binary_op.cpp:7-87 add / mul on 1x128x128x3; abs 1e-6
static void TestBinary(BinaryOp::BinaryOpType type) {
    Tensor a(DataType::kFloat32, {1, 128, 128, 3}, true), b(DataType::kFloat32, {1, 128, 128, 3}, true),
        o(DataType::kFloat32, {1, 128, 128, 3}, true);
    fill(a);
    fill(b);
    BinaryOp op;
    op.binary_op_type_ = type;
    std::vector<Tensor> ins{a, b};
    CHECK_EQ(Status::kSuccess, op.Forward(ins, o));
    for (size_t i = 0; i < o.NumElements(); ++i) {
        const float x = a.Data<float>()[i], y = b.Data<float>()[i];
        CHECK_FLOAT_EPS_EQ(o.Data<float>()[i], type == BinaryOp::BinaryOpType::kAdd ? x + y : x * y, 1e-6);
    }
}

// test_adaptive_avg_pool_2d.cpp:7-52 and test_flatten.cpp:7-45
static void TestPoolFlatten() {
    Tensor in(DataType::kFloat32, {1, 8, 8, 3}, true), out(DataType::kFloat32, {1, 1, 1, 3}, true);
    fill(in);
    AdaptiveAvgPool2d gap;
    gap.output_h_ = gap.output_w_ = 1;
    CHECK_EQ(Status::kSuccess, gap.Forward(in, out));
    for (int c = 0; c < 3; ++c) {
        float s = 0.f;
        for (int p = 0; p < 64; ++p) s += in.Data<float>()[p * 3 + c];
        CHECK_FLOAT_EPS_EQ(out.Data<float>()[c], s / 64.0f, 1e-6);
    }
    Tensor fin(DataType::kFloat32, {1, 2, 2, 128}, true), fout(DataType::kFloat32, {1, 512}, true);
    fill(fin);
    Flatten fl;
    CHECK_EQ(Status::kSuccess, fl.Forward(fin, fout));
    for (int c = 0; c < 128; ++c)
        for (int y = 0; y < 2; ++y)
            for (int x = 0; x < 2; ++x) CHECK_EQ(fout.Data<float>()[(c * 2 + y) * 2 + x], fin.Data<float>()[(y * 2 + x) * 128 + c]);
}

static std::vector<float> read_f32(const char* path) {
    std::vector<float> v;
    FILE* f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) v.clear();
    fclose(f);
    return v;
}

// the Engine walk-through of test/test_engine/test_engine.cpp:8-30 and bench/bench_yolo.cpp:12-28
static void TestEngine(const char* param, const char* bin, const char* input_path, const char* expected_path, const char* out_name) {
    InitializeContext();
    Engine engine;
    CHECK_EQ(Status::kSuccess, engine.LoadModel(param, bin));
    const std::vector<std::string> ins = engine.InputNames(), outs = engine.OutputNames();
    CHECK_EQ(ins.size(), (size_t)1);
    CHECK(outs.size() >= 1);
    std::vector<int> ishape;
    CHECK_EQ(Status::kSuccess, engine.OperandShape(ins[0], ishape));
    std::vector<float> x = read_f32(input_path), expect = read_f32(expected_path);
    Tensor input(DataType::kFloat32, ishape, false);
    CHECK_EQ(x.size(), input.NumElements());
    input.SetData(x.data());
    CHECK_EQ(Status::kSuccess, engine.Input(ins[0], input));
    CHECK_EQ(Status::kFail, engine.Input("not-an-input", input));
    for (int rep = 0; rep < 2; ++rep) {  // Input once, Forward many times (bench_yolo.cpp:22-27)
        CHECK_EQ(Status::kSuccess, engine.Forward());
        Tensor output;
        CHECK_EQ(Status::kSuccess, engine.Extract(out_name, output));
        CHECK_EQ(output.NumElements(), expect.size());
        CHECK(!output.OwnsData());  // Extract hands out a view (src/engine_impl.cpp:552)
        double maxref = 0, maxerr = 0;
        for (size_t i = 0; i < expect.size(); ++i) {
            maxref = std::max(maxref, (double)std::abs(expect[i]));
            maxerr = std::max(maxerr, (double)std::abs(expect[i] - output.Data<float>()[i]));
        }
        CHECK(maxerr <= 1e-4 * maxref);
        if (rep == 0) printf("engine: max|diff| %.3e vs max|ref| %.3e\n", maxerr, maxref);
    }
    Tensor bad;
    CHECK_EQ(Status::kFail, engine.Extract("not-an-output", bad));

    // extensions: results left on the device and written into a caller-owned buffer (Engine::Output)
    {
        Engine dev;
        CHECK_EQ(Status::kSuccess, dev.SetOption("outputs_to_host", 0));
        CHECK_EQ(Status::kUnsupport, dev.SetOption("no-such-option", 1));
        CHECK_EQ(Status::kSuccess, dev.LoadModel(param, bin));
        CHECK_EQ(Status::kSuccess, dev.Input(ins[0], input));
        Tensor mine(DataType::kFloat32, {(int)expect.size()}, MemoryType::kDevice, true);
        Tensor host_tensor(DataType::kFloat32, {(int)expect.size()}, true);
        CHECK_EQ(Status::kUnsupport, dev.Output(out_name, host_tensor));   // must be device memory
        CHECK_EQ(Status::kFail, dev.Output("not-an-output", mine));
        CHECK_EQ(Status::kSuccess, dev.Output(out_name, mine));
        CHECK_EQ(Status::kSuccess, dev.Forward());
        Tensor view;
        CHECK_EQ(Status::kSuccess, dev.Extract(out_name, view));
        CHECK(view.GetMemoryType() == MemoryType::kDevice);
        CHECK(view.RawData() == mine.RawData());
        std::vector<float> back(expect.size());
        CHECK_EQ(0, si_hip_memcpy_d2h(back.data(), mine.RawData(), back.size() * sizeof(float), nullptr));
        CHECK_EQ(0, si_hip_stream_sync(nullptr));
        double maxref = 0, maxerr = 0;
        for (size_t i = 0; i < expect.size(); ++i) {
            maxref = std::max(maxref, (double)std::abs(expect[i]));
            maxerr = std::max(maxerr, (double)std::abs(expect[i] - back[i]));
        }
        CHECK(maxerr <= 1e-4 * maxref);
    }
    // extension: one file, any batch (SetOption("batch", N)): the first image of a batch-1 engine equals image 0 above
    {
        Engine one;
        CHECK_EQ(Status::kSuccess, one.SetOption("batch", 1));
        CHECK_EQ(Status::kSuccess, one.LoadModel(param, bin));
        std::vector<int> s1;
        CHECK_EQ(Status::kSuccess, one.OperandShape(ins[0], s1));
        CHECK_EQ(s1[0], 1);
        Tensor in1(DataType::kFloat32, s1, false);
        in1.SetData(x.data());
        CHECK_EQ(Status::kSuccess, one.Input(ins[0], in1));
        CHECK_EQ(Status::kSuccess, one.Forward());
        Tensor o1;
        CHECK_EQ(Status::kSuccess, one.Extract(out_name, o1));
        CHECK_EQ(o1.NumElements() * (size_t)ishape[0], expect.size());
        Tensor full;
        CHECK_EQ(Status::kSuccess, engine.Extract(out_name, full));
        bool same = true;
        for (size_t i = 0; i < o1.NumElements(); ++i) same = same && o1.Data<float>()[i] == full.Data<float>()[i];
        CHECK(same);  // an image's result does not depend on the batch it travels in
    }
    CHECK_EQ(Status::kSuccess, engine.Release());
    CHECK_EQ(engine.InputNames().size(), (size_t)0);
    CHECK(nullptr != GetLayerRegistry("nn.Conv2d"));
    CHECK(nullptr == GetLayerRegistry("nn.DoesNotExist"));
}

int main(int argc, char** argv) {
    TestConv(128, 128, 32, 16, 1, 3, 1, 1);
    TestConv(128, 128, 32, 16, 2, 3, 1, 1);
    TestConv(160, 160, 3, 32, 1, 6, 2, 2);
    TestConv(10, 10, 256, 255, 1, 1, 1, 0);
    TestConv(28, 28, 48, 48, 48, 3, 1, 1);   // depthwise: the reference's slowest path (conv_2d.cpp:285-380)
    TestConv(29, 27, 40, 40, 40, 5, 2, 2);
    TestMaxPool(1, 8, 8, 3, 2, 2, 0);
    TestMaxPool(8, 20, 20, 256, 5, 1, 2);
    TestUpsample(1, 16, 16, 3, 2.0f);
    TestUpsample(4, 10, 10, 128, 2.0f);
    TestCat();
    TestActivation<SiLU>([](float v) { return v / (1.0f + std::exp(-v)); }, 2e-6);
    TestActivation<ReLU>([](float v) { return std::max(v, 0.0f); }, 1e-30);
    TestActivation<Sigmoid>([](float v) { return 1.0f / (1.0f + std::exp(-v)); }, 1e-6);
    TestActivation<HardSigmoid>([](float v) { return std::min(std::max(v / 6.0f + 0.5f, 0.0f), 1.0f); }, 1e-6);
    TestActivation<HardSwish>([](float v) { return v * std::min(std::max(v / 6.0f + 0.5f, 0.0f), 1.0f); }, 2e-6);
    TestBinary(BinaryOp::BinaryOpType::kAdd);
    TestBinary(BinaryOp::BinaryOpType::kMul);
    TestPoolFlatten();
    if (argc >= 6) TestEngine(argv[1], argv[2], argv[3], argv[4], argv[5]);
    printf("%d checks, %d failed\n", g_checks, g_fail);
    return g_fail == 0 ? 0 : 1;
}
