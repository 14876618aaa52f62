#include "layer_registry.h"

#include <map>
#include <set>

namespace SimpleInfer {

#define SI_DECLARE_LAYER(type)    \
    Layer* type##_LayerCreator(); \
    void type##_LayerDestroyer(Layer*);

SI_DECLARE_LAYER(AdaptiveAvgPool2d)
SI_DECLARE_LAYER(BatchNorm2d)
SI_DECLARE_LAYER(BinaryOp)
SI_DECLARE_LAYER(Cat)
SI_DECLARE_LAYER(Conv2d)
SI_DECLARE_LAYER(Flatten)
SI_DECLARE_LAYER(HardSigmoid)
SI_DECLARE_LAYER(HardSwish)
SI_DECLARE_LAYER(LeakyReLU)
SI_DECLARE_LAYER(Linear)
SI_DECLARE_LAYER(MaxPool2d)
SI_DECLARE_LAYER(ReLU)
SI_DECLARE_LAYER(Sigmoid)
SI_DECLARE_LAYER(SiLU)
SI_DECLARE_LAYER(UnaryOp)
SI_DECLARE_LAYER(Upsample)
SI_DECLARE_LAYER(YoloDetect)

#define SI_ENTRY(pnnx_type, type) \
    { pnnx_type, LayerRegistryEntry{type##_LayerCreator, type##_LayerDestroyer} }

static std::map<std::string, LayerRegistryEntry>& Table() {
    // the 15 type strings of reference src/layer_registry.cpp:33-49, plus nn.LeakyReLU
    // (north_star extension)
    static std::map<std::string, LayerRegistryEntry> table = {
        SI_ENTRY("nn.AdaptiveAvgPool2d", AdaptiveAvgPool2d),
        SI_ENTRY("nn.BatchNorm2d", BatchNorm2d),
        SI_ENTRY("BinaryOp", BinaryOp),
        SI_ENTRY("torch.cat", Cat),
        SI_ENTRY("nn.Conv2d", Conv2d),
        SI_ENTRY("torch.flatten", Flatten),
        SI_ENTRY("nn.Hardsigmoid", HardSigmoid),
        SI_ENTRY("nn.Hardswish", HardSwish),
        SI_ENTRY("nn.LeakyReLU", LeakyReLU),
        SI_ENTRY("nn.Linear", Linear),
        SI_ENTRY("nn.MaxPool2d", MaxPool2d),
        SI_ENTRY("nn.ReLU", ReLU),
        SI_ENTRY("nn.Sigmoid", Sigmoid),
        SI_ENTRY("nn.SiLU", SiLU),
        SI_ENTRY("UnaryOp", UnaryOp),   // emitted by expand_expression, never registered by the reference
        SI_ENTRY("nn.Upsample", Upsample),
        SI_ENTRY("models.yolo.Detect", YoloDetect),
    };
    return table;
}

const LayerRegistryEntry* GetLayerRegistry(std::string type) {
    auto& t = Table();
    auto it = t.find(type);
    return it == t.end() ? nullptr : &it->second;
}

static std::set<std::string>& UserTypes() {
    static std::set<std::string> types;
    return types;
}

bool RegisterLayer(const std::string& type, LayerCreatorFunc creator, LayerDestroyerFunc destroyer) {
    if (!creator || !destroyer) return false;
    Table()[type] = LayerRegistryEntry{creator, destroyer};
    UserTypes().insert(type);
    return true;
}

bool IsUserRegisteredLayer(const std::string& type) { return UserTypes().count(type) > 0; }

std::vector<std::string> RegisteredLayerTypes() {
    std::vector<std::string> out;
    for (auto& kv : Table()) out.push_back(kv.first);
    return out;
}

}  // namespace SimpleInfer
