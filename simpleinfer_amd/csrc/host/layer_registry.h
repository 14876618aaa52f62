// layer_registry.h -- pnnx type string -> {creator, destroyer}, plain C function pointers (reference
// src/layer_registry.h:10-18).  RegisterLayer / RegisteredLayerTypes are extensions: the reference's table is closed
// (src/layer_registry.cpp:33-49).
#pragma once

#include <string>
#include <vector>

namespace SimpleInfer {

class Layer;

using LayerCreatorFunc   = Layer* (*)();
using LayerDestroyerFunc = void (*)(Layer*);

struct LayerRegistryEntry {
    LayerCreatorFunc creator     = nullptr;
    LayerDestroyerFunc destroyer = nullptr;
};

const LayerRegistryEntry* GetLayerRegistry(std::string type);

// add (or replace) an entry at run time; returns false on null function pointers
bool RegisterLayer(const std::string& type, LayerCreatorFunc creator, LayerDestroyerFunc destroyer);

std::vector<std::string> RegisteredLayerTypes();

// true for a type whose entry came from RegisterLayer (also one that replaced a built-in): the engine knows nothing about such
// a layer beyond the Layer interface -- in particular not whether it is per-image -- and does not re-batch graphs that hold one
// unless asked to (Engine option "host_slices" > 1, "batch")
bool IsUserRegisteredLayer(const std::string& type);

}  // namespace SimpleInfer
