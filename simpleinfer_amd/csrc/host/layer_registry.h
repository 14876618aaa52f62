// layer_registry.h -- kept for source compatibility with the reference's include path; the registry lives in layer.h
#pragma once
#include "layer.h"
