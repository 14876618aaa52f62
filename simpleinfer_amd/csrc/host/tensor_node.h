// tensor_node.h -- kept for source compatibility with the reference's include path; TensorNode lives in layer.h
#pragma once
#include "layer.h"
