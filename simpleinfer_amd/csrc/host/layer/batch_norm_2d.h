// layer/batch_norm_2d.h -- kept for source compatibility with the reference's include path; the class lives in operators.h
#pragma once
#include "operators.h"
