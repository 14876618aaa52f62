// layer/operators.h -- every small operator of the path in one include: pooling, batch-norm, binary / unary arithmetic, concat,
// flatten, linear, nearest upsample.  Each class lives in its own header under the reference's file name (src/layer/<name>.h)
// and keeps the reference layer's name, public fields and virtuals, because the reference's own tests construct layers directly
// and poke those fields; what differs is underneath: Forward() enqueues one HIP kernel from
// include/si_hip.h on the context's stream.  Conv2d, YoloDetect and the activation family have their own headers too.
#pragma once

#include "adaptive_avg_pool_2d.h"
#include "batch_norm_2d.h"
#include "binary_op.h"
#include "cat.h"
#include "flatten.h"
#include "linear.h"
#include "max_pool_2d.h"
#include "output_cast.h"
#include "unary_op.h"
#include "upsample.h"
