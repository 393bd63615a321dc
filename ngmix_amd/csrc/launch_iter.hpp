// launch_iter.hpp -- launcher prototypes for the iterative / moment kernels
// (moments.hip, em.hip, derivs.hip).
#pragma once

#include "common.hpp"

namespace ngmix {
}  // namespace ngmix
