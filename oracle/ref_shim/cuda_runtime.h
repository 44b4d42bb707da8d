// Fake CUDA runtime surface for building the *reference* allocator on a CPU-only box.
// TEST INFRASTRUCTURE ONLY (oracle/_ref). Nothing in the product includes this file.
#pragma once
#include "cuda.h"
