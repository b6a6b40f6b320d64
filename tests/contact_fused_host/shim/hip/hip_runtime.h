// Host shim for tests/contact_fused_host: lets the device constraint header compile as plain C++ (the arithmetic is IEEE fp32 either way; no FMA contraction).
#pragma once
#include <cmath>
#include <cstring>
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
