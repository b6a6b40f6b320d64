// oracle/wide — TEST INFRASTRUCTURE (checker + cpu_baseline leg of bench.py), never part of the product path. PARITY UNPINNED like oracle/: the reference cannot be
// built or run here and holds no golden vectors for this path; what this directory adds is a second, independent reading.
//
// A second, independent CPU restatement of the reference's solver hot path, transcribed from the C# ONLY (it shares no text with
// oracle/bepu_*.h or the device headers; whoever edits this directory should keep it that way — its value is that a misreading of
// the C# would have to be made twice, in two differently shaped programs, to go unnoticed).
//
// Shape: the reference's own. `Vector<float>` on an AVX2 host is 8 lanes (BepuUtilities/BundleIndexing.cs:50-60); every
// *Wide type below holds `VF` members and every function works on a whole bundle, exactly like the C# it follows. Element-wise
// IEEE binary32, no FMA contraction (build with -ffp-contract=off), correctly rounded sqrt/div (vsqrtps/vdivps).
//
// This header: System.Numerics.Vector<T> for T = float / int, W = 8.
#pragma once
#include <immintrin.h>
#include <stdint.h>

namespace wide {

constexpr int W = 8;  // Vector<float>.Count on AVX2

typedef float VF __attribute__((vector_size(32)));
typedef int32_t VI __attribute__((vector_size(32)));

static inline VF vf(float s) { return VF{s, s, s, s, s, s, s, s}; }   // new Vector<float>(s)
static inline VI vi(int32_t s) { return VI{s, s, s, s, s, s, s, s}; } // new Vector<int>(s)
static const VF kZero = {0, 0, 0, 0, 0, 0, 0, 0};                    // Vector<float>.Zero
static const VF kOne = {1, 1, 1, 1, 1, 1, 1, 1};                     // Vector<float>.One

// Unary minus of a Vector<float> — the ONE place where the C# text alone does not fix the result. `-v` compiles to Vector<T>.op_UnaryNegation, whose
// definition belongs to the runtime, not to the library: up to .NET 8 (the library's target, CommonSettings.props:3) it is `Zero - value` (and the JIT
// imports it as a subtraction from zero), so -(+0) = +0; from .NET 9 on (the runtime the reference's own test project runs on, DemoTests.csproj) it
// flips the sign bit, so -(+0) = -0. The two differ ONLY in the sign of zero results (and a zero's sign can only ever reach another zero's sign here:
// every division by a possibly-zero quantity in these constraints is guarded by a select). Default = sign flip, the convention oracle/ and the device
// code also follow; -DWIDE_NEGATE_ZERO_MINUS builds the .NET <= 8 form, and tests/test_oracle_wide.py checks that it changes nothing but zero signs.
#ifdef WIDE_NEGATE_ZERO_MINUS
static inline VF neg(VF v) { return kZero - v; }
#else
static inline VF neg(VF v) { return -v; }
#endif

// Vector.Min / Vector.Max: minps / maxps on the reference's x86 hosts (second operand returned when either is NaN or both are zero).
static inline VF Min(VF a, VF b) { return (VF)_mm256_min_ps((__m256)a, (__m256)b); }
static inline VF Max(VF a, VF b) { return (VF)_mm256_max_ps((__m256)a, (__m256)b); }
static inline VF Abs(VF a) { return (VF)_mm256_andnot_ps(_mm256_set1_ps(-0.0f), (__m256)a); }
static inline VF SquareRoot(VF a) { return (VF)_mm256_sqrt_ps((__m256)a); }
static inline VF Floor(VF a) { return (VF)_mm256_floor_ps((__m256)a); }

// Comparisons return Vector<int> masks (all bits set where true), like Vector.LessThan & co.
static inline VI LessThan(VF a, VF b) { return (VI)(a < b); }
static inline VI GreaterThan(VF a, VF b) { return (VI)(a > b); }
static inline VI LessThanOrEqual(VF a, VF b) { return (VI)(a <= b); }
static inline VI GreaterThanOrEqual(VF a, VF b) { return (VI)(a >= b); }
static inline VI Equals(VF a, VF b) { return (VI)(a == b); }
static inline VI EqualsI(VI a, VI b) { return (VI)(a == b); }
static inline VI BitwiseAnd(VI a, VI b) { return a & b; }
static inline VI BitwiseOr(VI a, VI b) { return a | b; }
static inline VI AndNot(VI a, VI b) { return a & ~b; }  // Vector.AndNot(left, right) = left & ~right
static inline VI OnesComplement(VI a) { return ~a; }
static inline VF BitwiseOrF(VF a, VF b) { return (VF)((VI)a | (VI)b); }

// Vector.ConditionalSelect(mask, left, right): bitwise (mask & left) | (~mask & right).
static inline VF ConditionalSelect(VI mask, VF left, VF right) { return (VF)_mm256_blendv_ps((__m256)right, (__m256)left, (__m256)mask); }
static inline VI ConditionalSelectI(VI mask, VI left, VI right) { return (mask & left) | (~mask & right); }

static inline bool LessThanAny(VI a, VI b) { return _mm256_movemask_ps((__m256)(VI)(a < b)) != 0; }

}  // namespace wide
