// Device-side scalar-per-lane fp32 math for the gfx950 solver kernels (one GPU lane = one lane of the reference's
// Vector<float>). No MFMA: this is not a dense contraction. Operation order follows the reference file:line cited on
// each function so results are bit-identical to an IEEE evaluation of the C# (no FMA contraction: this translation
// unit is compiled with -ffp-contract=off and correctly rounded fp32 divide/sqrt).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)
#define BD_FN __device__ __forceinline__ static


namespace bd {

// pin(values...): an empty asm that "rewrites" every 32-bit word of its arguments in place. Everything the values depend on has to be
// computed before this point and nothing that uses them can be scheduled above it, so work placed before a gate (see the constraint
// functions) cannot be sunk past the gate's wait loop by the optimizer.
template <class T>
BD_FN void pin_one(T& t) {
    static_assert(sizeof(T) % 4 == 0, "pin works on 32-bit words");
    float* f = reinterpret_cast<float*>(&t);
    _Pragma("unroll") for (int i = 0; i < (int)(sizeof(T) / 4); ++i) asm volatile("" : "+v"(f[i]));
}
template <class... Ts>
BD_FN void pin(Ts&... ts) { (pin_one(ts), ...); }
// BD_GATE: pass the gate; gates that wait (G::kPin) first pin the listed velocity-independent values.
#ifndef BEPU_PIN_ENABLED
#define BEPU_PIN_ENABLED 1
#endif
// three- and four-body constraints hand the gate their velocity array
#define BD_GATE_N(v, ...)                                                            \
    do {                                                                             \
        if constexpr (std::remove_reference_t<G>::kPin && BEPU_PIN_ENABLED) pin(__VA_ARGS__); \
        gate.many(v);                                                                \
    } while (0)
#define BD_GATE(vA, vB, ...)                                                         \
    do {                                                                             \
        if constexpr (std::remove_reference_t<G>::kPin && BEPU_PIN_ENABLED) pin(__VA_ARGS__); \
        gate(vA, vB);                                                                \
    } while (0)


// Vector.Min/Max lower to minps/maxps on the reference's AVX2 hosts: "a<b?a:b" / "a>b?a:b"
// (second operand returned on NaN) — SURVEY.md A.11.
BD_FN float vmin(float a, float b) { return a < b ? a : b; }
BD_FN float vmax(float a, float b) { return a > b ? a : b; }
BD_FN float vabs(float a) { return fabsf(a); }
BD_FN float sel(bool c, float a, float b) { return c ? a : b; }

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct Q { float x, y, z, w; };
struct Sym2 { float xx, yx, yy; };
struct Sym3 { float xx, yx, yy, zx, zy, zz; };
struct M3 { V3 X, Y, Z; };
struct M23 { V3 X, Y; };
struct Sym4 { float xx, yx, yy, zx, zy, zz, wx, wy, wz, ww; };
struct Sym5 { Sym3 A; M23 B; Sym2 D; };
struct BodyVel { V3 lin, ang; };
struct Inertia { Sym3 t; float invMass; };

// ---- BepuUtilities/MathHelper.cs:11-17 constants ----
static constexpr float kPi = 3.141592653589793239f;
static constexpr float kTwoPi = 6.283185307179586477f;
static constexpr float kPiOver2 = 1.570796326794896619f;

// ---- Vector3Wide (BepuUtilities/Vector3Wide.cs) ----
BD_FN V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }              // :55
BD_FN V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }              // :127
BD_FN float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }            // :201
BD_FN V3 scale(V3 v, float s) { return {v.x * s, v.y * s, v.z * s}; }                // :343
BD_FN V3 neg(V3 v) { return {-v.x, -v.y, -v.z}; }                                    // :433
BD_FN V3 cross(V3 a, V3 b) {                                                         // :519
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
BD_FN float lengthSquared(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }        // :562
BD_FN float length(V3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }    // :573
BD_FN float distance(V3 a, V3 b) {                                                   // :627
    float x = b.x - a.x, y = b.y - a.y, z = b.z - a.z;
    return sqrtf(x * x + y * y + z * z);
}
BD_FN V3 sel3(bool c, V3 a, V3 b) { return c ? a : b; }                              // :716

// ---- Vector2Wide (BepuUtilities/Vector2Wide.cs) ----
BD_FN V2 add(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
BD_FN V2 sub(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
BD_FN V2 scale(V2 v, float s) { return {v.x * s, v.y * s}; }                          // :89
BD_FN float length(V2 v) { return sqrtf(v.x * v.x + v.y * v.y); }                 // :147

// ---- MathHelper.Cos/Sin/Acos rational approximations (BepuUtilities/MathHelper.cs:274-368) ----
BD_FN float bcos(float x) {                                                          // :274-304
    float periodCount = x * (float)(0.5 / 3.14159265358979323846);
    float periodFraction = periodCount - floorf(periodCount);
    float periodX = periodFraction * kTwoPi;
    float y;
    const float pi3Over2 = 3 * kPiOver2;
    y = sel(periodX > kPiOver2, kPi - periodX, periodX);
    y = sel(periodX > kPi, periodX - kPi, y);
    y = sel(periodX > pi3Over2, kTwoPi - periodX, y);
    float numerator = ((((-0.003436308368583229f * y + 0.021317031205957775f) * y + 0.06955843390178032f) * y - 0.4578088075324152f) * y - 0.15082367674208508f) * y + 1.0f;
    float denominator = ((((-0.00007650398834677185f * y + 0.0007451378206294365f) * y - 0.00585321045829395f) * y + 0.04219116713777847f) * y - 0.15082367538305258f) * y + 1.0f;
    float result = numerator / denominator;
    return sel((periodX > kPiOver2) && (periodX < pi3Over2), -result, result);
}
BD_FN float bsin(float x) {                                                          // :311-343
    float periodCount = x * (float)(0.5 / 3.14159265358979323846);
    float periodFraction = periodCount - floorf(periodCount);
    float periodX = periodFraction * kTwoPi;
    float y;
    y = sel(periodX > kPiOver2, kPi - periodX, periodX);
    bool inSecondHalf = periodX > kPi;
    y = sel(inSecondHalf, periodX - kPi, y);
    y = sel(periodX > (3 * kPiOver2), kTwoPi - periodX, y);
    float numerator = ((((0.0040507708755727605f * y - 0.006685815219853882f) * y - 0.13993701695343166f) * y + 0.06174562337697123f) * y + 1.00000000151466040f) * y;
    float denominator = ((((0.00009018370615921334f * y + 0.0001700784176413186f) * y + 0.003606014457152456f) * y + 0.02672943625500751f) * y + 0.061745651499203795f) * y + 1.0f;
    float result = numerator / denominator;
    return sel(inSecondHalf, -result, result);
}
BD_FN float bacos(float x) {                                                         // :353-368
    bool negativeInput = x < 0.0f;
    x = vmin(1.0f, vabs(x));
    float numerator = sqrtf(1.0f - x) * (62.95741097600742f + x * (69.6550664543659f + x * (17.54512349463405f + x * 0.6022076120669532f)));
    float denominator = 40.07993264439811f + x * (49.81949855726789f + x * (15.703851745284796f + x));
    float result = numerator / denominator;
    return sel(negativeInput, kPi - result, result);
}
BD_FN float signedAngleDifference(float a, float b) {                                // :371-376
    const float half = 0.5f;
    float x = (b - a) * (1.0f / kTwoPi) + half;
    return (x - floorf(x) - half) * kTwoPi;
}

// ---- QuaternionWide (BepuUtilities/QuaternionWide.cs) ----
BD_FN Q normalize(Q q) {                                                             // :124-134
    float inverseNorm = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return {q.x * inverseNorm, q.y * inverseNorm, q.z * inverseNorm, q.w * inverseNorm};
}
BD_FN Q concatenate(Q a, Q b) {                                                      // :500-506
    Q r;
    r.x = a.w * b.x + a.x * b.w + a.z * b.y - a.y * b.z;
    r.y = a.w * b.y + a.y * b.w + a.x * b.z - a.z * b.x;
    r.z = a.w * b.z + a.z * b.w + a.y * b.x - a.x * b.y;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return r;
}
BD_FN V3 transform(V3 v, Q r) {                                                      // :252-275
    float x2 = r.x + r.x, y2 = r.y + r.y, z2 = r.z + r.z;
    float xx2 = r.x * x2, xy2 = r.x * y2, xz2 = r.x * z2;
    float yy2 = r.y * y2, yz2 = r.y * z2, zz2 = r.z * z2;
    float wx2 = r.w * x2, wy2 = r.w * y2, wz2 = r.w * z2;
    V3 o;
    o.x = v.x * (1.0f - yy2 - zz2) + v.y * (xy2 - wz2) + v.z * (xz2 + wy2);
    o.y = v.x * (xy2 + wz2) + v.y * (1.0f - xx2 - zz2) + v.z * (yz2 - wx2);
    o.z = v.x * (xz2 - wy2) + v.y * (yz2 + wx2) + v.z * (1.0f - xx2 - yy2);
    return o;
}
BD_FN V3 transformUnitZ(Q r) {                                                       // :413-430
    float x2 = r.x + r.x, y2 = r.y + r.y, z2 = r.z + r.z;
    float xx2 = r.x * x2, xz2 = r.x * z2, yy2 = r.y * y2, yz2 = r.y * z2;
    float wx2 = r.w * x2, wy2 = r.w * y2;
    (void)z2;
    return {xz2 + wy2, yz2 - wx2, 1.0f - xx2 - yy2};
}
BD_FN void transformUnitXZ(Q r, V3& x, V3& z) {                                      // :467-490
    float qX2 = r.x + r.x, qY2 = r.y + r.y, qZ2 = r.z + r.z;
    float YY = qY2 * r.y, ZZ = qZ2 * r.z;
    x.x = 1.0f - YY - ZZ;
    float XY = qX2 * r.y, ZW = qZ2 * r.w;
    x.y = XY + ZW;
    float XZ = qX2 * r.z, YW = qY2 * r.w;
    x.z = XZ - YW;
    float XX = qX2 * r.x, XW = qX2 * r.w, YZ = qY2 * r.z;
    z.x = XZ + YW;
    z.y = YZ - XW;
    z.z = 1.0f - XX - YY;
}
BD_FN Q quaternionBetweenNormalizedVectors(V3 v1, V3 v2) {                           // :162-188
    float d = dot(v1, v2);
    V3 c = cross(v1, v2);
    bool useNormalCase = d > -0.999999f;
    float absX = vabs(v1.x), absY = vabs(v1.y), absZ = vabs(v1.z);
    bool xIsSmallest = (absX < absY) && (absX < absZ);
    bool yIsSmaller = absY < absZ;
    Q q;
    q.x = sel(useNormalCase, c.x, sel(xIsSmallest, 0.0f, sel(yIsSmaller, -v1.z, -v1.y)));
    q.y = sel(useNormalCase, c.y, sel(xIsSmallest, -v1.z, sel(yIsSmaller, 0.0f, v1.x)));
    q.z = sel(useNormalCase, c.z, sel(xIsSmallest, v1.y, sel(yIsSmaller, v1.x, 0.0f)));
    q.w = sel(useNormalCase, d + 1.0f, 0.0f);
    return normalize(q);
}

// ---- Matrix3x3Wide (BepuUtilities/Matrix3x3Wide.cs) ----
BD_FN M3 createFromQuaternion(Q q) {                                                 // :238-265
    float qX2 = q.x + q.x, qY2 = q.y + q.y, qZ2 = q.z + q.z;
    M3 r;
    float YY = qY2 * q.y, ZZ = qZ2 * q.z;
    r.X.x = 1.0f - YY - ZZ;
    float XY = qX2 * q.y, ZW = qZ2 * q.w;
    r.X.y = XY + ZW;
    float XZ = qX2 * q.z, YW = qY2 * q.w;
    r.X.z = XZ - YW;
    float XX = qX2 * q.x;
    r.Y.x = XY - ZW;
    r.Y.y = 1.0f - XX - ZZ;
    float XW = qX2 * q.w, YZ = qY2 * q.z;
    r.Y.z = YZ + XW;
    r.Z.x = XZ + YW;
    r.Z.y = YZ - XW;
    r.Z.z = 1.0f - XX - YY;
    return r;
}
BD_FN V3 transform(V3 v, const M3& m) {                                              // :109-114
    return {v.x * m.X.x + v.y * m.Y.x + v.z * m.Z.x,
            v.x * m.X.y + v.y * m.Y.y + v.z * m.Z.y,
            v.x * m.X.z + v.y * m.Y.z + v.z * m.Z.z};
}

// ---- Symmetric3x3Wide (BepuUtilities/Symmetric3x3Wide.cs) ----
BD_FN Sym3 invert(const Sym3& m) {                                                   // :42-60
    float xx = m.yy * m.zz - m.zy * m.zy;
    float yx = m.zy * m.zx - m.zz * m.yx;
    float zx = m.yx * m.zy - m.zx * m.yy;
    float determinantInverse = 1.0f / (xx * m.xx + yx * m.yx + zx * m.zx);
    float yy = m.zz * m.xx - m.zx * m.zx;
    float zy = m.zx * m.yx - m.xx * m.zy;
    float zz = m.xx * m.yy - m.yx * m.yx;
    Sym3 r;
    r.xx = xx * determinantInverse; r.yx = yx * determinantInverse; r.zx = zx * determinantInverse;
    r.yy = yy * determinantInverse; r.zy = zy * determinantInverse; r.zz = zz * determinantInverse;
    return r;
}
BD_FN Sym3 add(const Sym3& a, const Sym3& b) {                                       // :69-77
    return {a.xx + b.xx, a.yx + b.yx, a.yy + b.yy, a.zx + b.zx, a.zy + b.zy, a.zz + b.zz};
}
BD_FN Sym3 sub(const Sym3& a, const Sym3& b) {                                       // :105-113
    return {a.xx - b.xx, a.yx - b.yx, a.yy - b.yy, a.zx - b.zx, a.zy - b.zy, a.zz - b.zz};
}
BD_FN Sym3 scale(const Sym3& m, float s) {                                           // :143-151
    return {m.xx * s, m.yx * s, m.yy * s, m.zx * s, m.zy * s, m.zz * s};
}
BD_FN Sym3 skewSandwich(V3 v, const Sym3& m) {                                       // :182-206
    float xzy = v.x * m.zy, yzx = v.y * m.zx, zyx = v.z * m.yx;
    float ixy = v.y * m.zy - v.z * m.yy;
    float ixz = v.y * m.zz - v.z * m.zy;
    float iyx = v.z * m.xx - v.x * m.zx;
    float iyy = zyx - xzy;
    float iyz = v.z * m.zx - v.x * m.zz;
    float izx = v.x * m.yx - v.y * m.xx;
    float izy = v.x * m.yy - v.y * m.yx;
    float izz = xzy - yzx;
    Sym3 s;
    s.xx = v.y * ixz - v.z * ixy;
    s.yx = v.y * iyz - v.z * iyy;
    s.yy = v.z * iyx - v.x * iyz;
    s.zx = v.y * izz - v.z * izy;
    s.zy = v.z * izx - v.x * izz;
    s.zz = v.x * izy - v.y * izx;
    return s;
}
BD_FN float vectorSandwich(V3 v, const Sym3& m) {                                    // :209-217
    float x = v.x * m.xx + v.y * m.yx + v.z * m.zx;
    float y = v.x * m.yx + v.y * m.yy + v.z * m.zy;
    float z = v.x * m.zx + v.y * m.zy + v.z * m.zz;
    return x * v.x + y * v.y + z * v.z;
}
BD_FN Sym3 rotationSandwich(const M3& r, const Sym3& m) {                            // :231-257
    float ixx = r.X.x * m.xx + r.Y.x * m.yx + r.Z.x * m.zx;
    float ixy = r.X.x * m.yx + r.Y.x * m.yy + r.Z.x * m.zy;
    float ixz = r.X.x * m.zx + r.Y.x * m.zy + r.Z.x * m.zz;
    float iyx = r.X.y * m.xx + r.Y.y * m.yx + r.Z.y * m.zx;
    float iyy = r.X.y * m.yx + r.Y.y * m.yy + r.Z.y * m.zy;
    float iyz = r.X.y * m.zx + r.Y.y * m.zy + r.Z.y * m.zz;
    float izx = r.X.z * m.xx + r.Y.z * m.yx + r.Z.z * m.zx;
    float izy = r.X.z * m.yx + r.Y.z * m.yy + r.Z.z * m.zy;
    float izz = r.X.z * m.zx + r.Y.z * m.zy + r.Z.z * m.zz;
    Sym3 s;
    s.xx = ixx * r.X.x + ixy * r.Y.x + ixz * r.Z.x;
    s.yx = iyx * r.X.x + iyy * r.Y.x + iyz * r.Z.x;
    s.yy = iyx * r.X.y + iyy * r.Y.y + iyz * r.Z.y;
    s.zx = izx * r.X.x + izy * r.Y.x + izz * r.Z.x;
    s.zy = izx * r.X.y + izy * r.Y.y + izz * r.Z.y;
    s.zz = izx * r.X.z + izy * r.Y.z + izz * r.Z.z;
    return s;
}
BD_FN M23 multiply(const M23& a, const Sym3& b) {                                    // :260-268
    M23 r;
    r.X.x = a.X.x * b.xx + a.X.y * b.yx + a.X.z * b.zx;
    r.X.y = a.X.x * b.yx + a.X.y * b.yy + a.X.z * b.zy;
    r.X.z = a.X.x * b.zx + a.X.y * b.zy + a.X.z * b.zz;
    r.Y.x = a.Y.x * b.xx + a.Y.y * b.yx + a.Y.z * b.zx;
    r.Y.y = a.Y.x * b.yx + a.Y.y * b.yy + a.Y.z * b.zy;
    r.Y.z = a.Y.x * b.zx + a.Y.y * b.zy + a.Y.z * b.zz;
    return r;
}
BD_FN M23 multiplyByTransposed(const Sym3& a, const M23& b) {                        // :375-385
    M23 r;
    r.X.x = a.xx * b.X.x + a.yx * b.X.y + a.zx * b.X.z;
    r.Y.x = a.xx * b.Y.x + a.yx * b.Y.y + a.zx * b.Y.z;
    r.X.y = a.yx * b.X.x + a.yy * b.X.y + a.zy * b.X.z;
    r.Y.y = a.yx * b.Y.x + a.yy * b.Y.y + a.zy * b.Y.z;
    r.X.z = a.zx * b.X.x + a.zy * b.X.y + a.zz * b.X.z;
    r.Y.z = a.zx * b.Y.x + a.zy * b.Y.y + a.zz * b.Y.z;
    return r;
}
BD_FN Sym2 matrixSandwich(const M23& m, const Sym3& t) {                             // :388-400
    float ixx = m.X.x * t.xx + m.X.y * t.yx + m.X.z * t.zx;
    float ixy = m.X.x * t.yx + m.X.y * t.yy + m.X.z * t.zy;
    float ixz = m.X.x * t.zx + m.X.y * t.zy + m.X.z * t.zz;
    float iyx = m.Y.x * t.xx + m.Y.y * t.yx + m.Y.z * t.zx;
    float iyy = m.Y.x * t.yx + m.Y.y * t.yy + m.Y.z * t.zy;
    float iyz = m.Y.x * t.zx + m.Y.y * t.zy + m.Y.z * t.zz;
    Sym2 r;
    r.xx = ixx * m.X.x + ixy * m.X.y + ixz * m.X.z;
    r.yx = iyx * m.X.x + iyy * m.X.y + iyz * m.X.z;
    r.yy = iyx * m.Y.x + iyy * m.Y.y + iyz * m.Y.z;
    return r;
}
// CompleteMatrixSandwich(Matrix2x3Wide a, Matrix2x3Wide b) -> Symmetric3x3Wide: a^T * b   // :418-428
BD_FN Sym3 completeMatrixSandwich3(const M23& a, const M23& b) {
    Sym3 r;
    r.xx = a.X.x * b.X.x + a.Y.x * b.Y.x;
    r.yx = a.X.y * b.X.x + a.Y.y * b.Y.x;
    r.yy = a.X.y * b.X.y + a.Y.y * b.Y.y;
    r.zx = a.X.z * b.X.x + a.Y.z * b.Y.x;
    r.zy = a.X.z * b.X.y + a.Y.z * b.Y.y;
    r.zz = a.X.z * b.X.z + a.Y.z * b.Y.z;
    return r;
}
BD_FN V3 transform(V3 v, const Sym3& m) {                                            // :457-462
    return {v.x * m.xx + v.y * m.yx + v.z * m.zx,
            v.x * m.yx + v.y * m.yy + v.z * m.zy,
            v.x * m.zx + v.y * m.zy + v.z * m.zz};
}

// ---- Matrix2x3Wide (BepuUtilities/Matrix2x3Wide.cs) ----
BD_FN V2 transformByTranspose(V3 v, const M23& m) {                                  // :58-62
    return {v.x * m.X.x + v.y * m.X.y + v.z * m.X.z, v.x * m.Y.x + v.y * m.Y.y + v.z * m.Y.z};
}
BD_FN V3 transform(V2 v, const M23& m) {                                             // :82-87
    return {v.x * m.X.x + v.y * m.Y.x, v.x * m.X.y + v.y * m.Y.y, v.x * m.X.z + v.y * m.Y.z};
}
BD_FN M23 negate(const M23& m) { return {neg(m.X), neg(m.Y)}; }                       // :65-69

// ---- Symmetric2x2Wide (BepuUtilities/Symmetric2x2Wide.cs) ----
BD_FN Sym2 sandwichScale(const M23& m, float s) {                                    // :23-29
    Sym2 r;
    r.xx = s * (m.X.x * m.X.x + m.X.y * m.X.y + m.X.z * m.X.z);
    r.yx = s * (m.Y.x * m.X.x + m.Y.y * m.X.y + m.Y.z * m.X.z);
    r.yy = s * (m.Y.x * m.Y.x + m.Y.y * m.Y.y + m.Y.z * m.Y.z);
    return r;
}
BD_FN Sym2 add(const Sym2& a, const Sym2& b) { return {a.xx + b.xx, a.yx + b.yx, a.yy + b.yy}; }  // :40-45
BD_FN Sym2 invert(const Sym2& m) {                                                   // :55-61
    float denom = 1.0f / (m.yx * m.yx - m.xx * m.yy);
    return {-m.yy * denom, m.yx * denom, -m.xx * denom};
}
BD_FN V2 transform(V2 v, const Sym2& m) {                                            // :64-68
    return {v.x * m.xx + v.y * m.yx, v.x * m.yx + v.y * m.yy};
}
BD_FN M23 multiplyTransposed(const M23& a, const Sym2& b) {                          // :71-79
    M23 r;
    r.X.x = a.X.x * b.xx + a.Y.x * b.yx;
    r.X.y = a.X.y * b.xx + a.Y.y * b.yx;
    r.X.z = a.X.z * b.xx + a.Y.z * b.yx;
    r.Y.x = a.X.x * b.yx + a.Y.x * b.yy;
    r.Y.y = a.X.y * b.yx + a.Y.y * b.yy;
    r.Y.z = a.X.z * b.yx + a.Y.z * b.yy;
    return r;
}
// CompleteMatrixSandwich(Matrix2x3Wide a, Matrix2x3Wide b) -> Symmetric2x2Wide: a * b^T   // :82-87
BD_FN Sym2 completeMatrixSandwich2(const M23& a, const M23& b) {
    Sym2 r;
    r.xx = a.X.x * b.X.x + a.X.y * b.X.y + a.X.z * b.X.z;
    r.yx = a.Y.x * b.X.x + a.Y.y * b.X.y + a.Y.z * b.X.z;
    r.yy = a.Y.x * b.Y.x + a.Y.y * b.Y.y + a.Y.z * b.Y.z;
    return r;
}

// ---- Symmetric4x4Wide (BepuUtilities/Symmetric4x4Wide.cs) ----
BD_FN Sym4 invert(const Sym4& m) {                                                   // :62-94
    float s0 = m.xx * m.yy - m.yx * m.yx;
    float s1 = m.xx * m.zy - m.yx * m.zx;
    float s2 = m.xx * m.wy - m.yx * m.wx;
    float s3 = m.yx * m.zy - m.yy * m.zx;
    float s4 = m.yx * m.wy - m.yy * m.wx;
    float s5 = m.zx * m.wy - m.zy * m.wx;
    float c5 = m.zz * m.ww - m.wz * m.wz;
    float c4 = m.zy * m.ww - m.wy * m.wz;
    float c3 = m.zy * m.wz - m.wy * m.zz;
    float c2 = m.zx * m.ww - m.wx * m.wz;
    float c1 = m.zx * m.wz - m.wx * m.zz;
    float inverseDeterminant = 1.0f / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * s5);
    Sym4 r;
    r.xx = (m.yy * c5 - m.zy * c4 + m.wy * c3) * inverseDeterminant;
    r.yx = (-m.yx * c5 + m.zy * c2 - m.wy * c1) * inverseDeterminant;
    r.yy = (m.xx * c5 - m.zx * c2 + m.wx * c1) * inverseDeterminant;
    r.zx = (m.yx * c4 - m.yy * c2 + m.wy * s5) * inverseDeterminant;
    r.zy = (-m.xx * c4 + m.yx * c2 - m.wx * s5) * inverseDeterminant;
    r.zz = (m.wx * s4 - m.wy * s2 + m.ww * s0) * inverseDeterminant;
    r.wx = (-m.yx * c3 + m.yy * c1 - m.zy * s5) * inverseDeterminant;
    r.wy = (m.xx * c3 - m.yx * c1 + m.zx * s5) * inverseDeterminant;
    r.wz = (-m.wx * s3 + m.wy * s1 - m.wz * s0) * inverseDeterminant;
    r.ww = (m.zx * s3 - m.zy * s1 + m.zz * s0) * inverseDeterminant;
    return r;
}
BD_FN V4 transform(V4 v, const Sym4& m) {                                            // :97-103
    V4 r;
    r.x = v.x * m.xx + v.y * m.yx + v.z * m.zx + v.w * m.wx;
    r.y = v.x * m.yx + v.y * m.yy + v.z * m.zy + v.w * m.wy;
    r.z = v.x * m.zx + v.y * m.zy + v.z * m.zz + v.w * m.wz;
    r.w = v.x * m.wx + v.y * m.wy + v.z * m.wz + v.w * m.ww;
    return r;
}

// ---- Symmetric5x5Wide (BepuUtilities/Symmetric5x5Wide.cs) ----
BD_FN Sym5 invert5(const Sym3& a, const M23& b, const Sym2& d) {                     // :36-49
    Sym5 result;
    Sym2 invD = invert(d);
    M23 bTInvD = multiplyTransposed(b, invD);
    Sym3 bTInvDB = completeMatrixSandwich3(bTInvD, b);
    Sym3 resultAInverse = sub(a, bTInvDB);
    result.A = invert(resultAInverse);
    M23 negatedResultBT = multiplyByTransposed(result.A, bTInvD);
    result.B = negate(negatedResultBT);
    result.D = completeMatrixSandwich2(bTInvD, negatedResultBT);
    result.D = add(result.D, invD);
    return result;
}
BD_FN void transform5(V3 v0, V2 v1, const Sym5& m, V3& r0, V2& r1) {                 // :59-68
    r0.x = v0.x * m.A.xx + v0.y * m.A.yx + v0.z * m.A.zx + v1.x * m.B.X.x + v1.y * m.B.Y.x;
    r0.y = v0.x * m.A.yx + v0.y * m.A.yy + v0.z * m.A.zy + v1.x * m.B.X.y + v1.y * m.B.Y.y;
    r0.z = v0.x * m.A.zx + v0.y * m.A.zy + v0.z * m.A.zz + v1.x * m.B.X.z + v1.y * m.B.Y.z;
    r1.x = v0.x * m.B.X.x + v0.y * m.B.X.y + v0.z * m.B.X.z + v1.x * m.D.xx + v1.y * m.D.yx;
    r1.y = v0.x * m.B.Y.x + v0.y * m.B.Y.y + v0.z * m.B.Y.z + v1.x * m.D.yx + v1.y * m.D.yy;
}

// ---- BepuPhysics/Helpers.cs:21-47 ----
BD_FN void buildOrthonormalBasis(V3 n, V3& t1, V3& t2) {                             // :21-35
    float sign = sel(n.z < 0.0f, -1.0f, 1.0f);
    float scale = -1.0f / (sign + n.z);
    t1.x = n.x * n.y * scale;
    t1.y = sign + n.y * n.y * scale;
    t1.z = -n.y;
    t2.x = 1.0f + sign * n.x * n.x * scale;
    t2.y = sign * t1.x;
    t2.z = -sign * n.x;
}
BD_FN V3 findPerpendicular(V3 n) {                                                   // :38-47
    float sign = sel(n.z < 0.0f, -1.0f, 1.0f);
    float scale = -1.0f / (sign + n.z);
    return {n.x * n.y * scale, sign + n.y * n.y * scale, -n.y};
}

// ---- BepuPhysics/Constraints/SpringSettings.cs:37-55 ----
BD_FN void computeSpringiness(float angularFrequency, float twiceDampingRatio, float dt,
                                      float& positionErrorToVelocity, float& effectiveMassCFMScale, float& softnessImpulseScale) {
    float angularFrequencyDt = angularFrequency * dt;
    positionErrorToVelocity = angularFrequency / (angularFrequencyDt + twiceDampingRatio);
    float extra = 1.0f / (angularFrequencyDt * (angularFrequencyDt + twiceDampingRatio));
    effectiveMassCFMScale = 1.0f / (1.0f + extra);
    softnessImpulseScale = extra * effectiveMassCFMScale;
}

// ---- BepuPhysics/PoseIntegrator.cs:146-175 ----
BD_FN Q integrateOrientation(Q start, V3 angularVelocity, float halfDt) {            // :146-164
    float speed = length(angularVelocity);
    float halfAngle = speed * halfDt;
    Q q;
    float s = bsin(halfAngle);
    float scale = s / speed;
    q.x = angularVelocity.x * scale;
    q.y = angularVelocity.y * scale;
    q.z = angularVelocity.z * scale;
    q.w = bcos(halfAngle);
    Q end = normalize(concatenate(start, q));
    bool speedValid = speed > 1e-15f;
    return speedValid ? end : start;
}
BD_FN Sym3 rotateInverseInertia(const Sym3& local, Q orientation) {                  // :167-175
    M3 m = createFromQuaternion(orientation);
    return rotationSandwich(m, local);
}

// ---- momentum-conserving angular integration (BepuPhysics/PoseIntegrator.cs:176-253) ----
BD_FN V3 transformByTransposed(V3 v, const M3& m) {                                        // Matrix3x3Wide.cs:127-132
    return {v.x * m.X.x + v.y * m.X.y + v.z * m.X.z, v.x * m.Y.x + v.y * m.Y.y + v.z * m.Y.z, v.x * m.Z.x + v.y * m.Z.y + v.z * m.Z.z};
}
BD_FN M3 invert(const M3& m) {                                                             // Matrix3x3Wide.cs:142-166
    float m11 = m.Y.y * m.Z.z - m.Z.y * m.Y.z;
    float m21 = m.Y.z * m.Z.x - m.Z.z * m.Y.x;
    float m31 = m.Y.x * m.Z.y - m.Z.x * m.Y.y;
    float determinantInverse = 1.0f / (m11 * m.X.x + m21 * m.X.y + m31 * m.X.z);
    float m12 = m.Z.y * m.X.z - m.X.y * m.Z.z;
    float m22 = m.Z.z * m.X.x - m.X.z * m.Z.x;
    float m32 = m.Z.x * m.X.y - m.X.x * m.Z.y;
    float m13 = m.X.y * m.Y.z - m.Y.y * m.X.z;
    float m23 = m.X.z * m.Y.x - m.Y.z * m.X.x;
    float m33 = m.X.x * m.Y.y - m.Y.x * m.X.y;
    M3 inverse;
    inverse.X = {m11 * determinantInverse, m12 * determinantInverse, m13 * determinantInverse};
    inverse.Y = {m21 * determinantInverse, m22 * determinantInverse, m23 * determinantInverse};
    inverse.Z = {m31 * determinantInverse, m32 * determinantInverse, m33 * determinantInverse};
    return inverse;
}
BD_FN M3 createCrossProduct(V3 v) {                                                        // Matrix3x3Wide.cs:169-180
    M3 skew;
    skew.X = {0.0f, -v.z, v.y};
    skew.Y = {v.z, 0.0f, -v.x};
    skew.Z = {-v.y, v.x, 0.0f};
    return skew;
}
BD_FN M3 multiply(const M3& a, const Sym3& b) {                                            // Symmetric3x3Wide.cs:319-334
    M3 r;
    r.X = {a.X.x * b.xx + a.X.y * b.yx + a.X.z * b.zx, a.X.x * b.yx + a.X.y * b.yy + a.X.z * b.zy, a.X.x * b.zx + a.X.y * b.zy + a.X.z * b.zz};
    r.Y = {a.Y.x * b.xx + a.Y.y * b.yx + a.Y.z * b.zx, a.Y.x * b.yx + a.Y.y * b.yy + a.Y.z * b.zy, a.Y.x * b.zx + a.Y.y * b.zy + a.Y.z * b.zz};
    r.Z = {a.Z.x * b.xx + a.Z.y * b.yx + a.Z.z * b.zx, a.Z.x * b.yx + a.Z.y * b.yy + a.Z.z * b.zy, a.Z.x * b.zx + a.Z.y * b.zy + a.Z.z * b.zz};
    return r;
}
BD_FN V3 fallbackIfInertiaIncompatible(V3 previousAngularVelocity, V3 angularVelocity) {   // PoseIntegrator.cs:179-190
    const float infinity = __builtin_inff();
    bool useNewVelocity = (vabs(angularVelocity.x) < infinity) && (vabs(angularVelocity.y) < infinity) && (vabs(angularVelocity.z) < infinity);
    return sel3(useNewVelocity, angularVelocity, previousAngularVelocity);
}
BD_FN V3 integrateAngularVelocityConserveMomentum(Q previousOrientation, const Sym3& localInverseInertia, const Sym3& worldInverseInertia, V3 angularVelocity) {  // :192-207
    M3 previousOrientationMatrix = createFromQuaternion(previousOrientation);
    V3 localPreviousAngularVelocity = transformByTransposed(angularVelocity, previousOrientationMatrix);
    Sym3 localInertiaTensor = invert(localInverseInertia);
    V3 localAngularMomentum = transform(localPreviousAngularVelocity, localInertiaTensor);
    V3 angularMomentum = transform(localAngularMomentum, previousOrientationMatrix);
    V3 newVelocity = transform(angularMomentum, worldInverseInertia);
    return fallbackIfInertiaIncompatible(angularVelocity, newVelocity);
}
BD_FN V3 integrateAngularVelocityConserveMomentumWithGyroscopicTorque(Q orientation, const Sym3& localInverseInertia, V3 angularVelocity, float dt) {  // :209-253
    M3 orientationMatrix = createFromQuaternion(orientation);
    V3 localAngularVelocity = transformByTransposed(angularVelocity, orientationMatrix);
    Sym3 localInertiaTensor = invert(localInverseInertia);
    V3 localAngularMomentum = transform(localAngularVelocity, localInertiaTensor);
    V3 c = cross(localAngularMomentum, localAngularVelocity);
    V3 residual = {dt * c.x, dt * c.y, dt * c.z};
    M3 skewMomentum = createCrossProduct(localAngularMomentum);
    M3 skewVelocity = createCrossProduct(localAngularVelocity);
    M3 transformedSkewVelocity = multiply(skewVelocity, localInertiaTensor);
    M3 changeOverDt;
    changeOverDt.X = sub(transformedSkewVelocity.X, skewMomentum.X);
    changeOverDt.Y = sub(transformedSkewVelocity.Y, skewMomentum.Y);
    changeOverDt.Z = sub(transformedSkewVelocity.Z, skewMomentum.Z);
    M3 change = {scale(changeOverDt.X, dt), scale(changeOverDt.Y, dt), scale(changeOverDt.Z, dt)};
    M3 jacobian;                                                                          // Symmetric3x3Wide.cs:539-552 (symmetric + general)
    jacobian.X = {localInertiaTensor.xx + change.X.x, localInertiaTensor.yx + change.X.y, localInertiaTensor.zx + change.X.z};
    jacobian.Y = {localInertiaTensor.yx + change.Y.x, localInertiaTensor.yy + change.Y.y, localInertiaTensor.zy + change.Y.z};
    jacobian.Z = {localInertiaTensor.zx + change.Z.x, localInertiaTensor.zy + change.Z.y, localInertiaTensor.zz + change.Z.z};
    M3 inverseJacobian = invert(jacobian);
    V3 newtonStep = transform(residual, inverseJacobian);
    localAngularVelocity = sub(localAngularVelocity, newtonStep);
    V3 newVelocity = transform(localAngularVelocity, orientationMatrix);
    return fallbackIfInertiaIncompatible(angularVelocity, newVelocity);
}

// ---- helpers of the Weld constraint ----
BD_FN Q conjugate(Q q) { return {q.x, q.y, q.z, -q.w}; }                                    // QuaternionWide.cs:546-552 (negates W)
BD_FN void getAxisAngleFromQuaternion(Q q, V3& axis, float& angle) {                       // QuaternionWide.cs:227-243
    bool shouldNegate = q.w < 0.0f;
    axis = {sel(shouldNegate, -q.x, q.x), sel(shouldNegate, -q.y, q.y), sel(shouldNegate, -q.z, q.z)};
    float qw = sel(shouldNegate, -q.w, q.w);
    float axisLength = length(axis);
    axis = scale(axis, 1.0f / axisLength);
    bool useFallback = axisLength < 1e-14f;
    axis = {sel(useFallback, 1.0f, axis.x), sel(useFallback, 0.0f, axis.y), sel(useFallback, 0.0f, axis.z)};
    float halfAngle = bacos(qw);
    angle = 2.0f * halfAngle;
}
BD_FN M3 multiply(const Sym3& a, const M3& b) {                                            // Symmetric3x3Wide.cs:343-356
    M3 r;
    r.X = {a.xx * b.X.x + a.yx * b.Y.x + a.zx * b.Z.x, a.xx * b.X.y + a.yx * b.Y.y + a.zx * b.Z.y, a.xx * b.X.z + a.yx * b.Y.z + a.zx * b.Z.z};
    r.Y = {a.yx * b.X.x + a.yy * b.Y.x + a.zy * b.Z.x, a.yx * b.X.y + a.yy * b.Y.y + a.zy * b.Z.y, a.yx * b.X.z + a.yy * b.Y.z + a.zy * b.Z.z};
    r.Z = {a.zx * b.X.x + a.zy * b.Y.x + a.zz * b.Z.x, a.zx * b.X.y + a.zy * b.Y.y + a.zz * b.Z.y, a.zx * b.X.z + a.zy * b.Y.z + a.zz * b.Z.z};
    return r;
}
BD_FN Sym3 completeMatrixSandwichTranspose(const M3& a, const M3& b) {                     // Symmetric3x3Wide.cs:508-518
    Sym3 r;
    r.xx = a.X.x * b.X.x + a.Y.x * b.Y.x + a.Z.x * b.Z.x;
    r.yx = a.X.y * b.X.x + a.Y.y * b.Y.x + a.Z.y * b.Z.x;
    r.yy = a.X.y * b.X.y + a.Y.y * b.Y.y + a.Z.y * b.Z.y;
    r.zx = a.X.z * b.X.x + a.Y.z * b.Y.x + a.Z.z * b.Z.x;
    r.zy = a.X.z * b.X.y + a.Y.z * b.Y.y + a.Z.z * b.Z.y;
    r.zz = a.X.z * b.X.z + a.Y.z * b.Y.z + a.Z.z * b.Z.z;
    return r;
}
// Symmetric6x6Wide.LDLTSolve (Symmetric6x6Wide.cs:84-129) in two halves: the factorisation of [a b^T; b d] depends on the matrix only, the
// substitution on the right-hand side; every operation and its order are the reference's.
struct LDLT6 { float inverseD1, inverseD2, inverseD3, inverseD4, inverseD5, inverseD6, l21, l31, l41, l51, l61, l32, l42, l52, l62, l43, l53, l63, l54, l64, l65; };
BD_FN LDLT6 ldltFactor(const Sym3& a, const M3& b, const Sym3& d) {
    LDLT6 f;
    float d1 = a.xx;
    f.inverseD1 = 1.0f / d1;
    f.l21 = f.inverseD1 * a.yx;
    f.l31 = f.inverseD1 * a.zx;
    f.l41 = f.inverseD1 * b.X.x;
    f.l51 = f.inverseD1 * b.X.y;
    f.l61 = f.inverseD1 * b.X.z;
    float d2 = a.yy - f.l21 * f.l21 * d1;
    f.inverseD2 = 1.0f / d2;
    f.l32 = f.inverseD2 * (a.zy - f.l31 * f.l21 * d1);
    f.l42 = f.inverseD2 * (b.Y.x - f.l41 * f.l21 * d1);
    f.l52 = f.inverseD2 * (b.Y.y - f.l51 * f.l21 * d1);
    f.l62 = f.inverseD2 * (b.Y.z - f.l61 * f.l21 * d1);
    float d3 = a.zz - f.l31 * f.l31 * d1 - f.l32 * f.l32 * d2;
    f.inverseD3 = 1.0f / d3;
    f.l43 = f.inverseD3 * (b.Z.x - f.l41 * f.l31 * d1 - f.l42 * f.l32 * d2);
    f.l53 = f.inverseD3 * (b.Z.y - f.l51 * f.l31 * d1 - f.l52 * f.l32 * d2);
    f.l63 = f.inverseD3 * (b.Z.z - f.l61 * f.l31 * d1 - f.l62 * f.l32 * d2);
    float d4 = d.xx - f.l41 * f.l41 * d1 - f.l42 * f.l42 * d2 - f.l43 * f.l43 * d3;
    f.inverseD4 = 1.0f / d4;
    f.l54 = f.inverseD4 * (d.yx - f.l51 * f.l41 * d1 - f.l52 * f.l42 * d2 - f.l53 * f.l43 * d3);
    f.l64 = f.inverseD4 * (d.zx - f.l61 * f.l41 * d1 - f.l62 * f.l42 * d2 - f.l63 * f.l43 * d3);
    float d5 = d.yy - f.l51 * f.l51 * d1 - f.l52 * f.l52 * d2 - f.l53 * f.l53 * d3 - f.l54 * f.l54 * d4;
    f.inverseD5 = 1.0f / d5;
    f.l65 = f.inverseD5 * (d.zy - f.l61 * f.l51 * d1 - f.l62 * f.l52 * d2 - f.l63 * f.l53 * d3 - f.l64 * f.l54 * d4);
    float d6 = d.zz - f.l61 * f.l61 * d1 - f.l62 * f.l62 * d2 - f.l63 * f.l63 * d3 - f.l64 * f.l64 * d4 - f.l65 * f.l65 * d5;
    f.inverseD6 = 1.0f / d6;
    return f;
}
BD_FN void ldltSubstitute(const LDLT6& f, V3 v0, V3 v1, V3& r0, V3& r1) {
    r0.x = v0.x;
    r0.y = v0.y - f.l21 * r0.x;
    r0.z = v0.z - f.l31 * r0.x - f.l32 * r0.y;
    r1.x = v1.x - f.l41 * r0.x - f.l42 * r0.y - f.l43 * r0.z;
    r1.y = v1.y - f.l51 * r0.x - f.l52 * r0.y - f.l53 * r0.z - f.l54 * r1.x;
    r1.z = v1.z - f.l61 * r0.x - f.l62 * r0.y - f.l63 * r0.z - f.l64 * r1.x - f.l65 * r1.y;
    r1.z = r1.z * f.inverseD6;
    r1.y = r1.y * f.inverseD5 - f.l65 * r1.z;
    r1.x = r1.x * f.inverseD4 - f.l64 * r1.z - f.l54 * r1.y;
    r0.z = r0.z * f.inverseD3 - f.l63 * r1.z - f.l53 * r1.y - f.l43 * r1.x;
    r0.y = r0.y * f.inverseD2 - f.l62 * r1.z - f.l52 * r1.y - f.l42 * r1.x - f.l32 * r0.z;
    r0.x = r0.x * f.inverseD1 - f.l61 * r1.z - f.l51 * r1.y - f.l41 * r1.x - f.l31 * r0.z - f.l21 * r0.y;
}

}  // namespace bd
