// oracle/wide — TEST INFRASTRUCTURE (parity unpinned, see wide_vec.h). BepuUtilities' *Wide math, transcribed bundle-for-bundle from the C# (file:line cited per type).
// `in` parameters are const references and `out`/`ref` parameters references, so argument aliasing behaves as in the C#.
#pragma once
#include "wide_vec.h"

namespace wide {

// BepuUtilities/MathHelper.cs:17-32
constexpr float Pi = 3.141592653589793239f;
constexpr float TwoPi = 6.283185307179586477f;
constexpr float PiOver2 = 1.570796326794896619f;

struct Vector2Wide {  // BepuUtilities/Vector2Wide.cs
    VF X, Y;
    static void Add(const Vector2Wide& a, const Vector2Wide& b, Vector2Wide& result) { result.X = a.X + b.X; result.Y = a.Y + b.Y; }
    static void Subtract(const Vector2Wide& a, const Vector2Wide& b, Vector2Wide& result) { result.X = a.X - b.X; result.Y = a.Y - b.Y; }  // :76
    static void Dot(const Vector2Wide& a, const Vector2Wide& b, VF& result) { result = a.X * b.X + a.Y * b.Y; }                            // :83
    static void Scale(const Vector2Wide& vector, const VF& scalar, Vector2Wide& result) { result.X = vector.X * scalar; result.Y = vector.Y * scalar; }  // :89
    static void Negate(const Vector2Wide& v, Vector2Wide& result) { result.X = neg(v.X); result.Y = neg(v.Y); }  // :113
    static void LengthSquared(const Vector2Wide& v, VF& lengthSquared) { lengthSquared = v.X * v.X + v.Y * v.Y; }  // :141
    static void Length(const Vector2Wide& v, VF& length) { length = SquareRoot(v.X * v.X + v.Y * v.Y); }            // :147
};

static inline Vector2Wide operator+(const Vector2Wide& a, const Vector2Wide& b) { return Vector2Wide{a.X + b.X, a.Y + b.Y}; }      // Vector2Wide.cs:39
static inline Vector2Wide operator*(const Vector2Wide& vector, const VF& scalar) { return Vector2Wide{vector.X * scalar, vector.Y * scalar}; }  // :96

struct Vector3Wide {  // BepuUtilities/Vector3Wide.cs
    VF X, Y, Z;
    static void Add(const Vector3Wide& a, const Vector3Wide& b, Vector3Wide& result) { result.X = a.X + b.X; result.Y = a.Y + b.Y; result.Z = a.Z + b.Z; }       // :55
    static void Subtract(const Vector3Wide& a, const Vector3Wide& b, Vector3Wide& result) { result.X = a.X - b.X; result.Y = a.Y - b.Y; result.Z = a.Z - b.Z; }  // :127
    static void Dot(const Vector3Wide& a, const Vector3Wide& b, VF& result) { result = a.X * b.X + a.Y * b.Y + a.Z * b.Z; }                                     // :201
    static VF Dot(const Vector3Wide& a, const Vector3Wide& b) { return a.X * b.X + a.Y * b.Y + a.Z * b.Z; }                                                      // :213
    static void Scale(const Vector3Wide& vector, const VF& scalar, Vector3Wide& result) {  // :343
        result.X = vector.X * scalar; result.Y = vector.Y * scalar; result.Z = vector.Z * scalar;
    }
    static void Negate(const Vector3Wide& v, Vector3Wide& result) { result.X = neg(v.X); result.Y = neg(v.Y); result.Z = neg(v.Z); }  // :433
    static void ConditionallyNegate(const VI& shouldNegate, Vector3Wide& v) {  // :475
        v.X = wide::ConditionalSelect(shouldNegate, neg(v.X), v.X);
        v.Y = wide::ConditionalSelect(shouldNegate, neg(v.Y), v.Y);
        v.Z = wide::ConditionalSelect(shouldNegate, neg(v.Z), v.Z);
    }
    static void CrossWithoutOverlap(const Vector3Wide& a, const Vector3Wide& b, Vector3Wide& result) {  // :519
        result.X = a.Y * b.Z - a.Z * b.Y;
        result.Y = a.Z * b.X - a.X * b.Z;
        result.Z = a.X * b.Y - a.Y * b.X;
    }
    static void Cross(const Vector3Wide& a, const Vector3Wide& b, Vector3Wide& result) {  // :534
        Vector3Wide temp;
        CrossWithoutOverlap(a, b, temp);
        result = temp;
    }
    static Vector3Wide Cross(const Vector3Wide& a, const Vector3Wide& b) {  // :547
        Vector3Wide result;
        result.X = a.Y * b.Z - a.Z * b.Y;
        result.Y = a.Z * b.X - a.X * b.Z;
        result.Z = a.X * b.Y - a.Y * b.X;
        return result;
    }
    static void LengthSquared(const Vector3Wide& v, VF& lengthSquared) { lengthSquared = v.X * v.X + v.Y * v.Y + v.Z * v.Z; }  // :562
    static void Length(const Vector3Wide& v, VF& length) { length = SquareRoot(v.X * v.X + v.Y * v.Y + v.Z * v.Z); }            // :573
    static VF Length(const Vector3Wide& v) { return SquareRoot(v.X * v.X + v.Y * v.Y + v.Z * v.Z); }                            // :595
    static void Distance(const Vector3Wide& a, const Vector3Wide& b, VF& distance) {                                             // :627
        VF x = b.X - a.X, y = b.Y - a.Y, z = b.Z - a.Z;
        distance = SquareRoot(x * x + y * y + z * z);
    }
    static void Normalize(const Vector3Wide& v, Vector3Wide& result) {  // :688
        VF length;
        Length(v, length);
        VF scale = kOne / length;
        Scale(v, scale, result);
    }
    static void ConditionalSelect(const VI& condition, const Vector3Wide& left, const Vector3Wide& right, Vector3Wide& result) {  // :716
        result.X = wide::ConditionalSelect(condition, left.X, right.X);
        result.Y = wide::ConditionalSelect(condition, left.Y, right.Y);
        result.Z = wide::ConditionalSelect(condition, left.Z, right.Z);
    }
    static Vector3Wide Broadcast(float x, float y, float z) { return Vector3Wide{vf(x), vf(y), vf(z)}; }  // :828
};
static inline Vector3Wide operator+(const Vector3Wide& a, const Vector3Wide& b) { return Vector3Wide{a.X + b.X, a.Y + b.Y, a.Z + b.Z}; }  // :81
static inline Vector3Wide operator-(const Vector3Wide& a, const Vector3Wide& b) { return Vector3Wide{a.X - b.X, a.Y - b.Y, a.Z - b.Z}; }  // :155
static inline Vector3Wide operator*(const Vector3Wide& v, const VF& s) { return Vector3Wide{v.X * s, v.Y * s, v.Z * s}; }                  // :374
static inline Vector3Wide operator*(const VF& s, const Vector3Wide& v) { return Vector3Wide{s * v.X, s * v.Y, s * v.Z}; }                  // :390 (scalar * component)
static inline Vector3Wide operator-(const Vector3Wide& v) { return Vector3Wide{neg(v.X), neg(v.Y), neg(v.Z)}; }                            // :460

struct QuaternionWide {  // BepuUtilities/QuaternionWide.cs
    VF X, Y, Z, W;
    static QuaternionWide Normalize(const QuaternionWide& q) {  // :124
        VF inverseNorm = kOne / SquareRoot(q.X * q.X + q.Y * q.Y + q.Z * q.Z + q.W * q.W);
        QuaternionWide normalized;
        normalized.X = q.X * inverseNorm;
        normalized.Y = q.Y * inverseNorm;
        normalized.Z = q.Z * inverseNorm;
        normalized.W = q.W * inverseNorm;
        return normalized;
    }
    static void GetQuaternionBetweenNormalizedVectors(const Vector3Wide& v1, const Vector3Wide& v2, QuaternionWide& q) {  // :162
        VF dot;
        Vector3Wide::Dot(v1, v2, dot);
        Vector3Wide cross;
        Vector3Wide::CrossWithoutOverlap(v1, v2, cross);
        VI useNormalCase = GreaterThan(dot, vf(-0.999999f));
        VF absX = Abs(v1.X);
        VF absY = Abs(v1.Y);
        VF absZ = Abs(v1.Z);
        VI xIsSmallest = BitwiseAnd(LessThan(absX, absY), LessThan(absX, absZ));
        VI yIsSmaller = LessThan(absY, absZ);
        q.X = wide::ConditionalSelect(useNormalCase, cross.X, wide::ConditionalSelect(xIsSmallest, kZero, wide::ConditionalSelect(yIsSmaller, neg(v1.Z), neg(v1.Y))));
        q.Y = wide::ConditionalSelect(useNormalCase, cross.Y, wide::ConditionalSelect(xIsSmallest, neg(v1.Z), wide::ConditionalSelect(yIsSmaller, kZero, v1.X)));
        q.Z = wide::ConditionalSelect(useNormalCase, cross.Z, wide::ConditionalSelect(xIsSmallest, v1.Y, wide::ConditionalSelect(yIsSmaller, v1.X, kZero)));
        q.W = wide::ConditionalSelect(useNormalCase, dot + kOne, kZero);
        q = Normalize(q);
    }
    static void TransformWithoutOverlap(const Vector3Wide& v, const QuaternionWide& rotation, Vector3Wide& result) {  // :252
        VF x2 = rotation.X + rotation.X;
        VF y2 = rotation.Y + rotation.Y;
        VF z2 = rotation.Z + rotation.Z;
        VF xx2 = rotation.X * x2;
        VF xy2 = rotation.X * y2;
        VF xz2 = rotation.X * z2;
        VF yy2 = rotation.Y * y2;
        VF yz2 = rotation.Y * z2;
        VF zz2 = rotation.Z * z2;
        VF wx2 = rotation.W * x2;
        VF wy2 = rotation.W * y2;
        VF wz2 = rotation.W * z2;
        result.X = v.X * (kOne - yy2 - zz2) + v.Y * (xy2 - wz2) + v.Z * (xz2 + wy2);
        result.Y = v.X * (xy2 + wz2) + v.Y * (kOne - xx2 - zz2) + v.Z * (yz2 - wx2);
        result.Z = v.X * (xz2 - wy2) + v.Y * (yz2 + wx2) + v.Z * (kOne - xx2 - yy2);
    }
    static Vector3Wide TransformUnitY(const QuaternionWide& rotation) {  // :389
        VF x2 = rotation.X + rotation.X;
        VF y2 = rotation.Y + rotation.Y;
        VF z2 = rotation.Z + rotation.Z;
        VF xx2 = rotation.X * x2;
        VF xy2 = rotation.X * y2;
        VF yz2 = rotation.Y * z2;
        VF zz2 = rotation.Z * z2;
        VF wx2 = rotation.W * x2;
        VF wz2 = rotation.W * z2;
        Vector3Wide result;
        result.X = xy2 - wz2;
        result.Y = kOne - xx2 - zz2;
        result.Z = yz2 + wx2;
        return result;
    }
    static Vector3Wide TransformUnitZ(const QuaternionWide& rotation) {  // :413
        VF x2 = rotation.X + rotation.X;
        VF y2 = rotation.Y + rotation.Y;
        VF z2 = rotation.Z + rotation.Z;
        VF xx2 = rotation.X * x2;
        VF xz2 = rotation.X * z2;
        VF yy2 = rotation.Y * y2;
        VF yz2 = rotation.Y * z2;
        VF wx2 = rotation.W * x2;
        VF wy2 = rotation.W * y2;
        Vector3Wide result;
        result.X = xz2 + wy2;
        result.Y = yz2 - wx2;
        result.Z = kOne - xx2 - yy2;
        return result;
    }
    static void TransformUnitXY(const QuaternionWide& rotation, Vector3Wide& x, Vector3Wide& y) {  // :438
        VF x2 = rotation.X + rotation.X;
        VF y2 = rotation.Y + rotation.Y;
        VF z2 = rotation.Z + rotation.Z;
        VF xx2 = rotation.X * x2;
        VF xy2 = rotation.X * y2;
        VF xz2 = rotation.X * z2;
        VF yy2 = rotation.Y * y2;
        VF yz2 = rotation.Y * z2;
        VF zz2 = rotation.Z * z2;
        VF wx2 = rotation.W * x2;
        VF wy2 = rotation.W * y2;
        VF wz2 = rotation.W * z2;
        x.X = kOne - yy2 - zz2;
        x.Y = xy2 + wz2;
        x.Z = xz2 - wy2;
        y.X = xy2 - wz2;
        y.Y = kOne - xx2 - zz2;
        y.Z = yz2 + wx2;
    }
    static void TransformUnitXZ(const QuaternionWide& rotation, Vector3Wide& x, Vector3Wide& z) {  // :467
        VF qX2 = rotation.X + rotation.X;
        VF qY2 = rotation.Y + rotation.Y;
        VF qZ2 = rotation.Z + rotation.Z;
        VF YY = qY2 * rotation.Y;
        VF ZZ = qZ2 * rotation.Z;
        x.X = kOne - YY - ZZ;
        VF XY = qX2 * rotation.Y;
        VF ZW = qZ2 * rotation.W;
        x.Y = XY + ZW;
        VF XZ = qX2 * rotation.Z;
        VF YW = qY2 * rotation.W;
        x.Z = XZ - YW;
        VF XX = qX2 * rotation.X;
        VF XW = qX2 * rotation.W;
        VF YZ = qY2 * rotation.Z;
        z.X = XZ + YW;
        z.Y = YZ - XW;
        z.Z = kOne - XX - YY;
    }
    static void ConcatenateWithoutOverlap(const QuaternionWide& a, const QuaternionWide& b, QuaternionWide& result) {  // :500
        result.X = a.W * b.X + a.X * b.W + a.Z * b.Y - a.Y * b.Z;
        result.Y = a.W * b.Y + a.Y * b.W + a.X * b.Z - a.Z * b.X;
        result.Z = a.W * b.Z + a.Z * b.W + a.Y * b.X - a.X * b.Y;
        result.W = a.W * b.W - a.X * b.X - a.Y * b.Y - a.Z * b.Z;
    }
    static void Conjugate(const QuaternionWide& quaternion, QuaternionWide& result) {  // :546
        result.X = quaternion.X;
        result.Y = quaternion.Y;
        result.Z = quaternion.Z;
        result.W = neg(quaternion.W);
    }
    static void ConditionalSelect(const VI& condition, const QuaternionWide& left, const QuaternionWide& right, QuaternionWide& result) {  // :571
        result.X = wide::ConditionalSelect(condition, left.X, right.X);
        result.Y = wide::ConditionalSelect(condition, left.Y, right.Y);
        result.Z = wide::ConditionalSelect(condition, left.Z, right.Z);
        result.W = wide::ConditionalSelect(condition, left.W, right.W);
    }
};

struct Symmetric3x3Wide;
struct Matrix3x3Wide {  // BepuUtilities/Matrix3x3Wide.cs
    Vector3Wide X, Y, Z;
    static void MultiplyWithoutOverlap(const Matrix3x3Wide& a, const Matrix3x3Wide& b, Matrix3x3Wide& result) {  // :50
        result.X.X = a.X.X * b.X.X + a.X.Y * b.Y.X + a.X.Z * b.Z.X;
        result.X.Y = a.X.X * b.X.Y + a.X.Y * b.Y.Y + a.X.Z * b.Z.Y;
        result.X.Z = a.X.X * b.X.Z + a.X.Y * b.Y.Z + a.X.Z * b.Z.Z;
        result.Y.X = a.Y.X * b.X.X + a.Y.Y * b.Y.X + a.Y.Z * b.Z.X;
        result.Y.Y = a.Y.X * b.X.Y + a.Y.Y * b.Y.Y + a.Y.Z * b.Z.Y;
        result.Y.Z = a.Y.X * b.X.Z + a.Y.Y * b.Y.Z + a.Y.Z * b.Z.Z;
        result.Z.X = a.Z.X * b.X.X + a.Z.Y * b.Y.X + a.Z.Z * b.Z.X;
        result.Z.Y = a.Z.X * b.X.Y + a.Z.Y * b.Y.Y + a.Z.Z * b.Z.Y;
        result.Z.Z = a.Z.X * b.X.Z + a.Z.Y * b.Y.Z + a.Z.Z * b.Z.Z;
    }
    static void MultiplyByTransposeWithoutOverlap(const Matrix3x3Wide& a, const Matrix3x3Wide& b, Matrix3x3Wide& result) {  // :93
        result.X.X = a.X.X * b.X.X + a.X.Y * b.X.Y + a.X.Z * b.X.Z;
        result.X.Y = a.X.X * b.Y.X + a.X.Y * b.Y.Y + a.X.Z * b.Y.Z;
        result.X.Z = a.X.X * b.Z.X + a.X.Y * b.Z.Y + a.X.Z * b.Z.Z;
        result.Y.X = a.Y.X * b.X.X + a.Y.Y * b.X.Y + a.Y.Z * b.X.Z;
        result.Y.Y = a.Y.X * b.Y.X + a.Y.Y * b.Y.Y + a.Y.Z * b.Y.Z;
        result.Y.Z = a.Y.X * b.Z.X + a.Y.Y * b.Z.Y + a.Y.Z * b.Z.Z;
        result.Z.X = a.Z.X * b.X.X + a.Z.Y * b.X.Y + a.Z.Z * b.X.Z;
        result.Z.Y = a.Z.X * b.Y.X + a.Z.Y * b.Y.Y + a.Z.Z * b.Y.Z;
        result.Z.Z = a.Z.X * b.Z.X + a.Z.Y * b.Z.Y + a.Z.Z * b.Z.Z;
    }
    static void TransformWithoutOverlap(const Vector3Wide& v, const Matrix3x3Wide& m, Vector3Wide& result) {  // :109
        result.X = v.X * m.X.X + v.Y * m.Y.X + v.Z * m.Z.X;
        result.Y = v.X * m.X.Y + v.Y * m.Y.Y + v.Z * m.Z.Y;
        result.Z = v.X * m.X.Z + v.Y * m.Y.Z + v.Z * m.Z.Z;
    }
    static void TransformByTransposedWithoutOverlap(const Vector3Wide& v, const Matrix3x3Wide& m, Vector3Wide& result) {  // :127
        result.X = v.X * m.X.X + v.Y * m.X.Y + v.Z * m.X.Z;
        result.Y = v.X * m.Y.X + v.Y * m.Y.Y + v.Z * m.Y.Z;
        result.Z = v.X * m.Z.X + v.Y * m.Z.Y + v.Z * m.Z.Z;
    }
    static void Transform(const Vector3Wide& v, const Matrix3x3Wide& m, Vector3Wide& result) {  // :135
        Vector3Wide temp;
        TransformWithoutOverlap(v, m, temp);
        result = temp;
    }
    static void Invert(const Matrix3x3Wide& m, Matrix3x3Wide& inverse) {  // :142
        VF m11 = m.Y.Y * m.Z.Z - m.Z.Y * m.Y.Z;
        VF m21 = m.Y.Z * m.Z.X - m.Z.Z * m.Y.X;
        VF m31 = m.Y.X * m.Z.Y - m.Z.X * m.Y.Y;
        VF determinantInverse = kOne / (m11 * m.X.X + m21 * m.X.Y + m31 * m.X.Z);
        VF m12 = m.Z.Y * m.X.Z - m.X.Y * m.Z.Z;
        VF m22 = m.Z.Z * m.X.X - m.X.Z * m.Z.X;
        VF m32 = m.Z.X * m.X.Y - m.X.X * m.Z.Y;
        VF m13 = m.X.Y * m.Y.Z - m.Y.Y * m.X.Z;
        VF m23 = m.X.Z * m.Y.X - m.Y.Z * m.X.X;
        VF m33 = m.X.X * m.Y.Y - m.Y.X * m.X.Y;
        inverse.X.X = m11 * determinantInverse;
        inverse.Y.X = m21 * determinantInverse;
        inverse.Z.X = m31 * determinantInverse;
        inverse.X.Y = m12 * determinantInverse;
        inverse.Y.Y = m22 * determinantInverse;
        inverse.Z.Y = m32 * determinantInverse;
        inverse.X.Z = m13 * determinantInverse;
        inverse.Y.Z = m23 * determinantInverse;
        inverse.Z.Z = m33 * determinantInverse;
    }
    static void CreateCrossProduct(const Vector3Wide& v, Matrix3x3Wide& skew) {  // :169
        skew.X.X = kZero;
        skew.X.Y = neg(v.Z);
        skew.X.Z = v.Y;
        skew.Y.X = v.Z;
        skew.Y.Y = kZero;
        skew.Y.Z = neg(v.X);
        skew.Z.X = neg(v.Y);
        skew.Z.Y = v.X;
        skew.Z.Z = kZero;
    }
    static void Scale(const Matrix3x3Wide& m, const VF& scale, Matrix3x3Wide& result) {  // :224
        result.X.X = m.X.X * scale; result.X.Y = m.X.Y * scale; result.X.Z = m.X.Z * scale;
        result.Y.X = m.Y.X * scale; result.Y.Y = m.Y.Y * scale; result.Y.Z = m.Y.Z * scale;
        result.Z.X = m.Z.X * scale; result.Z.Y = m.Z.Y * scale; result.Z.Z = m.Z.Z * scale;
    }
    static void CreateFromQuaternion(const QuaternionWide& quaternion, Matrix3x3Wide& result) {  // :238
        VF qX2 = quaternion.X + quaternion.X;
        VF qY2 = quaternion.Y + quaternion.Y;
        VF qZ2 = quaternion.Z + quaternion.Z;
        VF YY = qY2 * quaternion.Y;
        VF ZZ = qZ2 * quaternion.Z;
        result.X.X = kOne - YY - ZZ;
        VF XY = qX2 * quaternion.Y;
        VF ZW = qZ2 * quaternion.W;
        result.X.Y = XY + ZW;
        VF XZ = qX2 * quaternion.Z;
        VF YW = qY2 * quaternion.W;
        result.X.Z = XZ - YW;
        VF XX = qX2 * quaternion.X;
        result.Y.X = XY - ZW;
        result.Y.Y = kOne - XX - ZZ;
        VF XW = qX2 * quaternion.W;
        VF YZ = qY2 * quaternion.Z;
        result.Y.Z = YZ + XW;
        result.Z.X = XZ + YW;
        result.Z.Y = YZ - XW;
        result.Z.Z = kOne - XX - YY;
    }
    static void Subtract(const Matrix3x3Wide& a, const Matrix3x3Wide& b, Matrix3x3Wide& result) {  // :267
        result.X.X = a.X.X - b.X.X; result.X.Y = a.X.Y - b.X.Y; result.X.Z = a.X.Z - b.X.Z;
        result.Y.X = a.Y.X - b.Y.X; result.Y.Y = a.Y.Y - b.Y.Y; result.Y.Z = a.Y.Z - b.Y.Z;
        result.Z.X = a.Z.X - b.Z.X; result.Z.Y = a.Z.Y - b.Z.Y; result.Z.Z = a.Z.Z - b.Z.Z;
    }
};

struct Matrix2x3Wide {  // BepuUtilities/Matrix2x3Wide.cs
    Vector3Wide X, Y;
    static void TransformByTransposeWithoutOverlap(const Vector3Wide& v, const Matrix2x3Wide& m, Vector2Wide& result) {  // :76
        result.X = v.X * m.X.X + v.Y * m.X.Y + v.Z * m.X.Z;
        result.Y = v.X * m.Y.X + v.Y * m.Y.Y + v.Z * m.Y.Z;
    }
    static void Negate(const Matrix2x3Wide& m, Matrix2x3Wide& result) {  // :83
        Vector3Wide::Negate(m.X, result.X);
        Vector3Wide::Negate(m.Y, result.Y);
    }
    static void Scale(const Matrix2x3Wide& m, const VF& scale, Matrix2x3Wide& result) {  // :96
        result.X.X = m.X.X * scale; result.X.Y = m.X.Y * scale; result.X.Z = m.X.Z * scale;
        result.Y.X = m.Y.X * scale; result.Y.Y = m.Y.Y * scale; result.Y.Z = m.Y.Z * scale;
    }
    static void Transform(const Vector2Wide& v, const Matrix2x3Wide& m, Vector3Wide& result) {  // :107
        result.X = v.X * m.X.X + v.Y * m.Y.X;
        result.Y = v.X * m.X.Y + v.Y * m.Y.Y;
        result.Z = v.X * m.X.Z + v.Y * m.Y.Z;
    }
    static void Add(const Matrix2x3Wide& a, const Matrix2x3Wide& b, Matrix2x3Wide& result) {  // :115
        Vector3Wide::Add(a.X, b.X, result.X);
        Vector3Wide::Add(a.Y, b.Y, result.Y);
    }
};

struct Symmetric2x2Wide {  // BepuUtilities/Symmetric2x2Wide.cs
    VF XX, YX, YY;
    static void Scale(const Symmetric2x2Wide& t, const VF& scale, Symmetric2x2Wide& result) {  // :31
        result.XX = t.XX * scale; result.YX = t.YX * scale; result.YY = t.YY * scale;
    }
    static void Add(const Symmetric2x2Wide& a, const Symmetric2x2Wide& b, Symmetric2x2Wide& result) {  // :39
        result.XX = a.XX + b.XX; result.YX = a.YX + b.YX; result.YY = a.YY + b.YY;
    }
    static void InvertWithoutOverlap(const Symmetric2x2Wide& m, Symmetric2x2Wide& inverse) {  // :55
        VF denom = kOne / (m.YX * m.YX - m.XX * m.YY);
        inverse.XX = neg(m.YY) * denom;
        inverse.YX = m.YX * denom;
        inverse.YY = neg(m.XX) * denom;
    }
    static void TransformWithoutOverlap(const Vector2Wide& v, const Symmetric2x2Wide& m, Vector2Wide& result) {  // :64
        result.X = v.X * m.XX + v.Y * m.YX;
        result.Y = v.X * m.YX + v.Y * m.YY;
    }
    static void MultiplyTransposed(const Matrix2x3Wide& a, const Symmetric2x2Wide& b, Matrix2x3Wide& result) {  // :77
        result.X.X = a.X.X * b.XX + a.Y.X * b.YX;
        result.X.Y = a.X.Y * b.XX + a.Y.Y * b.YX;
        result.X.Z = a.X.Z * b.XX + a.Y.Z * b.YX;
        result.Y.X = a.X.X * b.YX + a.Y.X * b.YY;
        result.Y.Y = a.X.Y * b.YX + a.Y.Y * b.YY;
        result.Y.Z = a.X.Z * b.YX + a.Y.Z * b.YY;
    }
    static void CompleteMatrixSandwich(const Matrix2x3Wide& a, const Matrix2x3Wide& b, Symmetric2x2Wide& result) {  // :94
        result.XX = a.X.X * b.X.X + a.X.Y * b.X.Y + a.X.Z * b.X.Z;
        result.YX = a.Y.X * b.X.X + a.Y.Y * b.X.Y + a.Y.Z * b.X.Z;
        result.YY = a.Y.X * b.Y.X + a.Y.Y * b.Y.Y + a.Y.Z * b.Y.Z;
    }
};

struct Symmetric3x3Wide {  // BepuUtilities/Symmetric3x3Wide.cs
    VF XX, YX, YY, ZX, ZY, ZZ;
    static void Invert(const Symmetric3x3Wide& m, Symmetric3x3Wide& inverse) {  // :42
        VF xx = m.YY * m.ZZ - m.ZY * m.ZY;
        VF yx = m.ZY * m.ZX - m.ZZ * m.YX;
        VF zx = m.YX * m.ZY - m.ZX * m.YY;
        VF determinantInverse = kOne / (xx * m.XX + yx * m.YX + zx * m.ZX);
        VF yy = m.ZZ * m.XX - m.ZX * m.ZX;
        VF zy = m.ZX * m.YX - m.XX * m.ZY;
        VF zz = m.XX * m.YY - m.YX * m.YX;
        inverse.XX = xx * determinantInverse;
        inverse.YX = yx * determinantInverse;
        inverse.ZX = zx * determinantInverse;
        inverse.YY = yy * determinantInverse;
        inverse.ZY = zy * determinantInverse;
        inverse.ZZ = zz * determinantInverse;
    }
    static void Add(const Symmetric3x3Wide& a, const Symmetric3x3Wide& b, Symmetric3x3Wide& result) {  // :69
        result.XX = a.XX + b.XX; result.YX = a.YX + b.YX; result.YY = a.YY + b.YY;
        result.ZX = a.ZX + b.ZX; result.ZY = a.ZY + b.ZY; result.ZZ = a.ZZ + b.ZZ;
    }
    static void Subtract(const Symmetric3x3Wide& a, const Symmetric3x3Wide& b, Symmetric3x3Wide& result) {  // :105
        result.XX = a.XX - b.XX; result.YX = a.YX - b.YX; result.YY = a.YY - b.YY;
        result.ZX = a.ZX - b.ZX; result.ZY = a.ZY - b.ZY; result.ZZ = a.ZZ - b.ZZ;
    }
    static void Scale(const Symmetric3x3Wide& m, const VF& scale, Symmetric3x3Wide& result) {  // :135
        result.XX = m.XX * scale; result.YX = m.YX * scale; result.YY = m.YY * scale;
        result.ZX = m.ZX * scale; result.ZY = m.ZY * scale; result.ZZ = m.ZZ * scale;
    }
    static void SkewSandwichWithoutOverlap(const Vector3Wide& v, const Symmetric3x3Wide& m, Symmetric3x3Wide& sandwich) {  // :182
        VF xzy = v.X * m.ZY;
        VF yzx = v.Y * m.ZX;
        VF zyx = v.Z * m.YX;
        VF ixx = yzx - zyx;
        VF ixy = v.Y * m.ZY - v.Z * m.YY;
        VF ixz = v.Y * m.ZZ - v.Z * m.ZY;
        VF iyx = v.Z * m.XX - v.X * m.ZX;
        VF iyy = zyx - xzy;
        VF iyz = v.Z * m.ZX - v.X * m.ZZ;
        VF izx = v.X * m.YX - v.Y * m.XX;
        VF izy = v.X * m.YY - v.Y * m.YX;
        VF izz = xzy - yzx;
        (void)ixx;
        sandwich.XX = v.Y * ixz - v.Z * ixy;
        sandwich.YX = v.Y * iyz - v.Z * iyy;
        sandwich.YY = v.Z * iyx - v.X * iyz;
        sandwich.ZX = v.Y * izz - v.Z * izy;
        sandwich.ZY = v.Z * izx - v.X * izz;
        sandwich.ZZ = v.X * izy - v.Y * izx;
    }
    static void VectorSandwich(const Vector3Wide& v, const Symmetric3x3Wide& m, VF& sandwich) {  // :214
        VF x = v.X * m.XX + v.Y * m.YX + v.Z * m.ZX;
        VF y = v.X * m.YX + v.Y * m.YY + v.Z * m.ZY;
        VF z = v.X * m.ZX + v.Y * m.ZY + v.Z * m.ZZ;
        sandwich = x * v.X + y * v.Y + z * v.Z;
    }
    static void RotationSandwich(const Matrix3x3Wide& r, const Symmetric3x3Wide& m, Symmetric3x3Wide& sandwich) {  // :231
        VF ixx = r.X.X * m.XX + r.Y.X * m.YX + r.Z.X * m.ZX;
        VF ixy = r.X.X * m.YX + r.Y.X * m.YY + r.Z.X * m.ZY;
        VF ixz = r.X.X * m.ZX + r.Y.X * m.ZY + r.Z.X * m.ZZ;
        VF iyx = r.X.Y * m.XX + r.Y.Y * m.YX + r.Z.Y * m.ZX;
        VF iyy = r.X.Y * m.YX + r.Y.Y * m.YY + r.Z.Y * m.ZY;
        VF iyz = r.X.Y * m.ZX + r.Y.Y * m.ZY + r.Z.Y * m.ZZ;
        VF izx = r.X.Z * m.XX + r.Y.Z * m.YX + r.Z.Z * m.ZX;
        VF izy = r.X.Z * m.YX + r.Y.Z * m.YY + r.Z.Z * m.ZY;
        VF izz = r.X.Z * m.ZX + r.Y.Z * m.ZY + r.Z.Z * m.ZZ;
        sandwich.XX = ixx * r.X.X + ixy * r.Y.X + ixz * r.Z.X;
        sandwich.YX = iyx * r.X.X + iyy * r.Y.X + iyz * r.Z.X;
        sandwich.YY = iyx * r.X.Y + iyy * r.Y.Y + iyz * r.Z.Y;
        sandwich.ZX = izx * r.X.X + izy * r.Y.X + izz * r.Z.X;
        sandwich.ZY = izx * r.X.Y + izy * r.Y.Y + izz * r.Z.Y;
        sandwich.ZZ = izx * r.X.Z + izy * r.Y.Z + izz * r.Z.Z;
    }
    static void MultiplyWithoutOverlap(const Matrix2x3Wide& a, const Symmetric3x3Wide& b, Matrix2x3Wide& result) {  // :260
        result.X.X = a.X.X * b.XX + a.X.Y * b.YX + a.X.Z * b.ZX;
        result.X.Y = a.X.X * b.YX + a.X.Y * b.YY + a.X.Z * b.ZY;
        result.X.Z = a.X.X * b.ZX + a.X.Y * b.ZY + a.X.Z * b.ZZ;
        result.Y.X = a.Y.X * b.XX + a.Y.Y * b.YX + a.Y.Z * b.ZX;
        result.Y.Y = a.Y.X * b.YX + a.Y.Y * b.YY + a.Y.Z * b.ZY;
        result.Y.Z = a.Y.X * b.ZX + a.Y.Y * b.ZY + a.Y.Z * b.ZZ;
    }
    static void MultiplyWithoutOverlap(const Matrix3x3Wide& a, const Symmetric3x3Wide& b, Matrix3x3Wide& result) {  // :296
        result.X.X = a.X.X * b.XX + a.X.Y * b.YX + a.X.Z * b.ZX;
        result.X.Y = a.X.X * b.YX + a.X.Y * b.YY + a.X.Z * b.ZY;
        result.X.Z = a.X.X * b.ZX + a.X.Y * b.ZY + a.X.Z * b.ZZ;
        result.Y.X = a.Y.X * b.XX + a.Y.Y * b.YX + a.Y.Z * b.ZX;
        result.Y.Y = a.Y.X * b.YX + a.Y.Y * b.YY + a.Y.Z * b.ZY;
        result.Y.Z = a.Y.X * b.ZX + a.Y.Y * b.ZY + a.Y.Z * b.ZZ;
        result.Z.X = a.Z.X * b.XX + a.Z.Y * b.YX + a.Z.Z * b.ZX;
        result.Z.Y = a.Z.X * b.YX + a.Z.Y * b.YY + a.Z.Z * b.ZY;
        result.Z.Z = a.Z.X * b.ZX + a.Z.Y * b.ZY + a.Z.Z * b.ZZ;
    }
    static void MatrixSandwich(const Matrix2x3Wide& m, const Symmetric3x3Wide& t, Symmetric2x2Wide& result) {  // :429
        VF ixx = m.X.X * t.XX + m.X.Y * t.YX + m.X.Z * t.ZX;
        VF ixy = m.X.X * t.YX + m.X.Y * t.YY + m.X.Z * t.ZY;
        VF ixz = m.X.X * t.ZX + m.X.Y * t.ZY + m.X.Z * t.ZZ;
        VF iyx = m.Y.X * t.XX + m.Y.Y * t.YX + m.Y.Z * t.ZX;
        VF iyy = m.Y.X * t.YX + m.Y.Y * t.YY + m.Y.Z * t.ZY;
        VF iyz = m.Y.X * t.ZX + m.Y.Y * t.ZY + m.Y.Z * t.ZZ;
        result.XX = ixx * m.X.X + ixy * m.X.Y + ixz * m.X.Z;
        result.YX = iyx * m.X.X + iyy * m.X.Y + iyz * m.X.Z;
        result.YY = iyx * m.Y.X + iyy * m.Y.Y + iyz * m.Y.Z;
    }
    static void CompleteMatrixSandwich(const Matrix3x3Wide& a, const Matrix3x3Wide& b, Symmetric3x3Wide& result) {  // :449
        result.XX = a.X.X * b.X.X + a.X.Y * b.Y.X + a.X.Z * b.Z.X;
        result.YX = a.Y.X * b.X.X + a.Y.Y * b.Y.X + a.Y.Z * b.Z.X;
        result.YY = a.Y.X * b.X.Y + a.Y.Y * b.Y.Y + a.Y.Z * b.Z.Y;
        result.ZX = a.Z.X * b.X.X + a.Z.Y * b.Y.X + a.Z.Z * b.Z.X;
        result.ZY = a.Z.X * b.X.Y + a.Z.Y * b.Y.Y + a.Z.Z * b.Z.Y;
        result.ZZ = a.Z.X * b.X.Z + a.Z.Y * b.Y.Z + a.Z.Z * b.Z.Z;
    }
    static void MultiplyByTransposed(const Symmetric3x3Wide& a, const Matrix2x3Wide& b, Matrix2x3Wide& result) {  // :410
        result.X.X = a.XX * b.X.X + a.YX * b.X.Y + a.ZX * b.X.Z;
        result.Y.X = a.XX * b.Y.X + a.YX * b.Y.Y + a.ZX * b.Y.Z;
        result.X.Y = a.YX * b.X.X + a.YY * b.X.Y + a.ZY * b.X.Z;
        result.Y.Y = a.YX * b.Y.X + a.YY * b.Y.Y + a.ZY * b.Y.Z;
        result.X.Z = a.ZX * b.X.X + a.ZY * b.X.Y + a.ZZ * b.X.Z;
        result.Y.Z = a.ZX * b.Y.X + a.ZY * b.Y.Y + a.ZZ * b.Y.Z;
    }
    static void CompleteMatrixSandwich(const Matrix2x3Wide& a, const Matrix2x3Wide& b, Symmetric3x3Wide& result) {  // :470
        result.XX = a.X.X * b.X.X + a.Y.X * b.Y.X;
        result.YX = a.X.Y * b.X.X + a.Y.Y * b.Y.X;
        result.YY = a.X.Y * b.X.Y + a.Y.Y * b.Y.Y;
        result.ZX = a.X.Z * b.X.X + a.Y.Z * b.Y.X;
        result.ZY = a.X.Z * b.X.Y + a.Y.Z * b.Y.Y;
        result.ZZ = a.X.Z * b.X.Z + a.Y.Z * b.Y.Z;
    }
    static void CompleteMatrixSandwichByTranspose(const Matrix3x3Wide& a, const Matrix3x3Wide& b, Symmetric3x3Wide& result) {  // :489
        result.XX = a.X.X * b.X.X + a.X.Y * b.X.Y + a.X.Z * b.X.Z;
        result.YX = a.Y.X * b.X.X + a.Y.Y * b.X.Y + a.Y.Z * b.X.Z;
        result.YY = a.Y.X * b.Y.X + a.Y.Y * b.Y.Y + a.Y.Z * b.Y.Z;
        result.ZX = a.Z.X * b.X.X + a.Z.Y * b.X.Y + a.Z.Z * b.X.Z;
        result.ZY = a.Z.X * b.Y.X + a.Z.Y * b.Y.Y + a.Z.Z * b.Y.Z;
        result.ZZ = a.Z.X * b.Z.X + a.Z.Y * b.Z.Y + a.Z.Z * b.Z.Z;
    }
    static void TransformWithoutOverlap(const Vector3Wide& v, const Symmetric3x3Wide& m, Vector3Wide& result) {  // :521
        result.X = v.X * m.XX + v.Y * m.YX + v.Z * m.ZX;
        result.Y = v.X * m.YX + v.Y * m.YY + v.Z * m.ZY;
        result.Z = v.X * m.ZX + v.Y * m.ZY + v.Z * m.ZZ;
    }
};
static inline Symmetric3x3Wide operator+(const Symmetric3x3Wide& a, const Symmetric3x3Wide& b) {  // Symmetric3x3Wide.cs:85
    return Symmetric3x3Wide{a.XX + b.XX, a.YX + b.YX, a.YY + b.YY, a.ZX + b.ZX, a.ZY + b.ZY, a.ZZ + b.ZZ};
}
static inline Symmetric3x3Wide operator*(const Symmetric3x3Wide& m, const VF& scale) {  // :146
    return Symmetric3x3Wide{m.XX * scale, m.YX * scale, m.YY * scale, m.ZX * scale, m.ZY * scale, m.ZZ * scale};
}
static inline Vector3Wide operator*(const Vector3Wide& v, const Symmetric3x3Wide& m) {  // :529
    Vector3Wide result;
    result.X = v.X * m.XX + v.Y * m.YX + v.Z * m.ZX;
    result.Y = v.X * m.YX + v.Y * m.YY + v.Z * m.ZY;
    result.Z = v.X * m.ZX + v.Y * m.ZY + v.Z * m.ZZ;
    return result;
}
static inline Matrix3x3Wide operator*(const Matrix3x3Wide& a, const Symmetric3x3Wide& b) {  // :319
    Matrix3x3Wide result;
    Symmetric3x3Wide::MultiplyWithoutOverlap(a, b, result);
    return result;
}
static inline Matrix3x3Wide operator+(const Symmetric3x3Wide& a, const Matrix3x3Wide& b) {  // :539
    Matrix3x3Wide result;
    result.X.X = a.XX + b.X.X; result.X.Y = a.YX + b.X.Y; result.X.Z = a.ZX + b.X.Z;
    result.Y.X = a.YX + b.Y.X; result.Y.Y = a.YY + b.Y.Y; result.Y.Z = a.ZY + b.Y.Z;
    result.Z.X = a.ZX + b.Z.X; result.Z.Y = a.ZY + b.Z.Y; result.Z.Z = a.ZZ + b.Z.Z;
    return result;
}

struct Vector4Wide {  // BepuUtilities/Vector4Wide.cs
    VF X, Y, Z, W;
    static void Subtract(const Vector4Wide& a, const Vector4Wide& b, Vector4Wide& result) {  // :104
        result.X = a.X - b.X; result.Y = a.Y - b.Y; result.Z = a.Z - b.Z; result.W = a.W - b.W;
    }
    static void Scale(const Vector4Wide& vector, const VF& scalar, Vector4Wide& result) {
        result.X = vector.X * scalar; result.Y = vector.Y * scalar; result.Z = vector.Z * scalar; result.W = vector.W * scalar;
    }
};
static inline Vector4Wide operator+(const Vector4Wide& a, const Vector4Wide& b) { return Vector4Wide{a.X + b.X, a.Y + b.Y, a.Z + b.Z, a.W + b.W}; }  // :61

struct Symmetric4x4Wide {  // BepuUtilities/Symmetric4x4Wide.cs
    VF XX, YX, YY, ZX, ZY, ZZ, WX, WY, WZ, WW;  // the first six alias a Symmetric3x3Wide (:25), WX..WZ a Vector3Wide (:35)
    static void InvertWithoutOverlap(const Symmetric4x4Wide& m, Symmetric4x4Wide& result) {  // :62
        VF s0 = m.XX * m.YY - m.YX * m.YX;
        VF s1 = m.XX * m.ZY - m.YX * m.ZX;
        VF s2 = m.XX * m.WY - m.YX * m.WX;
        VF s3 = m.YX * m.ZY - m.YY * m.ZX;
        VF s4 = m.YX * m.WY - m.YY * m.WX;
        VF s5 = m.ZX * m.WY - m.ZY * m.WX;
        VF c5 = m.ZZ * m.WW - m.WZ * m.WZ;
        VF c4 = m.ZY * m.WW - m.WY * m.WZ;
        VF c3 = m.ZY * m.WZ - m.WY * m.ZZ;
        VF c2 = m.ZX * m.WW - m.WX * m.WZ;
        VF c1 = m.ZX * m.WZ - m.WX * m.ZZ;
        VF inverseDeterminant = kOne / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * s5);
        result.XX = (m.YY * c5 - m.ZY * c4 + m.WY * c3) * inverseDeterminant;
        result.YX = (neg(m.YX) * c5 + m.ZY * c2 - m.WY * c1) * inverseDeterminant;
        result.YY = (m.XX * c5 - m.ZX * c2 + m.WX * c1) * inverseDeterminant;
        result.ZX = (m.YX * c4 - m.YY * c2 + m.WY * s5) * inverseDeterminant;
        result.ZY = (neg(m.XX) * c4 + m.YX * c2 - m.WX * s5) * inverseDeterminant;
        result.ZZ = (m.WX * s4 - m.WY * s2 + m.WW * s0) * inverseDeterminant;
        result.WX = (neg(m.YX) * c3 + m.YY * c1 - m.ZY * s5) * inverseDeterminant;
        result.WY = (m.XX * c3 - m.YX * c1 + m.ZX * s5) * inverseDeterminant;
        result.WZ = (neg(m.WX) * s3 + m.WY * s1 - m.WZ * s0) * inverseDeterminant;
        result.WW = (m.ZX * s3 - m.ZY * s1 + m.ZZ * s0) * inverseDeterminant;
    }
    static void TransformWithoutOverlap(const Vector4Wide& v, const Symmetric4x4Wide& m, Vector4Wide& result) {  // :101
        result.X = v.X * m.XX + v.Y * m.YX + v.Z * m.ZX + v.W * m.WX;
        result.Y = v.X * m.YX + v.Y * m.YY + v.Z * m.ZY + v.W * m.WY;
        result.Z = v.X * m.ZX + v.Y * m.ZY + v.Z * m.ZZ + v.W * m.WZ;
        result.W = v.X * m.WX + v.Y * m.WY + v.Z * m.WZ + v.W * m.WW;
    }
};

struct Symmetric5x5Wide {  // BepuUtilities/Symmetric5x5Wide.cs
    Symmetric3x3Wide A;
    Matrix2x3Wide B;
    Symmetric2x2Wide D;
    static void Invert(const Symmetric3x3Wide& a, const Matrix2x3Wide& b, const Symmetric2x2Wide& d, Symmetric5x5Wide& result) {  // :36
        Symmetric2x2Wide invD;
        Symmetric2x2Wide::InvertWithoutOverlap(d, invD);
        Matrix2x3Wide bTInvD;
        Symmetric2x2Wide::MultiplyTransposed(b, invD, bTInvD);
        Symmetric3x3Wide bTInvDB;
        Symmetric3x3Wide::CompleteMatrixSandwich(bTInvD, b, bTInvDB);
        Symmetric3x3Wide resultAInverse;
        Symmetric3x3Wide::Subtract(a, bTInvDB, resultAInverse);
        Symmetric3x3Wide::Invert(resultAInverse, result.A);
        Matrix2x3Wide negatedResultBT;
        Symmetric3x3Wide::MultiplyByTransposed(result.A, bTInvD, negatedResultBT);
        Matrix2x3Wide::Negate(negatedResultBT, result.B);
        Symmetric2x2Wide::CompleteMatrixSandwich(bTInvD, negatedResultBT, result.D);
        Symmetric2x2Wide::Add(result.D, invD, result.D);
    }
    static void InvertWithoutOverlap(const Symmetric5x5Wide& m, Symmetric5x5Wide& result) { Invert(m.A, m.B, m.D, result); }  // :53
    static void TransformWithoutOverlap(const Vector3Wide& v0, const Vector2Wide& v1, const Symmetric5x5Wide& m, Vector3Wide& result0, Vector2Wide& result1) {  // :67
        result0.X = v0.X * m.A.XX + v0.Y * m.A.YX + v0.Z * m.A.ZX + v1.X * m.B.X.X + v1.Y * m.B.Y.X;
        result0.Y = v0.X * m.A.YX + v0.Y * m.A.YY + v0.Z * m.A.ZY + v1.X * m.B.X.Y + v1.Y * m.B.Y.Y;
        result0.Z = v0.X * m.A.ZX + v0.Y * m.A.ZY + v0.Z * m.A.ZZ + v1.X * m.B.X.Z + v1.Y * m.B.Y.Z;
        result1.X = v0.X * m.B.X.X + v0.Y * m.B.X.Y + v0.Z * m.B.X.Z + v1.X * m.D.XX + v1.Y * m.D.YX;
        result1.Y = v0.X * m.B.Y.X + v0.Y * m.B.Y.Y + v0.Z * m.B.Y.Z + v1.X * m.D.YX + v1.Y * m.D.YY;
    }
};

// BepuUtilities/MathHelper.cs:274-376. Constants are float literals, written as in the C#.
namespace MathHelper {
static inline VF Cos(VF x) {  // :274
    VF periodCount = x * vf((float)(0.5 / 3.14159265358979323846));
    VF periodFraction = periodCount - Floor(periodCount);
    VF twoPi = vf(TwoPi);
    VF periodX = periodFraction * twoPi;
    VF y;
    VF piOver2 = vf(PiOver2);
    VF pi = vf(Pi);
    VF pi3Over2 = vf(3 * PiOver2);
    y = ConditionalSelect(GreaterThan(periodX, piOver2), pi - periodX, periodX);
    y = ConditionalSelect(GreaterThan(periodX, pi), periodX - pi, y);
    y = ConditionalSelect(GreaterThan(periodX, pi3Over2), vf(TwoPi) - periodX, y);
    VF numerator = ((((vf(-0.003436308368583229f) * y + vf(0.021317031205957775f)) * y + vf(0.06955843390178032f)) * y - vf(0.4578088075324152f)) * y - vf(0.15082367674208508f)) * y + kOne;
    VF denominator = ((((vf(-0.00007650398834677185f) * y + vf(0.0007451378206294365f)) * y - vf(0.00585321045829395f)) * y + vf(0.04219116713777847f)) * y - vf(0.15082367538305258f)) * y + kOne;
    VF result = numerator / denominator;
    return ConditionalSelect(BitwiseAnd(GreaterThan(periodX, piOver2), LessThan(periodX, pi3Over2)), neg(result), result);
}
static inline VF Sin(VF x) {  // :313
    VF periodCount = x * vf((float)(0.5 / 3.14159265358979323846));
    VF periodFraction = periodCount - Floor(periodCount);
    VF twoPi = vf(TwoPi);
    VF periodX = periodFraction * twoPi;
    VF y;
    VF pi = vf(Pi);
    VF piOver2 = vf(PiOver2);
    y = ConditionalSelect(GreaterThan(periodX, piOver2), pi - periodX, periodX);
    VI inSecondHalf = GreaterThan(periodX, pi);
    y = ConditionalSelect(inSecondHalf, periodX - pi, y);
    y = ConditionalSelect(GreaterThan(periodX, vf(3 * PiOver2)), twoPi - periodX, y);
    VF numerator = ((((vf(0.0040507708755727605f) * y - vf(0.006685815219853882f)) * y - vf(0.13993701695343166f)) * y + vf(0.06174562337697123f)) * y + vf(1.00000000151466040f)) * y;
    VF denominator = ((((vf(0.00009018370615921334f) * y + vf(0.0001700784176413186f)) * y + vf(0.003606014457152456f)) * y + vf(0.02672943625500751f)) * y + vf(0.061745651499203795f)) * y + kOne;
    VF result = numerator / denominator;
    return ConditionalSelect(inSecondHalf, neg(result), result);
}
static inline VF Acos(VF x) {  // :353
    VI negativeInput = LessThan(x, kZero);
    x = Min(kOne, Abs(x));
    VF numerator = SquareRoot(kOne - x) * (vf(62.95741097600742f) + x * (vf(69.6550664543659f) + x * (vf(17.54512349463405f) + x * vf(0.6022076120669532f))));
    VF denominator = vf(40.07993264439811f) + x * (vf(49.81949855726789f) + x * (vf(15.703851745284796f) + x));
    VF result = numerator / denominator;
    return ConditionalSelect(negativeInput, vf(Pi) - result, result);
}
static inline void GetSignedAngleDifference(const VF& a, const VF& b, VF& difference) {  // :371
    VF half = vf(0.5f);
    VF x = (b - a) * vf(1.0f / TwoPi) + half;
    difference = (x - Floor(x) - half) * vf(TwoPi);
}
}  // namespace MathHelper

// BepuPhysics/Helpers.cs:21-47
namespace Helpers {
static inline void BuildOrthonormalBasis(const Vector3Wide& normal, Vector3Wide& t1, Vector3Wide& t2) {  // :21
    VF sign = ConditionalSelect(LessThan(normal.Z, kZero), neg(kOne), kOne);
    VF scale = neg(kOne) / (sign + normal.Z);
    t1.X = normal.X * normal.Y * scale;
    t1.Y = sign + normal.Y * normal.Y * scale;
    t1.Z = neg(normal.Y);
    t2.X = kOne + sign * normal.X * normal.X * scale;
    t2.Y = sign * t1.X;
    t2.Z = neg(sign) * normal.X;
}
static inline void FindPerpendicular(const Vector3Wide& normal, Vector3Wide& perpendicular) {  // :38
    VF sign = ConditionalSelect(LessThan(normal.Z, kZero), neg(kOne), kOne);
    VF scale = neg(kOne) / (sign + normal.Z);
    perpendicular.X = normal.X * normal.Y * scale;
    perpendicular.Y = sign + normal.Y * normal.Y * scale;
    perpendicular.Z = neg(normal.Y);
}
}  // namespace Helpers

}  // namespace wide
