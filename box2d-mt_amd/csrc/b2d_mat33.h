// b2d_mat33.h - 3-vectors and the 3x3 column matrix of the joint solvers (b2Vec3 / b2Mat33, b2Math.h:115-291,
// b2Math.cpp:25-95), in the reference's operand order. Shared by the device joint code (b2d_joint.h) and by the
// drop-in host API (host/src/b2_common.cpp: b2Mat33's out-of-line members forward here), so that there is ONE statement of
// this arithmetic in the tree.
#ifndef B2D_MAT33_H
#define B2D_MAT33_H

#include "b2d_math.h"

struct V3
{
	float x, y, z;
};

B2D_HD V3 b2dCross3(V3 a, V3 b)
{
	V3 r;
	r.x = a.y * b.z - a.z * b.y;
	r.y = a.z * b.x - a.x * b.z;
	r.z = a.x * b.y - a.y * b.x;
	return r;
}
B2D_HD float b2dDot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

B2D_HD V3 v3(float x, float y, float z)
{
	V3 r;
	r.x = x; r.y = y; r.z = z;
	return r;
}

// b2Mat33 (b2Math.h:237-291) as three columns
struct M33
{
	V3 ex, ey, ez;
};

B2D_HD V3 b2dM33Solve33(const M33& K, V3 b)
{
	float det = b2dDot3(K.ex, b2dCross3(K.ey, K.ez));
	if (det != 0.0f) det = 1.0f / det;
	V3 x;
	x.x = det * b2dDot3(b, b2dCross3(K.ey, K.ez));
	x.y = det * b2dDot3(K.ex, b2dCross3(b, K.ez));
	x.z = det * b2dDot3(K.ex, b2dCross3(K.ey, b));
	return x;
}

B2D_HD V2 b2dM33Solve22(const M33& K, V2 b)
{
	float a11 = K.ex.x, a12 = K.ey.x, a21 = K.ex.y, a22 = K.ey.y;
	float det = a11 * a22 - a12 * a21;
	if (det != 0.0f) det = 1.0f / det;
	V2 x;
	x.x = det * (a22 * b.x - a12 * b.y);
	x.y = det * (a11 * b.y - a21 * b.x);
	return x;
}

// b2Mat33::GetInverse22 (b2Math.cpp:56-68)
B2D_HD M33 b2dM33Inverse22(const M33& K)
{
	float a = K.ex.x, b = K.ey.x, c = K.ex.y, d = K.ey.y;
	float det = a * d - b * c;
	if (det != 0.0f) det = 1.0f / det;
	M33 M;
	M.ex = v3(det * d, -det * c, 0.0f);
	M.ey = v3(-det * b, det * a, 0.0f);
	M.ez = v3(0.0f, 0.0f, 0.0f);
	return M;
}

// b2Mat33::GetSymInverse33 (b2Math.cpp:71-94)
B2D_HD M33 b2dM33SymInverse33(const M33& K)
{
	float det = b2dDot3(K.ex, b2dCross3(K.ey, K.ez));
	if (det != 0.0f) det = 1.0f / det;
	float a11 = K.ex.x, a12 = K.ey.x, a13 = K.ez.x;
	float a22 = K.ey.y, a23 = K.ez.y;
	float a33 = K.ez.z;
	M33 M;
	M.ex.x = det * (a22 * a33 - a23 * a23);
	M.ex.y = det * (a13 * a23 - a12 * a33);
	M.ex.z = det * (a12 * a23 - a13 * a22);
	M.ey.x = M.ex.y;
	M.ey.y = det * (a11 * a33 - a13 * a13);
	M.ey.z = det * (a13 * a12 - a11 * a23);
	M.ez.x = M.ex.z;
	M.ez.y = M.ey.z;
	M.ez.z = det * (a11 * a22 - a12 * a12);
	return M;
}

// b2Mul(b2Mat33, b2Vec3) (b2Math.h:515-518): v.x * ex + v.y * ey + v.z * ez
B2D_HD V3 b2dM33Mul(const M33& A, V3 v)
{
	return v3(v.x * A.ex.x + v.y * A.ey.x + v.z * A.ez.x, v.x * A.ex.y + v.y * A.ey.y + v.z * A.ez.y,
		v.x * A.ex.z + v.y * A.ey.z + v.z * A.ez.z);
}

// b2Mul22(b2Mat33, b2Vec2) (b2Math.h:521-524)
B2D_HD V2 b2dM33Mul22(const M33& A, V2 v)
{
	return v2(A.ex.x * v.x + A.ey.x * v.y, A.ex.y * v.x + A.ey.y * v.y);
}

#endif
