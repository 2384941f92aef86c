// Host-side implementation of the drop-in Box2D-MT API: common utilities, math out-of-lines.
#include "Box2D/Common/b2Settings.h"
#include "Box2D/Common/b2Math.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

b2Version b2_version = { 2, 3, 2 };
b2Version b2_mtVersion = { 0, 1, 0 };

const b2Vec2 b2Vec2_zero(0.0f, 0.0f);

void* b2Alloc(int32 size)
{
	return malloc((size_t)size);
}

void b2Free(void* mem)
{
	free(mem);
}

void b2Log(const char* string, ...)
{
	va_list args;
	va_start(args, string);
	vprintf(string, args);
	va_end(args);
}

// Cramer's rule, reference semantics (b2Math.cpp:25-53)
b2Vec3 b2Mat33::Solve33(const b2Vec3& b) const
{
	float32 det = b2Dot(ex, b2Cross(ey, ez));
	if (det != 0.0f) det = 1.0f / det;
	b2Vec3 x;
	x.x = det * b2Dot(b, b2Cross(ey, ez));
	x.y = det * b2Dot(ex, b2Cross(b, ez));
	x.z = det * b2Dot(ex, b2Cross(ey, b));
	return x;
}

b2Vec2 b2Mat33::Solve22(const b2Vec2& b) const
{
	float32 a11 = ex.x, a12 = ey.x, a21 = ex.y, a22 = ey.y;
	float32 det = a11 * a22 - a12 * a21;
	if (det != 0.0f) det = 1.0f / det;
	b2Vec2 x;
	x.x = det * (a22 * b.x - a12 * b.y);
	x.y = det * (a11 * b.y - a21 * b.x);
	return x;
}

void b2Mat33::GetInverse22(b2Mat33* M) const
{
	float32 a = ex.x, b = ey.x, c = ex.y, d = ey.y;
	float32 det = a * d - b * c;
	if (det != 0.0f) det = 1.0f / det;
	M->ex.x = det * d;
	M->ey.x = -det * b;
	M->ex.z = 0.0f;
	M->ex.y = -det * c;
	M->ey.y = det * a;
	M->ey.z = 0.0f;
	M->ez.x = 0.0f;
	M->ez.y = 0.0f;
	M->ez.z = 0.0f;
}

void b2Mat33::GetSymInverse33(b2Mat33* M) const
{
	float32 det = b2Dot(ex, b2Cross(ey, ez));
	if (det != 0.0f) det = 1.0f / det;
	float32 a11 = ex.x, a12 = ey.x, a13 = ez.x;
	float32 a22 = ey.y, a23 = ez.y;
	float32 a33 = ez.z;
	M->ex.x = det * (a22 * a33 - a23 * a23);
	M->ex.y = det * (a13 * a23 - a12 * a33);
	M->ex.z = det * (a12 * a23 - a13 * a22);
	M->ey.x = M->ex.y;
	M->ey.y = det * (a11 * a33 - a13 * a13);
	M->ey.z = det * (a13 * a12 - a11 * a23);
	M->ez.x = M->ex.z;
	M->ez.y = M->ey.z;
	M->ez.z = det * (a11 * a22 - a12 * a12);
}
