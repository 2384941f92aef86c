// Host-side implementation of the drop-in Box2D-MT API: common utilities, math out-of-lines.
#include "Box2D/Common/b2Settings.h"
#include "Box2D/Common/b2Math.h"

#include "../../csrc/b2d_mat33.h"

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

// b2Mat33's out-of-line members (b2Math.h:237-291; b2Math.cpp:25-95): the arithmetic is stated once, in the header the device
// joint solvers use (csrc/b2d_mat33.h); these are the host API's doors to it.
static M33 columns(const b2Mat33& A)
{
	M33 K;
	K.ex = v3(A.ex.x, A.ex.y, A.ex.z);
	K.ey = v3(A.ey.x, A.ey.y, A.ey.z);
	K.ez = v3(A.ez.x, A.ez.y, A.ez.z);
	return K;
}

static void store(b2Mat33* out, const M33& K)
{
	out->ex.Set(K.ex.x, K.ex.y, K.ex.z);
	out->ey.Set(K.ey.x, K.ey.y, K.ey.z);
	out->ez.Set(K.ez.x, K.ez.y, K.ez.z);
}

b2Vec3 b2Mat33::Solve33(const b2Vec3& b) const
{
	const V3 x = b2dM33Solve33(columns(*this), v3(b.x, b.y, b.z));
	return b2Vec3(x.x, x.y, x.z);
}

b2Vec2 b2Mat33::Solve22(const b2Vec2& b) const
{
	const V2 x = b2dM33Solve22(columns(*this), v2(b.x, b.y));
	return b2Vec2(x.x, x.y);
}

void b2Mat33::GetInverse22(b2Mat33* M) const { store(M, b2dM33Inverse22(columns(*this))); }
void b2Mat33::GetSymInverse33(b2Mat33* M) const { store(M, b2dM33SymInverse33(columns(*this))); }

// The reference's debugging switch for the 2-point block solver (b2ContactSolver.cpp:30; Testbed/Tests/VerticalStack.h flips
// it from the keyboard). The symbol exists so that such scenes link; the device solver always runs the block solver
// (the reference's default) and does not read it.
bool g_blockSolve = true;
