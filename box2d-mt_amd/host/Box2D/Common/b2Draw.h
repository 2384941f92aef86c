// Drop-in header: debug-draw interface (reference: Box2D/Common/b2Draw.h:25-88). Interface only: the Testbed's DebugDraw
// derives from it; b2World::DrawDebugData is not part of the Step() hot path and is not provided.
#ifndef B2_DRAW_H
#define B2_DRAW_H

#include "Box2D/Common/b2Math.h"

struct b2Color
{
	b2Color() {}
	b2Color(float32 rIn, float32 gIn, float32 bIn, float32 aIn = 1.0f) : r(rIn), g(gIn), b(bIn), a(aIn) {}
	void Set(float32 rIn, float32 gIn, float32 bIn, float32 aIn = 1.0f) { r = rIn; g = gIn; b = bIn; a = aIn; }
	float32 r, g, b, a;
};

class b2Draw
{
public:
	b2Draw() : m_drawFlags(0) {}
	virtual ~b2Draw() {}
	enum
	{
		e_shapeBit = 0x0001,
		e_jointBit = 0x0002,
		e_aabbBit = 0x0004,
		e_pairBit = 0x0008,
		e_centerOfMassBit = 0x0010
	};
	void SetFlags(uint32 flags) { m_drawFlags = flags; }
	uint32 GetFlags() const { return m_drawFlags; }
	void AppendFlags(uint32 flags) { m_drawFlags |= flags; }
	void ClearFlags(uint32 flags) { m_drawFlags &= ~flags; }
	virtual void DrawPolygon(const b2Vec2* vertices, int32 vertexCount, const b2Color& color) = 0;
	virtual void DrawSolidPolygon(const b2Vec2* vertices, int32 vertexCount, const b2Color& color) = 0;
	virtual void DrawCircle(const b2Vec2& center, float32 radius, const b2Color& color) = 0;
	virtual void DrawSolidCircle(const b2Vec2& center, float32 radius, const b2Vec2& axis, const b2Color& color) = 0;
	virtual void DrawSegment(const b2Vec2& p1, const b2Vec2& p2, const b2Color& color) = 0;
	virtual void DrawTransform(const b2Transform& xf) = 0;
	virtual void DrawPoint(const b2Vec2& p, float32 size, const b2Color& color) = 0;

protected:
	uint32 m_drawFlags;
};

#endif
