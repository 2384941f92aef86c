// Drop-in header: a dynamic AABB tree with the public interface of the reference's b2DynamicTree
// (Box2D/Collision/b2DynamicTree.h:52-130: CreateProxy / DestroyProxy / MoveProxy with the fat-AABB rule, Query, RayCast,
// height / balance / area-ratio statistics). The device broad-phase does not use a tree (it rebuilds a hashed grid per step,
// DESIGN.md section 3); this host-side tree serves user code that wants one (Testbed/Tests/DynamicTreeTest.h) and is the
// shadow structure b2World::QueryAABB / RayCast / GetTreeHeight / GetTreeBalance / GetTreeQuality run on: its leaves are the
// device's fat AABBs, refitted on demand after a step (host/src/b2_world.cpp).
//
// Own implementation: a node pool with a free list, insertion by a branch-and-bound descent on the perimeter heuristic,
// height-balancing rotations on the way back up. Stored boxes are fat by b2_aabbExtension (public API) or taken as they
// are (the *Fat calls of the shadow tree).
#ifndef B2_DYNAMIC_TREE_H
#define B2_DYNAMIC_TREE_H

#include "Box2D/Collision/b2Collision.h"

#define b2_nullNode (-1)

struct b2TreeNode
{
	bool IsLeaf() const { return child1 == b2_nullNode; }
	b2AABB aabb;
	void* userData;
	union
	{
		int32 parent;
		int32 next; // free list
	};
	int32 child1, child2;
	int32 height; // leaf 0, free node -1
};

class b2DynamicTree
{
public:
	b2DynamicTree();
	~b2DynamicTree();
	b2DynamicTree(const b2DynamicTree&) = delete;
	b2DynamicTree& operator=(const b2DynamicTree&) = delete;

	int32 CreateProxy(const b2AABB& aabb, void* userData);
	void DestroyProxy(int32 proxyId);
	bool MoveProxy(int32 proxyId, const b2AABB& aabb1, const b2Vec2& displacement);
	void* GetUserData(int32 proxyId) const { return m_nodes[proxyId].userData; }
	const b2AABB& GetFatAABB(int32 proxyId) const { return m_nodes[proxyId].aabb; }

	template <typename T> void Query(T* callback, const b2AABB& aabb) const;
	template <typename T> void RayCast(T* callback, const b2RayCastInput& input) const;

	void Validate() const;
	int32 GetHeight() const { return m_root == b2_nullNode ? 0 : m_nodes[m_root].height; }
	int32 GetMaxBalance() const;
	float32 GetAreaRatio() const;
	void RebuildBottomUp();
	void ShiftOrigin(const b2Vec2& newOrigin);

	// ---- the shadow tree's interface: leaves whose box is stored exactly as given ----------------------------------------
	int32 InsertFat(const b2AABB& fat, void* userData);
	void RemoveFat(int32 leaf) { DestroyProxy(leaf); }
	void ReplaceFat(int32 leaf, const b2AABB& fat);
	int32 GetLeafCount() const { return m_leafCount; }
	void Clear();

private:
	int32 TakeNode();
	void GiveBack(int32 node);
	void Attach(int32 leaf);
	void Detach(int32 leaf);
	int32 Rebalance(int32 node);
	void Refit(int32 node);
	void CheckSubtree(int32 node, int32 expectedParent) const;

	int32 m_root;
	b2TreeNode* m_nodes;
	int32 m_capacity, m_used, m_freeHead, m_leafCount;
};

// Every leaf whose box overlaps `aabb`, depth first; the callback may stop the walk by returning false.
template <typename T> inline void b2DynamicTree::Query(T* callback, const b2AABB& aabb) const
{
	if (m_root == b2_nullNode) return;
	int32 small[64];
	int32* stack = small;
	int32 cap = 64, top = 0;
	stack[top++] = m_root;
	while (top > 0)
	{
		const int32 id = stack[--top];
		const b2TreeNode& node = m_nodes[id];
		if (!b2TestOverlap(node.aabb, aabb)) continue;
		if (node.IsLeaf())
		{
			if (!callback->QueryCallback(id)) break;
			continue;
		}
		if (top + 2 > cap)
		{
			int32* bigger = (int32*)b2Alloc(2 * cap * (int32)sizeof(int32));
			for (int32 k = 0; k < top; ++k) bigger[k] = stack[k];
			if (stack != small) b2Free(stack);
			stack = bigger;
			cap *= 2;
		}
		stack[top++] = node.child1;
		stack[top++] = node.child2;
	}
	if (stack != small) b2Free(stack);
}

// Leaves met by the segment p1 -> p1 + maxFraction (p2 - p1). The callback returns the new clip fraction: 0 ends the cast,
// a negative value leaves the clip as it is, a positive one shortens (or keeps) the segment.
template <typename T> inline void b2DynamicTree::RayCast(T* callback, const b2RayCastInput& input) const
{
	if (m_root == b2_nullNode) return;
	const b2Vec2 p1 = input.p1, p2 = input.p2;
	b2Vec2 dir = p2 - p1;
	if (dir.LengthSquared() <= 0.0f) return;
	dir.Normalize();
	// a box is missed when its centre is further from the supporting line than its extent projected on the line's normal
	const b2Vec2 side = b2Cross(1.0f, dir);
	const b2Vec2 sideAbs = b2Abs(side);
	float32 clip = input.maxFraction;
	b2AABB reach;
	{
		const b2Vec2 end = p1 + clip * (p2 - p1);
		reach.lowerBound = b2Min(p1, end);
		reach.upperBound = b2Max(p1, end);
	}
	int32 small[64];
	int32* stack = small;
	int32 cap = 64, top = 0;
	stack[top++] = m_root;
	while (top > 0)
	{
		const int32 id = stack[--top];
		const b2TreeNode& node = m_nodes[id];
		if (!b2TestOverlap(node.aabb, reach)) continue;
		const b2Vec2 centre = node.aabb.GetCenter(), half = node.aabb.GetExtents();
		if (b2Abs(b2Dot(side, p1 - centre)) - b2Dot(sideAbs, half) > 0.0f) continue;
		if (node.IsLeaf())
		{
			b2RayCastInput sub;
			sub.p1 = p1;
			sub.p2 = p2;
			sub.maxFraction = clip;
			const float32 answer = callback->RayCastCallback(sub, id);
			if (answer == 0.0f) break;
			if (answer > 0.0f)
			{
				clip = answer;
				const b2Vec2 end = p1 + clip * (p2 - p1);
				reach.lowerBound = b2Min(p1, end);
				reach.upperBound = b2Max(p1, end);
			}
			continue;
		}
		if (top + 2 > cap)
		{
			int32* bigger = (int32*)b2Alloc(2 * cap * (int32)sizeof(int32));
			for (int32 k = 0; k < top; ++k) bigger[k] = stack[k];
			if (stack != small) b2Free(stack);
			stack = bigger;
			cap *= 2;
		}
		stack[top++] = node.child1;
		stack[top++] = node.child2;
	}
	if (stack != small) b2Free(stack);
}

#endif
