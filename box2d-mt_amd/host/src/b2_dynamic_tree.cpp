// b2DynamicTree of the drop-in API (interface: Box2D/Collision/b2DynamicTree.h:52-130 of the reference). Own implementation,
// see the header. Balance rule and cost heuristic are the usual ones for an incremental AABB tree (perimeter as the cost
// of a node, a rotation where the heights of two siblings differ by more than one), written for this repo.
#include "Box2D/Collision/b2DynamicTree.h"

#include <string.h>

namespace
{
inline b2AABB Hull(const b2AABB& a, const b2AABB& b)
{
	b2AABB h;
	h.Combine(a, b);
	return h;
}
inline int32 Taller(int32 a, int32 b) { return a > b ? a : b; }
}

b2DynamicTree::b2DynamicTree() : m_root(b2_nullNode), m_nodes(nullptr), m_capacity(0), m_used(0), m_freeHead(b2_nullNode), m_leafCount(0)
{
}

b2DynamicTree::~b2DynamicTree()
{
	b2Free(m_nodes);
}

void b2DynamicTree::Clear()
{
	m_root = b2_nullNode;
	m_used = 0;
	m_freeHead = b2_nullNode;
	m_leafCount = 0;
}

// A node from the free list, from the untouched tail of the pool, or from a pool grown to twice its size.
int32 b2DynamicTree::TakeNode()
{
	int32 id;
	if (m_freeHead != b2_nullNode)
	{
		id = m_freeHead;
		m_freeHead = m_nodes[id].next;
	}
	else
	{
		if (m_used == m_capacity)
		{
			const int32 grown = m_capacity ? 2 * m_capacity : 16;
			b2TreeNode* pool = (b2TreeNode*)b2Alloc(grown * (int32)sizeof(b2TreeNode));
			if (m_nodes) memcpy(pool, m_nodes, (size_t)m_capacity * sizeof(b2TreeNode));
			b2Free(m_nodes);
			m_nodes = pool;
			m_capacity = grown;
		}
		id = m_used++;
	}
	b2TreeNode& n = m_nodes[id];
	n.parent = b2_nullNode;
	n.child1 = n.child2 = b2_nullNode;
	n.height = 0;
	n.userData = nullptr;
	return id;
}

void b2DynamicTree::GiveBack(int32 node)
{
	m_nodes[node].next = m_freeHead;
	m_nodes[node].height = -1;
	m_freeHead = node;
}

int32 b2DynamicTree::InsertFat(const b2AABB& fat, void* userData)
{
	const int32 leaf = TakeNode();
	m_nodes[leaf].aabb = fat;
	m_nodes[leaf].userData = userData;
	Attach(leaf);
	++m_leafCount;
	return leaf;
}

void b2DynamicTree::ReplaceFat(int32 leaf, const b2AABB& fat)
{
	Detach(leaf);
	m_nodes[leaf].aabb = fat;
	Attach(leaf);
}

int32 b2DynamicTree::CreateProxy(const b2AABB& aabb, void* userData)
{
	const b2Vec2 margin(b2_aabbExtension, b2_aabbExtension);
	b2AABB fat;
	fat.lowerBound = aabb.lowerBound - margin;
	fat.upperBound = aabb.upperBound + margin;
	return InsertFat(fat, userData);
}

void b2DynamicTree::DestroyProxy(int32 proxyId)
{
	Detach(proxyId);
	GiveBack(proxyId);
	--m_leafCount;
}

// The fat-AABB rule (b2DynamicTree.cpp:130-174 of the reference): nothing happens while the tight box stays inside the
// stored one; otherwise the new stored box is the tight one plus the margin, stretched along the predicted motion.
bool b2DynamicTree::MoveProxy(int32 proxyId, const b2AABB& aabb, const b2Vec2& displacement)
{
	if (m_nodes[proxyId].aabb.Contains(aabb)) return false;
	const b2Vec2 margin(b2_aabbExtension, b2_aabbExtension);
	b2AABB fat;
	fat.lowerBound = aabb.lowerBound - margin;
	fat.upperBound = aabb.upperBound + margin;
	const b2Vec2 ahead = b2_aabbMultiplier * displacement;
	if (ahead.x < 0.0f) fat.lowerBound.x += ahead.x; else fat.upperBound.x += ahead.x;
	if (ahead.y < 0.0f) fat.lowerBound.y += ahead.y; else fat.upperBound.y += ahead.y;
	ReplaceFat(proxyId, fat);
	return true;
}

// Hang `leaf` into the tree: walk down from the root, at every inner node weighing "make a new parent here" against "go on
// into the cheaper child" by the perimeter each choice adds; then repair boxes and heights up to the root.
void b2DynamicTree::Attach(int32 leaf)
{
	if (m_root == b2_nullNode)
	{
		m_root = leaf;
		m_nodes[leaf].parent = b2_nullNode;
		return;
	}
	const b2AABB box = m_nodes[leaf].aabb;
	int32 at = m_root;
	while (!m_nodes[at].IsLeaf())
	{
		const b2TreeNode& n = m_nodes[at];
		const float32 here = n.aabb.GetPerimeter();
		const float32 grown = Hull(n.aabb, box).GetPerimeter();
		const float32 pairHere = 2.0f * grown;            // a new parent over this node and the leaf
		const float32 passDown = 2.0f * (grown - here);   // what every node below pays for this one growing
		float32 price[2];
		const int32 kids[2] = { n.child1, n.child2 };
		for (int32 k = 0; k < 2; ++k)
		{
			const b2TreeNode& c = m_nodes[kids[k]];
			const float32 merged = Hull(c.aabb, box).GetPerimeter();
			price[k] = (c.IsLeaf() ? merged : merged - c.aabb.GetPerimeter()) + passDown;
		}
		if (pairHere < price[0] && pairHere < price[1]) break;
		at = price[0] < price[1] ? kids[0] : kids[1];
	}
	const int32 sibling = at;
	const int32 above = m_nodes[sibling].parent;
	const int32 joint = TakeNode();
	m_nodes[joint].parent = above;
	m_nodes[joint].aabb = Hull(box, m_nodes[sibling].aabb);
	m_nodes[joint].height = m_nodes[sibling].height + 1;
	m_nodes[joint].child1 = sibling;
	m_nodes[joint].child2 = leaf;
	m_nodes[sibling].parent = joint;
	m_nodes[leaf].parent = joint;
	if (above == b2_nullNode) m_root = joint;
	else if (m_nodes[above].child1 == sibling) m_nodes[above].child1 = joint;
	else m_nodes[above].child2 = joint;
	Refit(m_nodes[leaf].parent);
}

// Take `leaf` out: its sibling moves up into the place of their parent.
void b2DynamicTree::Detach(int32 leaf)
{
	if (leaf == m_root)
	{
		m_root = b2_nullNode;
		return;
	}
	const int32 joint = m_nodes[leaf].parent;
	const int32 above = m_nodes[joint].parent;
	const int32 sibling = m_nodes[joint].child1 == leaf ? m_nodes[joint].child2 : m_nodes[joint].child1;
	if (above == b2_nullNode)
	{
		m_root = sibling;
		m_nodes[sibling].parent = b2_nullNode;
	}
	else
	{
		if (m_nodes[above].child1 == joint) m_nodes[above].child1 = sibling; else m_nodes[above].child2 = sibling;
		m_nodes[sibling].parent = above;
	}
	GiveBack(joint);
	Refit(above);
	m_nodes[leaf].parent = b2_nullNode;
}

// From `node` to the root: rotate where needed, recompute height and box.
void b2DynamicTree::Refit(int32 node)
{
	while (node != b2_nullNode)
	{
		node = Rebalance(node);
		b2TreeNode& n = m_nodes[node];
		n.height = 1 + Taller(m_nodes[n.child1].height, m_nodes[n.child2].height);
		n.aabb = Hull(m_nodes[n.child1].aabb, m_nodes[n.child2].aabb);
		node = n.parent;
	}
}

// If one child of `a` is more than one level taller than the other, lift it: the tall child takes a's place, a becomes its
// child and adopts the shorter of the tall child's children. Returns the root of the subtree afterwards.
int32 b2DynamicTree::Rebalance(int32 a)
{
	b2TreeNode& A = m_nodes[a];
	if (A.IsLeaf() || A.height < 2) return a;
	const int32 lean = m_nodes[A.child2].height - m_nodes[A.child1].height;
	if (lean >= -1 && lean <= 1) return a;
	const bool right = lean > 1;
	const int32 up = right ? A.child2 : A.child1;      // the tall child, lifted
	const int32 other = right ? A.child1 : A.child2;
	b2TreeNode& U = m_nodes[up];
	int32 g1 = U.child1, g2 = U.child2;
	// the taller grandchild stays under `up`, the other goes to `a`
	const int32 keep = m_nodes[g1].height > m_nodes[g2].height ? g1 : g2;
	const int32 hand = keep == g1 ? g2 : g1;
	// `up` replaces `a` under a's parent
	U.parent = A.parent;
	if (A.parent == b2_nullNode) m_root = up;
	else if (m_nodes[A.parent].child1 == a) m_nodes[A.parent].child1 = up;
	else m_nodes[A.parent].child2 = up;
	U.child1 = a;
	U.child2 = keep;
	A.parent = up;
	m_nodes[keep].parent = up;
	if (right) { A.child1 = other; A.child2 = hand; } else { A.child1 = hand; A.child2 = other; }
	m_nodes[hand].parent = a;
	A.aabb = Hull(m_nodes[A.child1].aabb, m_nodes[A.child2].aabb);
	A.height = 1 + Taller(m_nodes[A.child1].height, m_nodes[A.child2].height);
	U.aabb = Hull(A.aabb, m_nodes[keep].aabb);
	U.height = 1 + Taller(A.height, m_nodes[keep].height);
	return up;
}

int32 b2DynamicTree::GetMaxBalance() const
{
	int32 worst = 0;
	for (int32 i = 0; i < m_used; ++i)
	{
		const b2TreeNode& n = m_nodes[i];
		if (n.height <= 1) continue; // free (-1), leaf (0) or a pair of leaves (1)
		const int32 d = m_nodes[n.child2].height - m_nodes[n.child1].height;
		worst = Taller(worst, d < 0 ? -d : d);
	}
	return worst;
}

float32 b2DynamicTree::GetAreaRatio() const
{
	if (m_root == b2_nullNode) return 0.0f;
	const float32 whole = m_nodes[m_root].aabb.GetPerimeter();
	float32 sum = 0.0f;
	for (int32 i = 0; i < m_used; ++i)
	{
		if (m_nodes[i].height >= 0) sum += m_nodes[i].aabb.GetPerimeter();
	}
	return sum / whole;
}

// The tree rebuilt greedily from its leaves: again and again the two roots whose union is smallest get a common parent.
void b2DynamicTree::RebuildBottomUp()
{
	if (m_leafCount == 0) return;
	int32* roots = (int32*)b2Alloc(m_leafCount * (int32)sizeof(int32));
	int32 n = 0;
	for (int32 i = 0; i < m_used; ++i)
	{
		if (m_nodes[i].height < 0) continue;
		if (m_nodes[i].IsLeaf())
		{
			m_nodes[i].parent = b2_nullNode;
			roots[n++] = i;
		}
		else GiveBack(i);
	}
	while (n > 1)
	{
		float32 best = b2_maxFloat;
		int32 bi = 0, bj = 1;
		for (int32 i = 0; i < n; ++i)
			for (int32 j = i + 1; j < n; ++j)
			{
				const float32 cost = Hull(m_nodes[roots[i]].aabb, m_nodes[roots[j]].aabb).GetPerimeter();
				if (cost < best) { best = cost; bi = i; bj = j; }
			}
		const int32 a = roots[bi], b = roots[bj];
		const int32 joint = TakeNode();
		m_nodes[joint].child1 = a;
		m_nodes[joint].child2 = b;
		m_nodes[joint].height = 1 + Taller(m_nodes[a].height, m_nodes[b].height);
		m_nodes[joint].aabb = Hull(m_nodes[a].aabb, m_nodes[b].aabb);
		m_nodes[a].parent = m_nodes[b].parent = joint;
		roots[bj] = roots[n - 1];
		roots[bi] = joint;
		--n;
	}
	m_root = roots[0];
	b2Free(roots);
}

void b2DynamicTree::ShiftOrigin(const b2Vec2& newOrigin)
{
	for (int32 i = 0; i < m_used; ++i)
	{
		if (m_nodes[i].height < 0) continue;
		m_nodes[i].aabb.lowerBound -= newOrigin;
		m_nodes[i].aabb.upperBound -= newOrigin;
	}
}

void b2DynamicTree::CheckSubtree(int32 node, int32 expectedParent) const
{
	if (node == b2_nullNode) return;
	const b2TreeNode& n = m_nodes[node];
	b2Assert(n.parent == expectedParent);
	if (n.IsLeaf())
	{
		b2Assert(n.child2 == b2_nullNode && n.height == 0);
		return;
	}
	b2Assert(n.height == 1 + Taller(m_nodes[n.child1].height, m_nodes[n.child2].height));
	b2Assert(n.aabb.Contains(m_nodes[n.child1].aabb) && n.aabb.Contains(m_nodes[n.child2].aabb));
	CheckSubtree(n.child1, node);
	CheckSubtree(n.child2, node);
	B2_NOT_USED(expectedParent);
}

void b2DynamicTree::Validate() const
{
	CheckSubtree(m_root, b2_nullNode);
}
