// Drop-in header: per-thread scratch stack handed to tasks through b2ThreadContext
// (reference: Box2D/Common/b2StackAllocator.h:24-60). Kept as a simple LIFO arena.
#ifndef B2_STACK_ALLOCATOR_H
#define B2_STACK_ALLOCATOR_H

#include "Box2D/Common/b2Settings.h"

const int32 b2_stackSize = 256 * 1024;
const int32 b2_maxStackEntries = 32;

class b2StackAllocator
{
public:
	b2StackAllocator() : m_top(0), m_count(0) {}
	void* Allocate(int32 size)
	{
		b2Assert(m_count < b2_maxStackEntries);
		Entry& e = m_entries[m_count++];
		e.size = size;
		if (m_top + size > b2_stackSize)
		{
			e.data = (char*)b2Alloc(size);
			e.heap = true;
		}
		else
		{
			e.data = m_data + m_top;
			e.heap = false;
			m_top += size;
		}
		return e.data;
	}
	void Free(void* p)
	{
		b2Assert(m_count > 0);
		Entry& e = m_entries[--m_count];
		b2Assert(p == e.data);
		B2_NOT_USED(p);
		if (e.heap) b2Free(e.data); else m_top -= e.size;
	}

private:
	struct Entry
	{
		char* data;
		int32 size;
		bool heap;
	};
	char m_data[b2_stackSize];
	int32 m_top;
	Entry m_entries[b2_maxStackEntries];
	int32 m_count;
};

#endif
