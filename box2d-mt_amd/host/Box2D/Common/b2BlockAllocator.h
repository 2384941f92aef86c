// Drop-in header: the small-object allocator type that appears in public signatures
// (b2Shape::Clone). The MI355X build keeps simulation state in HBM; this allocator only serves
// host-side clones and is a thin wrapper over b2Alloc / b2Free.
#ifndef B2_BLOCK_ALLOCATOR_H
#define B2_BLOCK_ALLOCATOR_H

#include "Box2D/Common/b2Settings.h"

const int32 b2_maxBlockSize = 1152;

class b2BlockAllocator
{
public:
	void* Allocate(int32 size) { return b2Alloc(size); }
	void Free(void* p, int32 size) { B2_NOT_USED(size); b2Free(p); }
	void Clear() {}
};

#endif
