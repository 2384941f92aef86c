// Drop-in header: the task half of the b2TaskExecutor plugin surface
// (reference: Box2D/MT/b2Task.h:27-159). User code derives from b2Task / b2RangeTask and runs
// them through an executor (b2ExecuteRangeTask); the device Step() itself does not create tasks.
#ifndef B2_TASK_H
#define B2_TASK_H

#include "Box2D/Common/b2Settings.h"

class b2StackAllocator;

struct b2ThreadContext
{
	b2StackAllocator* stack;
	uint32 threadId;
};

class b2TaskGroup
{
};

class b2Task
{
public:
	enum Type
	{
		e_broadPhaseFindContacts = 0,
		e_broadPhaseSyncFixtures,
		e_clearContactSolveFlags,
		e_clearContactSolveToiFlags,
		e_clearBodySolveFlags,
		e_clearBodySolveToiFlags,
		e_clearForces,
		e_collide,
		e_findMinToiContact,
		e_rangeTypeCount,
		e_merge = e_rangeTypeCount,
		e_solve,
		e_sort,
		e_typeCount,
		e_userTask = e_typeCount
	};

	b2Task() : m_costEstimate(0), m_taskGroup(nullptr) {}
	virtual ~b2Task() {}
	virtual void Execute(const b2ThreadContext& ctx) = 0;
	virtual Type GetType() const { return e_userTask; }
	void SetCost(uint32 costEstimate) { m_costEstimate = costEstimate; }
	uint32 GetCost() const { return m_costEstimate; }
	void SetTaskGroup(b2TaskGroup* taskGroup) { m_taskGroup = taskGroup; }
	b2TaskGroup* GetTaskGroup() const { return m_taskGroup; }

private:
	uint32 m_costEstimate;
	b2TaskGroup* m_taskGroup;
};

inline bool b2IsRangeTask(b2Task::Type type) { return type < b2Task::e_rangeTypeCount; }

struct b2RangeTaskRange
{
	b2RangeTaskRange() : begin(0), end(0) {}
	b2RangeTaskRange(uint32 beginIn, uint32 endIn) : begin(beginIn), end(endIn) {}
	uint32 GetCount() const { return end - begin; }
	uint32 begin;
	uint32 end;
};

struct b2PartitionedRange
{
	b2PartitionedRange() : count(0) {}
	b2RangeTaskRange& operator[](size_t i) { return ranges[i]; }
	const b2RangeTaskRange& operator[](size_t i) const { return ranges[i]; }
	b2RangeTaskRange ranges[b2_maxRangeSubTasks];
	uint32 count;
};

class b2RangeTask : public b2Task
{
public:
	b2RangeTask() {}
	b2RangeTask(const b2RangeTaskRange& range) : m_range(range) {}
	virtual void Execute(const b2ThreadContext& ctx, const b2RangeTaskRange& range) = 0;
	void Execute(const b2ThreadContext& ctx) final { Execute(ctx, m_range); }
	const b2RangeTaskRange& GetRange() const { return m_range; }
	void SetRange(const b2RangeTaskRange& range) { m_range = range; }

protected:
	b2RangeTaskRange m_range;
};

// Evenly divides [begin, end) into at most maxOutputRanges ranges of at least minElementsPerRange.
void b2PartitionRange(uint32 begin, uint32 end, uint32 maxOutputRanges, uint32 minElementsPerRange, b2PartitionedRange& output);

#endif
