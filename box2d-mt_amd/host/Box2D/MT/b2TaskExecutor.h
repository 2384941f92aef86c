// Drop-in header: the executor plugin interface (reference: Box2D/MT/b2TaskExecutor.h:27-79).
// b2World::Step takes one by reference exactly like the reference. In the MI355X build the physics
// phases are HIP kernels, so the executor only runs host-side residual work (user range tasks,
// listener callbacks).
#ifndef B2_TASK_EXECUTOR_H
#define B2_TASK_EXECUTOR_H

#include "Box2D/MT/b2Task.h"

class b2World;

class b2TaskExecutor
{
public:
	virtual ~b2TaskExecutor() {}
	virtual uint32 GetThreadCount() const = 0;
	virtual void SubmitTask(b2TaskGroup* taskGroup, b2Task* task)
	{
		B2_NOT_USED(taskGroup);
		B2_NOT_USED(task);
	}
	virtual void Wait(b2TaskGroup* taskGroup, const b2ThreadContext& ctx)
	{
		B2_NOT_USED(taskGroup);
		B2_NOT_USED(ctx);
	}
	virtual void SubmitTasks(b2TaskGroup* taskGroup, b2Task** tasks, uint32 count)
	{
		for (uint32 i = 0; i < count; ++i) SubmitTask(taskGroup, tasks[i]);
	}
	virtual b2TaskGroup* AcquireTaskGroup() { return nullptr; }
	virtual void ReleaseTaskGroup(b2TaskGroup* taskGroup) { B2_NOT_USED(taskGroup); }
	virtual void PartitionRange(b2Task::Type type, uint32 begin, uint32 end, b2PartitionedRange& output)
	{
		B2_NOT_USED(type);
		output.ranges[0].begin = begin;
		output.ranges[0].end = end;
		output.count = 1;
	}
};

#endif
