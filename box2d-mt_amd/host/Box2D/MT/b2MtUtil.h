// Drop-in header: submit helpers (reference: Box2D/MT/b2MtUtil.h:27-79).
#ifndef B2_MT_UTIL_H
#define B2_MT_UTIL_H

#include "Box2D/MT/b2TaskExecutor.h"

inline b2ThreadContext b2MainThreadCtx(b2StackAllocator* stack)
{
	b2ThreadContext ctx;
	ctx.stack = stack;
	ctx.threadId = 0;
	return ctx;
}

inline void b2SubmitTask(b2TaskExecutor& executor, b2TaskGroup* taskGroup, b2Task* task)
{
	task->SetTaskGroup(taskGroup);
	executor.SubmitTask(taskGroup, task);
}

template <typename TaskType>
inline void b2SubmitTasks(b2TaskExecutor& executor, b2TaskGroup* taskGroup, TaskType* tasks, uint32 count)
{
	b2Task* ptrs[b2_maxRangeSubTasks];
	b2Assert(count <= b2_maxRangeSubTasks);
	for (uint32 i = 0; i < count; ++i)
	{
		tasks[i].SetTaskGroup(taskGroup);
		ptrs[i] = tasks + i;
	}
	executor.SubmitTasks(taskGroup, ptrs, count);
}

// Runs a user range task on the executor, split into sub-ranges, and waits for it.
template <typename TaskType>
inline void b2ExecuteRangeTask(b2TaskExecutor& executor, const TaskType& task)
{
	b2PartitionedRange ranges;
	executor.PartitionRange(task.GetType(), task.GetRange().begin, task.GetRange().end, ranges);
	TaskType subTasks[b2_maxRangeSubTasks];
	for (uint32 i = 0; i < ranges.count; ++i)
	{
		subTasks[i] = task;
		subTasks[i].SetRange(ranges[i]);
	}
	b2TaskGroup* group = executor.AcquireTaskGroup();
	b2SubmitTasks(executor, group, subTasks, ranges.count);
	b2StackAllocator* stack = new b2StackAllocator;
	executor.Wait(group, b2MainThreadCtx(stack));
	delete stack;
	executor.ReleaseTaskGroup(group);
}

#endif
