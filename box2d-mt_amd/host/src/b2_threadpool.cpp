// Host-side thread pool executor of the drop-in API (own implementation of the behaviour described
// by the reference's Box2D/MT/b2ThreadPool.{h,cpp} and b2Task.cpp:22-71): a vector queue ordered by
// task cost, N-1 worker threads plus the user thread helping inside Wait().
#include "Box2D/MT/b2ThreadPool.h"
#include <chrono>
#include "Box2D/MT/b2MtUtil.h"

#include <algorithm>

void b2PartitionRange(uint32 begin, uint32 end, uint32 maxOutputRanges, uint32 minElementsPerRange, b2PartitionedRange& output)
{
	output.count = 0;
	if (end <= begin) return;
	uint32 elements = end - begin;
	if (maxOutputRanges < 1) maxOutputRanges = 1;
	if (maxOutputRanges > b2_maxRangeSubTasks) maxOutputRanges = b2_maxRangeSubTasks;
	if (minElementsPerRange < 1) minElementsPerRange = 1;
	uint32 ranges = elements / minElementsPerRange;
	if (ranges < 1) ranges = 1;
	if (ranges > maxOutputRanges) ranges = maxOutputRanges;
	uint32 per = elements / ranges;
	uint32 extra = elements % ranges;
	uint32 at = begin;
	for (uint32 i = 0; i < ranges; ++i)
	{
		uint32 n = per + (i < extra ? 1 : 0);
		output.ranges[i].begin = at;
		output.ranges[i].end = at + n;
		at += n;
	}
	output.count = ranges;
}

b2ThreadPool::b2ThreadPool(const b2ThreadPoolOptions& options)
{
	m_shutdown = false;
	m_threadCount = 0;
	m_lockMilliseconds = 0.0f;
	SetBusyWaitTimeout(options.busyWaitTimeoutMs);
	int32 total = options.totalThreadCount;
	if (total < 0) total = (int32)std::thread::hardware_concurrency();
	if (total < 1) total = 1;
	if (total > b2_maxThreads) total = b2_maxThreads;
	Start(total);
}

b2ThreadPool::~b2ThreadPool()
{
	Shutdown();
}

void b2ThreadPool::Start(int32 totalThreads)
{
	m_shutdown = false;
	m_threadCount = totalThreads;
	for (int32 i = 1; i < totalThreads; ++i)
	{
		m_stacks.push_back(new b2StackAllocator);
		m_threads.emplace_back(&b2ThreadPool::WorkerMain, this, (uint32)i);
	}
}

void b2ThreadPool::Shutdown()
{
	{
		std::lock_guard<std::mutex> lock(m_mutex);
		m_shutdown = true;
	}
	m_cv.notify_all();
	for (size_t i = 0; i < m_threads.size(); ++i) m_threads[i].join();
	m_threads.clear();
	for (size_t i = 0; i < m_stacks.size(); ++i) delete m_stacks[i];
	m_stacks.clear();
}

void b2ThreadPool::Restart(int32 threadCount)
{
	Shutdown();
	if (threadCount < 1) threadCount = 1;
	if (threadCount > b2_maxThreads) threadCount = b2_maxThreads;
	Start(threadCount);
}

b2Task* b2ThreadPool::Pop()
{
	// highest cost first
	if (m_queue.empty()) return nullptr;
	size_t best = 0;
	for (size_t i = 1; i < m_queue.size(); ++i)
	{
		if (m_queue[i]->GetCost() > m_queue[best]->GetCost()) best = i;
	}
	b2Task* t = m_queue[best];
	m_queue[best] = m_queue.back();
	m_queue.pop_back();
	return t;
}

void b2ThreadPool::SubmitTasks(b2ThreadPoolTaskGroup& group, b2Task** tasks, uint32 count)
{
	group.m_remaining.fetch_add(count);
	{
		std::lock_guard<std::mutex> lock(m_mutex);
		for (uint32 i = 0; i < count; ++i)
		{
			tasks[i]->SetTaskGroup(&group);
			m_queue.push_back(tasks[i]);
		}
	}
	m_cv.notify_all();
}

void b2ThreadPool::SubmitTask(b2ThreadPoolTaskGroup& group, b2Task* task)
{
	SubmitTasks(group, &task, 1);
}

void b2ThreadPool::Wait(const b2ThreadPoolTaskGroup& group, const b2ThreadContext& ctx)
{
	b2ThreadPoolTaskGroup& g = const_cast<b2ThreadPoolTaskGroup&>(group);
	while (g.m_remaining.load() > 0)
	{
		b2Task* task = nullptr;
		{
			std::lock_guard<std::mutex> lock(m_mutex);
			task = Pop();
		}
		if (task)
		{
			b2ThreadPoolTaskGroup* tg = static_cast<b2ThreadPoolTaskGroup*>(task->GetTaskGroup());
			task->Execute(ctx);
			tg->m_remaining.fetch_sub(1);
		}
		else
		{
			std::this_thread::yield();
		}
	}
}

void b2ThreadPool::WorkerMain(uint32 threadId)
{
	b2ThreadContext ctx;
	ctx.stack = m_stacks[threadId - 1];
	ctx.threadId = threadId;
	for (;;)
	{
		b2Task* task = nullptr;
		{
			// an idle worker spins for a task for the busy-wait time first (b2ThreadPoolOptions::busyWaitTimeoutMs,
			// SetBusyWaitTimeout: a wake-up through the condition variable costs more than a short phase of a step lasts)
			const long long spinNs = m_busyWaitNs.load(std::memory_order_relaxed);
			if (spinNs > 0)
			{
				const auto until = std::chrono::steady_clock::now() + std::chrono::nanoseconds(spinNs);
				while (std::chrono::steady_clock::now() < until)
				{
					std::unique_lock<std::mutex> peek(m_mutex, std::try_to_lock);
					if (peek.owns_lock() && (m_shutdown || !m_queue.empty())) break;
				}
			}
			std::unique_lock<std::mutex> lock(m_mutex);
			m_cv.wait(lock, [this] { return m_shutdown || !m_queue.empty(); });
			if (m_shutdown) return;
			task = Pop();
		}
		if (task)
		{
			b2ThreadPoolTaskGroup* tg = static_cast<b2ThreadPoolTaskGroup*>(task->GetTaskGroup());
			task->Execute(ctx);
			tg->m_remaining.fetch_sub(1);
		}
	}
}

b2ThreadPoolTaskExecutor::b2ThreadPoolTaskExecutor(const b2ThreadPoolOptions& options)
	: m_threadPool(options), m_taskGroup(m_threadPool), m_taskGroupInUse(false)
{
}

b2TaskGroup* b2ThreadPoolTaskExecutor::AcquireTaskGroup()
{
	b2Assert(m_taskGroupInUse == false);
	m_taskGroupInUse = true;
	return &m_taskGroup;
}

void b2ThreadPoolTaskExecutor::ReleaseTaskGroup(b2TaskGroup* taskGroup)
{
	B2_NOT_USED(taskGroup);
	m_taskGroupInUse = false;
}

void b2ThreadPoolTaskExecutor::PartitionRange(b2Task::Type type, uint32 begin, uint32 end, b2PartitionedRange& output)
{
	B2_NOT_USED(type);
	b2PartitionRange(begin, end, (uint32)m_threadPool.GetThreadCount(), 1, output);
}

void b2ThreadPoolTaskExecutor::SubmitTask(b2TaskGroup* taskGroup, b2Task* task)
{
	m_threadPool.SubmitTask(*static_cast<b2ThreadPoolTaskGroup*>(taskGroup), task);
}

void b2ThreadPoolTaskExecutor::SubmitTasks(b2TaskGroup* taskGroup, b2Task** tasks, uint32 count)
{
	m_threadPool.SubmitTasks(*static_cast<b2ThreadPoolTaskGroup*>(taskGroup), tasks, count);
}

void b2ThreadPoolTaskExecutor::Wait(b2TaskGroup* taskGroup, const b2ThreadContext& ctx)
{
	m_threadPool.Wait(*static_cast<b2ThreadPoolTaskGroup*>(taskGroup), ctx);
}
