// Drop-in header: b2ThreadPool / b2ThreadPoolTaskExecutor (reference: Box2D/MT/b2ThreadPool.h:33-169).
// Own implementation: a small std::thread pool with one queue per task group; the user thread helps
// inside Wait(). Same public surface (options, Restart, timers) so the Testbed can embed it by value.
#ifndef B2_THREAD_POOL_H
#define B2_THREAD_POOL_H

#include "Box2D/MT/b2TaskExecutor.h"
#include "Box2D/Common/b2StackAllocator.h"

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

struct b2ThreadPoolOptions
{
	b2ThreadPoolOptions()
	{
		totalThreadCount = -1;
		busyWaitTimeoutMs = 0.03f;
	}
	int32 totalThreadCount;     // includes the user thread; -1 = logical cores
	float32 busyWaitTimeoutMs;
};

class b2ThreadPool;

class b2ThreadPoolTaskGroup : public b2TaskGroup
{
public:
	explicit b2ThreadPoolTaskGroup(b2ThreadPool& threadPool) : m_pool(&threadPool), m_remaining(0) {}

private:
	friend class b2ThreadPool;
	b2ThreadPool* m_pool;
	std::atomic<uint32> m_remaining;
};

class b2ThreadPool
{
public:
	explicit b2ThreadPool(const b2ThreadPoolOptions& options = b2ThreadPoolOptions());
	~b2ThreadPool();

	/// How long an idle worker spins for a task before it waits on the condition variable (b2ThreadPool.h:79-81 of the reference)
	void SetBusyWaitTimeout(float32 busyWaitTimeoutMs) { m_busyWaitNs.store((long long)(busyWaitTimeoutMs < 0.0f ? 0.0f : busyWaitTimeoutMs * 1.0e6f)); }
	void SubmitTasks(b2ThreadPoolTaskGroup& group, b2Task** tasks, uint32 count);
	void SubmitTask(b2ThreadPoolTaskGroup& group, b2Task* task);
	void Wait(const b2ThreadPoolTaskGroup& group, const b2ThreadContext& ctx);
	void Restart(int32 threadCount);
	int32 GetThreadCount() const { return m_threadCount; }
	float32 GetLockMilliseconds() const { return m_lockMilliseconds; }
	void ResetTimers() { m_lockMilliseconds = 0.0f; }

private:
	void Start(int32 totalThreads);
	void Shutdown();
	void WorkerMain(uint32 threadId);
	b2Task* Pop();

	std::mutex m_mutex;
	std::condition_variable m_cv;
	std::vector<b2Task*> m_queue;
	std::vector<std::thread> m_threads;
	std::vector<b2StackAllocator*> m_stacks;
	bool m_shutdown;
	std::atomic<long long> m_busyWaitNs{0};
	int32 m_threadCount;
	float32 m_lockMilliseconds;
};

class b2ThreadPoolTaskExecutor : public b2TaskExecutor
{
public:
	explicit b2ThreadPoolTaskExecutor(const b2ThreadPoolOptions& options = b2ThreadPoolOptions());

	b2ThreadPool* GetThreadPool() { return &m_threadPool; }
	const b2ThreadPool* GetThreadPool() const { return &m_threadPool; }

	uint32 GetThreadCount() const override { return (uint32)m_threadPool.GetThreadCount(); }
	b2TaskGroup* AcquireTaskGroup() override;
	void ReleaseTaskGroup(b2TaskGroup* taskGroup) override;
	void PartitionRange(b2Task::Type type, uint32 begin, uint32 end, b2PartitionedRange& output) override;
	void SubmitTask(b2TaskGroup* taskGroup, b2Task* task) override;
	void SubmitTasks(b2TaskGroup* taskGroup, b2Task** tasks, uint32 count) override;
	void Wait(b2TaskGroup* taskGroup, const b2ThreadContext& ctx) override;

private:
	b2ThreadPool m_threadPool;
	b2ThreadPoolTaskGroup m_taskGroup;
	bool m_taskGroupInUse;
};

#endif
