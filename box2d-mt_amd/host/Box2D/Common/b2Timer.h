// Drop-in header: millisecond wall timer (reference: Box2D/Common/b2Timer.h).
#ifndef B2_TIMER_H
#define B2_TIMER_H

#include "Box2D/Common/b2Settings.h"
#include <chrono>

class b2Timer
{
public:
	b2Timer() { Reset(); }
	void Reset() { m_start = std::chrono::steady_clock::now(); }
	float32 GetMilliseconds() const
	{
		return std::chrono::duration<float32, std::milli>(std::chrono::steady_clock::now() - m_start).count();
	}

private:
	std::chrono::steady_clock::time_point m_start;
};

#endif
