// b2d_wave.h - wave64 aggregation of atomics that many lanes aim at few addresses.
//
// A single large island means ten thousand lanes doing atomicAdd / atomicMin / atomicMax on ONE root
// slot (census, penetration maximum, sleep minimum): that serialises in L2 (measured: 80-150 us per
// kernel on Pyramid-10k). These helpers combine the lanes of a wave first, so the memory system sees
// one atomic per wave. They must be called from wave-uniform control flow (every lane of the wave
// calls, lanes without work pass valid = false).
#ifndef B2D_WAVE_H
#define B2D_WAVE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ int waveLane() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ int waveSumInt(int v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
	return v;
}

__device__ __forceinline__ int waveMinInt(int v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		int o = __shfl_xor(v, off);
		v = o < v ? o : v;
	}
	return v;
}

__device__ __forceinline__ uint32_t waveMaxU32(uint32_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		uint32_t o = (uint32_t)__shfl_xor((int)v, off);
		v = o > v ? o : v;
	}
	return v;
}

__device__ __forceinline__ uint32_t waveMinU32(uint32_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		uint32_t o = (uint32_t)__shfl_xor((int)v, off);
		v = o < v ? o : v;
	}
	return v;
}

// Is `key` the same in every valid lane? Returns that key through *k0 (undefined if no lane is valid).
__device__ __forceinline__ bool waveUniformKey(int key, bool valid, int* k0, int* leader)
{
	unsigned long long vm = __ballot(valid);
	if (vm == 0ull)
	{
		*leader = -1;
		*k0 = 0;
		return true;
	}
	*leader = __ffsll((long long)vm) - 1;
	*k0 = __shfl(key, *leader);
	return __all(!valid || key == *k0) != 0;
}

__device__ __forceinline__ void waveAtomicAddInt(int* base, int key, int val, bool valid)
{
	int k0, leader;
	if (waveUniformKey(key, valid, &k0, &leader))
	{
		int s = waveSumInt(valid ? val : 0);
		if (waveLane() == leader) atomicAdd(&base[k0], s);
	}
	else if (valid)
	{
		atomicAdd(&base[key], val);
	}
}

__device__ __forceinline__ void waveAtomicMinInt(int* base, int key, int val, bool valid)
{
	int k0, leader;
	if (waveUniformKey(key, valid, &k0, &leader))
	{
		int s = waveMinInt(valid ? val : 0x7fffffff);
		if (waveLane() == leader) atomicMin(&base[k0], s);
	}
	else if (valid)
	{
		atomicMin(&base[key], val);
	}
}

__device__ __forceinline__ void waveAtomicMaxU32(uint32_t* base, int key, uint32_t val, bool valid)
{
	int k0, leader;
	if (waveUniformKey(key, valid, &k0, &leader))
	{
		uint32_t s = waveMaxU32(valid ? val : 0u);
		if (waveLane() == leader) atomicMax(&base[k0], s);
	}
	else if (valid)
	{
		atomicMax(&base[key], val);
	}
}

__device__ __forceinline__ void waveAtomicMinU32(uint32_t* base, int key, uint32_t val, bool valid)
{
	int k0, leader;
	if (waveUniformKey(key, valid, &k0, &leader))
	{
		uint32_t s = waveMinU32(valid ? val : 0xffffffffu);
		if (waveLane() == leader) atomicMin(&base[k0], s);
	}
	else if (valid)
	{
		atomicMin(&base[key], val);
	}
}

// Every valid lane gets a unique slot of counter[key]: one atomicAdd per distinct key per wave.
__device__ __forceinline__ int waveKeyedAlloc(int* counter, int key, bool valid)
{
	int result = 0;
	bool pending = valid;
	const int lane = waveLane();
	while (__any(pending))
	{
		unsigned long long pm = __ballot(pending);
		const int leader = __ffsll((long long)pm) - 1;
		const int k = __shfl(key, leader);
		const bool mine = pending && key == k;
		unsigned long long mm = __ballot(mine);
		int start = 0;
		if (lane == leader) start = atomicAdd(&counter[k], __popcll(mm));
		start = __shfl(start, leader);
		if (mine) result = start + __popcll(mm & ((1ull << lane) - 1ull));
		pending = pending && !mine;
	}
	return result;
}

// The same for a whole 256-lane workgroup and keys in [0, 64]: ranks through an LDS histogram, then ONE global atomicAdd
// per key that occurs, all of them in flight together (waveKeyedAlloc pays one atomic round trip per distinct key and
// wave, one after the other: seven colours on the 10k-body pyramid made k_color_fill 24 us). Every thread of the
// workgroup must call it (it contains barriers). `fetch` = false: only count (no slot returned).
__device__ __forceinline__ int blockKeyedAlloc65(int* counter, int key, bool valid, bool fetch)
{
	__shared__ int s_cnt[65], s_base[65];
	__syncthreads(); // a previous call's readers are done
	if (threadIdx.x < 65) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	int local = 0;
	if (valid) local = atomicAdd(&s_cnt[key], 1);
	__syncthreads();
	if (threadIdx.x < 65)
	{
		const int c = s_cnt[threadIdx.x];
		if (c > 0)
		{
			if (fetch) s_base[threadIdx.x] = atomicAdd(&counter[threadIdx.x], c);
			else atomicAdd(&counter[threadIdx.x], c);
		}
	}
	if (!fetch) return 0;
	__syncthreads();
	return valid ? s_base[key] + local : 0;
}

#endif
