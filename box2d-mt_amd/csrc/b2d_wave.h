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

// One combine operation over the lanes of a wave that aim at the same slot: the lanes holding the key of the first pending
// lane are reduced and their leader issues ONE atomic; that is done twice (the two most frequent keys of a wave in practice:
// "the big island" and "something else"), what is left goes out lane by lane. A wave whose valid lanes all share a key
// - the usual case on a single large island - costs one round; bodies of a big island interleaved with free bodies (the
// Tumbler: 35 000 of 100 000 bodies on one root, spread over every wave) no longer send one atomic per lane to that root
// (measured there: k_island_flatten 192 us, nearly all of it the same-address queue in L2).
// Integer sums / minima / maxima do not depend on the order of combination: the results are the same bits either way.
// WORTH(slot, value): minima / maxima look first (a load past the L2) and join the queue of the word's atomics only if they
// would change it - the words only move one way while a kernel runs, a stale read costs an atomic, never a result.
#define B2D_WAVE_ATOMIC(NAME, T, REDUCE, IDENTITY, ATOMIC, WORTH)                                  \
	__device__ __forceinline__ void NAME(T* base, int key, T val, bool valid)                      \
	{                                                                                              \
		const int lane = waveLane();                                                               \
		bool pending = valid;                                                                      \
		for (int round = 0; round < 2; ++round)                                                    \
		{                                                                                          \
			const unsigned long long pm = __ballot(pending);                                       \
			if (pm == 0ull) return;                                                                \
			const int leader = __ffsll((long long)pm) - 1;                                         \
			const int k0 = __shfl(key, leader);                                                    \
			const bool mine = pending && key == k0;                                                \
			const T s = REDUCE(mine ? val : (T)(IDENTITY));                                        \
			if (lane == leader && WORTH(&base[k0], s)) ATOMIC(&base[k0], s);                       \
			pending = pending && !mine;                                                            \
		}                                                                                          \
		if (pending && WORTH(&base[key], val)) ATOMIC(&base[key], val);                            \
	}

#define B2D_WORTH_ALWAYS(P, V) true
#define B2D_WORTH_BELOW(P, V) ((V) < __hip_atomic_load((P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#define B2D_WORTH_ABOVE(P, V) ((V) > __hip_atomic_load((P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
B2D_WAVE_ATOMIC(waveAtomicAddInt, int, waveSumInt, 0, atomicAdd, B2D_WORTH_ALWAYS)
B2D_WAVE_ATOMIC(waveAtomicMinInt, int, waveMinInt, 0x7fffffff, atomicMin, B2D_WORTH_BELOW)
B2D_WAVE_ATOMIC(waveAtomicMaxU32, uint32_t, waveMaxU32, 0u, atomicMax, B2D_WORTH_ALWAYS) // (inside the resident solvers: fire and forget, no load to wait for)
B2D_WAVE_ATOMIC(waveAtomicMaxU32Guarded, uint32_t, waveMaxU32, 0u, atomicMax, B2D_WORTH_ABOVE) // (launch per colour: the launch ends when the word's queue has drained)
B2D_WAVE_ATOMIC(waveAtomicMinU32, uint32_t, waveMinU32, 0xffffffffu, atomicMin, B2D_WORTH_BELOW)
#undef B2D_WAVE_ATOMIC

// A maximum per key that (nearly) every lane of a launch offers to the SAME key - the penetration of the one large island a
// colour launch works on (DW::rootPen): kept per lane over the kernel's loop (blockMaxU32Offer) and offered once per
// WORKGROUP at its end (blockMaxU32Flush) when all its lanes hold one key - else wave by wave as before. One look past the
// L2 per wave was 640 of them on one word at the end of every position launch. Every thread calls both, convergently.
struct BlockMaxU32
{
	int key;      // -1: nothing yet
	uint32_t val;
};
__device__ __forceinline__ void blockMaxU32Offer(BlockMaxU32& run, uint32_t* base, int key, uint32_t val, bool valid)
{
	// (a lane that meets a second key sends the first one on its way - a launch over several islands)
	const bool other = valid && run.key >= 0 && run.key != key;
	if (__ballot(other) != 0ull) waveAtomicMaxU32Guarded(base, run.key, run.val, other);
	if (valid)
	{
		if (run.key != key) { run.key = key; run.val = val; }
		else run.val = val > run.val ? val : run.val;
	}
}
__device__ __forceinline__ void blockMaxU32Flush(const BlockMaxU32& run, uint32_t* base)
{
	__shared__ int s_k[16];
	__shared__ uint32_t s_v[16];
	__shared__ int s_one;
	const bool have = run.key >= 0;
	const unsigned long long pm = __ballot(have);
	int k0 = -1;
	uint32_t m = 0u;
	bool uniform = true;
	if (pm != 0ull)
	{
		k0 = __shfl(run.key, __ffsll((long long)pm) - 1);
		uniform = __ballot(have && run.key != k0) == 0ull;
		m = waveMaxU32(have && run.key == k0 ? run.val : 0u);
	}
	if (waveLane() == 0) { s_k[threadIdx.x >> 6] = uniform ? k0 : -2; s_v[threadIdx.x >> 6] = m; }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		const int nw = (int)((blockDim.x + 63u) >> 6);
		int k = -1;
		uint32_t v = 0u;
		bool one = true;
		for (int q = 0; q < nw; ++q)
		{
			if (s_k[q] == -1) continue;
			if (s_k[q] == -2 || (k >= 0 && s_k[q] != k)) { one = false; break; }
			k = s_k[q];
			v = s_v[q] > v ? s_v[q] : v;
		}
		s_one = one ? 1 : 0;
		if (one && k >= 0 && v > __hip_atomic_load(&base[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&base[k], v);
	}
	__syncthreads();
	if (!s_one) waveAtomicMaxU32Guarded(base, run.key, run.val, have);
}

// Counters every lane of a kernel adds to (island and contact censuses): ONE atomic per workgroup. Same-address atomics are
// served one after the other in L2, ~6-10 ns each - one per wave of a pass over a million bodies is 16 000 of them, 100 us
// and more, longer than the pass itself. Every lane of the workgroup must call (barriers inside).
__device__ __forceinline__ void blockAtomicAddInt2(int* a0, int v0, int* a1, int v1)
{
	__shared__ int s_sum[2];
	if (threadIdx.x == 0) { s_sum[0] = 0; s_sum[1] = 0; }
	__syncthreads();
	v0 = waveSumInt(v0);
	v1 = waveSumInt(v1);
	if (waveLane() == 0)
	{
		if (v0) atomicAdd(&s_sum[0], v0);
		if (v1) atomicAdd(&s_sum[1], v1);
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		if (s_sum[0]) atomicAdd(a0, s_sum[0]);
		if (s_sum[1]) atomicAdd(a1, s_sum[1]);
	}
}

// A running maximum many waves offer to: most offers do not raise it - look first (a plain load), add to the queue of the
// word's atomics only then. (The value only grows while the kernel runs: a stale read can cost an atomic, never lose a maximum.)
__device__ __forceinline__ void atomicMaxIfAbove(int* addr, int v)
{
	if (v > __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(addr, v);
}
__device__ __forceinline__ void atomicMaxIfAbove(uint32_t* addr, uint32_t v)
{
	if (v > __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(addr, v);
}

// ... and once per WORKGROUP where every lane has a candidate (kept in a register over the kernel's loop): even the look is a
// load past the L2 of one word that every wave of the launch wants - 32 000 of them were 80 of k_island_flatten's 99 us on a
// million bodies (round 5). EVERY thread of the workgroup calls it, convergently (a caller that leaves early must do so
// uniformly, before the call: k_bp_clear's `nMoves == 0`); workgroups of whole waves, at most 16 of them; values <= 0 are no
// offer. The scratch is shared by all calls of a kernel: the trailing barrier makes a second call safe.
__device__ __forceinline__ void blockAtomicMaxIfAbove(int* addr, int v)
{
	__shared__ int s_blockMax[16];
	for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = o > v ? o : v; }
	if ((threadIdx.x & 63u) == 0) s_blockMax[(threadIdx.x >> 6) & 15u] = v;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		int nw = (int)((blockDim.x + 63u) >> 6);
		nw = nw > 16 ? 16 : nw;
		for (int k = 1; k < nw; ++k) v = s_blockMax[k] > v ? s_blockMax[k] : v;
		if (v > 0) atomicMaxIfAbove(addr, v);
	}
	__syncthreads();
}

// A workgroup's running sum for ONE hot key, kept in LDS across the iterations of a grid-stride loop: a single island of
// 350 000 constraints among 2 M contacts (the settled 100 000-box Tumbler) still sends one atomic per wave and iteration to
// the same word - 32 000 of them, ~6.5 ns each in L2: k_island_count 214 us for 80 MB of reads. The first key a workgroup
// meets becomes its hot key; a wave's combined sum for that key goes to LDS, everything else takes the wave path;
// blockHotFlush (after a barrier) sends the total with one atomic. Integer sums: the same bits in any order.
struct BlockHot
{
	int key; // -1: none yet
	int sum;
};

__device__ __forceinline__ void blockHotInit(BlockHot* h)
{
	if (threadIdx.x == 0) { h->key = -1; h->sum = 0; }
	__syncthreads();
}

// (wave-uniform control flow, like the waveAtomic* helpers)
__device__ __forceinline__ void blockHotAddInt(BlockHot* h, int* base, int key, int val, bool valid)
{
	const int lane = waveLane();
	const unsigned long long pm = __ballot(valid);
	if (pm == 0ull) return;
	const int leader = __ffsll((long long)pm) - 1;
	const int k0 = __shfl(key, leader);
	const bool mine = valid && key == k0;
	const int s = waveSumInt(mine ? val : 0);
	if (lane == leader)
	{
		int hk = __hip_atomic_load(&h->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (hk < 0)
		{
			int expected = -1;
			hk = __hip_atomic_compare_exchange_strong(&h->key, &expected, k0, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? k0 : expected;
		}
		if (hk == k0) atomicAdd(&h->sum, s); else atomicAdd(&base[k0], s);
	}
	waveAtomicAddInt(base, key, val, valid && !mine);
}

__device__ __forceinline__ void blockHotFlush(BlockHot* h, int* base)
{
	__syncthreads();
	if (threadIdx.x == 0 && h->key >= 0 && h->sum != 0) atomicAdd(&base[h->key], h->sum);
}

// Slots of ONE shared cursor for the valid lanes of a whole workgroup: one global atomicAdd per workgroup and call instead of
// one per wave. Every thread of the workgroup must call it (barriers inside); slots are handed out in lane order.
__device__ __forceinline__ int blockAlloc(int* cursor, bool valid)
{
	__shared__ int s_wave[16], s_base;
	const int lane = waveLane(), wv = (int)(threadIdx.x >> 6), nw = (int)((blockDim.x + 63u) >> 6);
	const unsigned long long m = __ballot(valid);
	__syncthreads(); // (a previous call's readers are done)
	if (lane == 0) s_wave[wv] = __popcll(m);
	__syncthreads();
	if (threadIdx.x == 0)
	{
		int run = 0;
		for (int q = 0; q < nw; ++q) { const int c = s_wave[q]; s_wave[q] = run; run += c; }
		s_base = run > 0 ? atomicAdd(cursor, run) : 0;
	}
	__syncthreads();
	return s_base + s_wave[wv] + __popcll(m & ((1ull << lane) - 1ull));
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

// The same for keys that come in RUNS (neighbouring lanes with the same key): the first lane of a run allocates for the run -
// one atomic per run, all in flight together, no loop over the keys. Equal keys in lanes that are not neighbours simply make
// two runs. (k_island_union: the solid contacts of a tile in contact order - contacts are created in key order, so the
// contacts of one fixture sit side by side and a body's degree was counted up one returning atomic at a time.)
__device__ __forceinline__ int waveRunAlloc(int* counter, int key, bool valid)
{
	const int lane = waveLane();
	const int prevKey = __shfl_up(key, 1);
	const bool prevValid = (bool)__shfl_up((int)valid, 1);
	const bool head = valid && (lane == 0 || !prevValid || prevKey != key);
	// (a run ends where the next head or the next invalid lane is)
	const unsigned long long breaks = __ballot(head || !valid);
	const unsigned long long above = lane == 63 ? 0ull : (breaks >> (lane + 1)) << (lane + 1);
	const int end = above ? __ffsll((long long)above) - 1 : 64; // first lane behind this lane that starts something else
	const unsigned long long heads = __ballot(head);
	const unsigned long long below = heads & ((2ull << lane) - 1ull); // heads at or below this lane
	const int myHead = below ? 63 - __clzll((long long)below) : lane;
	int base = 0;
	if (head) base = atomicAdd(&counter[key], end - lane);
	base = __shfl(base, myHead);
	return valid ? base + (lane - myHead) : 0;
}

// The same in ONE atomic round trip whatever the number of distinct keys: the lanes that hold the same key find one another
// with a ballot per key bit (as the radix sort's scatter ranks equal digits), the first lane of every group allocates for
// the group, and all those atomics are in flight together. waveKeyedAlloc pays one round trip per distinct key, one after
// the other - k_color_fill hands rows to blocks in arrival order of the island's contact list, ~60 different blocks in a
// wave of the 50 086-box pyramid: 99 us. `keyBits`: keys are below 1 << keyBits.
__device__ __forceinline__ int waveKeyedAllocOnce(int* counter, int key, bool valid, int keyBits, int stride = 1)
{
	const int lane = waveLane();
	unsigned long long peers = __ballot(valid);
	for (int b = 0; b < keyBits; ++b)
	{
		const unsigned long long m = __ballot(valid && ((key >> b) & 1));
		peers &= ((key >> b) & 1) ? m : ~m;
	}
	if (!valid) peers = 0ull;
	const int leader = peers ? __ffsll((long long)peers) - 1 : lane;
	int base = 0;
	if (valid && lane == leader) base = atomicAdd(&counter[(size_t)key * stride], __popcll(peers));
	base = __shfl(base, leader);
	return valid ? base + __popcll(peers & ((1ull << lane) - 1ull)) : 0;
}

// The same for a whole 256-lane workgroup and keys in [0, 64]: ranks through an LDS histogram, then ONE global atomicAdd
// per key that occurs, all of them in flight together (waveKeyedAlloc pays one atomic round trip per distinct key and
// wave, one after the other: seven colours on the 10k-body pyramid made k_color_fill 24 us). Every thread of the
// workgroup must call it (it contains barriers). `fetch` = false: only count (no slot returned).
// Counters indexed by a colour (colorCount, colorCursor): the first 65 - what a block partition or a settled world uses - lie
// on 65 different 128-byte lines; atomics on different WORDS of one line still queue (5 ns each against 11 on one word,
// tools/microbench/atomic_cost.hip), and k_color_check / k_color_fill send ~20 per workgroup and round to these arrays.
// Colours beyond 64 (exact-order mode: dependency levels, thousands of them) follow densely.
#define COLOR_SLOT_STRIDE 32
#define COLOR_SLOT_PADDED 65
__host__ __device__ __forceinline__ int colorSlot(int c) { return c < COLOR_SLOT_PADDED ? c * COLOR_SLOT_STRIDE : COLOR_SLOT_PADDED * COLOR_SLOT_STRIDE + (c - COLOR_SLOT_PADDED); }

__device__ __forceinline__ int blockKeyedAlloc65(int* counter, int key, bool valid, bool fetch, bool slotted = false)
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
			int* word = &counter[slotted ? colorSlot((int)threadIdx.x) : (int)threadIdx.x];
			if (fetch) s_base[threadIdx.x] = atomicAdd(word, c);
			else atomicAdd(word, c);
		}
	}
	if (!fetch) return 0;
	__syncthreads();
	return valid ? s_base[key] + local : 0;
}

#endif
