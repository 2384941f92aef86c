// b2d_scan.h - device-wide exclusive scan and stable LSD radix sort, hand-written for wave64.
// Element counts live in device memory (the host never learns them mid-step), so every launch is
// sized by a capacity and blocks beyond the live count exit immediately.
#ifndef B2D_SCAN_H
#define B2D_SCAN_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SCAN_THREADS 256
#define SCAN_ITEMS 4
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)
#define SCAN_CHAIN_MAX_TILES 256 // single-pass scan up to 256 tiles: 256 K elements in tiles of 1 024, 1 M in tiles of 4 096; three kernels beyond
#define SCAN_ITEMS_WIDE 16
#define SCAN_TILE_WIDE (SCAN_THREADS * SCAN_ITEMS_WIDE)

__device__ __forceinline__ int scanAdd(int a, int b) { return a + b; }
__device__ __forceinline__ int4 scanAdd(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ void scanZero(int& a) { a = 0; }
__device__ __forceinline__ void scanZero(int4& a) { a = make_int4(0, 0, 0, 0); }
// Tile aggregates / prefixes handed from workgroup to workgroup of one launch: stores and loads that go past the XCD's L2
// (sc1), so the status words need no release / acquire (an agent-scope release writes the whole L2 back, an acquire
// invalidates it - per tile, and per spin of the look-back).
__device__ __forceinline__ void scanStoreAgent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int scanLoadAgent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void scanStoreAgent(int4* p, int4 v)
{
	typedef int i4 __attribute__((ext_vector_type(4)));
	i4 q;
	q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
	// (s_nop 1: the store's data registers must not be written by the next two instructions - see stRow in b2d_handover.h)
	asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ int4 scanLoadAgent(const int4* p)
{
	typedef int i4 __attribute__((ext_vector_type(4)));
	i4 r;
	asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
	return make_int4(r.x, r.y, r.z, r.w);
}

// Exclusive scan of one value per thread across a 256-thread block (Hillis-Steele through LDS).
template <typename T>
__device__ __forceinline__ T blockExclusiveScan(T val, T* total, T* lds /* [2 * SCAN_THREADS] */)
{
	int tid = threadIdx.x;
	int pin = 0;
	lds[tid] = val;
	__syncthreads();
	for (int off = 1; off < SCAN_THREADS; off <<= 1)
	{
		T v = lds[pin * SCAN_THREADS + tid];
		if (tid >= off) v = scanAdd(v, lds[pin * SCAN_THREADS + tid - off]);
		lds[(1 - pin) * SCAN_THREADS + tid] = v;
		pin = 1 - pin;
		__syncthreads();
	}
	T incl = lds[pin * SCAN_THREADS + tid];
	*total = lds[pin * SCAN_THREADS + SCAN_THREADS - 1];
	T excl;
	scanZero(excl);
	if (tid > 0) excl = lds[pin * SCAN_THREADS + tid - 1];
	__syncthreads();
	(void)incl;
	return excl;
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const T* __restrict__ in, T* __restrict__ blockSums, const int* nPtr)
{
	__shared__ T lds[2 * SCAN_THREADS];
	int n = *nPtr;
	int base = blockIdx.x * SCAN_TILE;
	if (base >= n) return;
	T sum;
	scanZero(sum);
	for (int k = 0; k < SCAN_ITEMS; ++k)
	{
		int i = base + threadIdx.x * SCAN_ITEMS + k;
		if (i < n) sum = scanAdd(sum, in[i]);
	}
	T total;
	blockExclusiveScan(sum, &total, lds);
	if (threadIdx.x == 0) blockSums[blockIdx.x] = total;
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_blocksums(T* __restrict__ blockSums, const int* nPtr, T* totalOut)
{
	__shared__ T lds[2 * SCAN_THREADS];
	int n = *nPtr;
	int numBlocks = (n + SCAN_TILE - 1) / SCAN_TILE;
	T carry;
	scanZero(carry);
	for (int base = 0; base < numBlocks; base += SCAN_THREADS)
	{
		int i = base + threadIdx.x;
		T v;
		scanZero(v);
		if (i < numBlocks) v = blockSums[i];
		T total;
		T excl = blockExclusiveScan(v, &total, lds);
		if (i < numBlocks) blockSums[i] = scanAdd(carry, excl);
		carry = scanAdd(carry, total);
	}
	if (threadIdx.x == 0 && totalOut) *totalOut = carry;
}

// out[i] = sum of in[0..i) ; out[n] = total (out must hold n + 1 entries)
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_final(const T* __restrict__ in, T* __restrict__ out, const T* __restrict__ blockSums, const int* nPtr)
{
	__shared__ T lds[2 * SCAN_THREADS];
	int n = *nPtr;
	int base = blockIdx.x * SCAN_TILE;
	if (n == 0)
	{
		if (blockIdx.x == 0 && threadIdx.x == 0)
		{
			T z;
			scanZero(z);
			out[0] = z;
		}
		return;
	}
	if (base >= n) return;
	T v[SCAN_ITEMS];
	T sum;
	scanZero(sum);
	for (int k = 0; k < SCAN_ITEMS; ++k)
	{
		int i = base + threadIdx.x * SCAN_ITEMS + k;
		scanZero(v[k]);
		if (i < n) v[k] = in[i];
		sum = scanAdd(sum, v[k]);
	}
	T total;
	T excl = blockExclusiveScan(sum, &total, lds);
	T run = scanAdd(blockSums[blockIdx.x], excl);
	for (int k = 0; k < SCAN_ITEMS; ++k)
	{
		int i = base + threadIdx.x * SCAN_ITEMS + k;
		if (i < n) out[i] = run;
		run = scanAdd(run, v[k]);
		if (i == n - 1) out[n] = run;
	}
}

// ---- single-pass scan (decoupled look-back) ---------------------------------------------------------------------------------
// One launch instead of three: every tile publishes its aggregate, then walks back over its predecessors' status words
// until it meets one whose inclusive prefix is known, and publishes its own. Status per tile in `work`:
//   work[t]     aggregate of tile t          work[cap + t]   inclusive prefix up to and including tile t
//   flags[t]    (epoch << 2) | state, state 1 = aggregate published, 2 = prefix published
// `epoch` is unique per launch (the host counts them), so the flag words are never reset (they live in an array of their
// own that only ever holds flag words: zeroed when allocated). Progress: workgroups are dispatched in
// index order on every XCD, so the lowest unfinished tile is always running.
__device__ __forceinline__ int scanWaveSum(int v)
{
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
	return v;
}
__device__ __forceinline__ int4 scanWaveSum(int4 v)
{
	return make_int4(scanWaveSum(v.x), scanWaveSum(v.y), scanWaveSum(v.z), scanWaveSum(v.w));
}

// The look-back spins on status words other workgroups publish: bounded. A tile that has waited SCAN_SPIN_MAX polls (seconds:
// a poll is a memory round trip, ~1 us) for one word gives up, raises SCAN_ABORT_BIT in *abortWord and publishes a zero
// prefix so that its successors end too; every other waiting lane leaves as soon as it sees the bit. The host finds the bit
// with the step's read-back and fails the step (B2HIP_ERR_HIP) instead of hanging in it.
#define SCAN_SPIN_MAX (1u << 23)
#define SCAN_ABORT_BIT 256

// ITEMS per thread: 4 (tiles of 1 024 elements) up to 256 K elements; 16 (tiles of 4 096) up to 1 M - the look-back is a chain
// over the TILES, so a million-element scan (the body and grid tables of BASELINE config 5, the tables of a rank of a sharded
// world) is as short a chain as a 256 K one and still one launch instead of three.
template <typename T, int ITEMS>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chain(const T* __restrict__ in, T* __restrict__ out, T* work, int* flags, int cap, const int* nPtr, unsigned epoch,
	int* abortWord)
{
	constexpr int TILE = SCAN_THREADS * ITEMS;
	__shared__ T lds[2 * SCAN_THREADS];
	__shared__ T s_prefix;
	const int n = *nPtr;
	const int tile = blockIdx.x;
	const int base = tile * TILE;
	if (n == 0)
	{
		if (tile == 0 && threadIdx.x == 0)
		{
			T z;
			scanZero(z);
			out[0] = z;
		}
		return;
	}
	if (base >= n) return;
	T v[ITEMS];
	T sum;
	scanZero(sum);
	for (int k = 0; k < ITEMS; ++k)
	{
		const int i = base + threadIdx.x * ITEMS + k;
		scanZero(v[k]);
		if (i < n) v[k] = in[i];
		sum = scanAdd(sum, v[k]);
	}
	T total;
	const T excl = blockExclusiveScan(sum, &total, lds);
	if (threadIdx.x < 64)
	{
		// the first wave looks back over 64 predecessors at a time (one status word per lane)
		const int lane = threadIdx.x;
		T* agg = work;
		T* pre = work + cap;
		T running;
		scanZero(running);
		if (tile > 0)
		{
			if (lane == 0)
			{
				scanStoreAgent(&agg[tile], total);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__hip_atomic_store(flags + tile, (int)((epoch << 2) | 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			int hi = tile - 1; // nearest predecessor not yet accounted for
			while (true)
			{
				const int t = hi - lane;
				unsigned state = 2u; // (lanes before tile 0 count as "prefix known, zero")
				T val;
				scanZero(val);
				if (t >= 0)
				{
					unsigned word;
					unsigned polls = 0;
					bool gaveUp = false;
					do
					{
						word = (unsigned)__hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						if ((++polls & 0xfffu) == 0u)
						{
							if (polls >= SCAN_SPIN_MAX) atomicOr(abortWord, SCAN_ABORT_BIT);
							gaveUp = (__hip_atomic_load(abortWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & SCAN_ABORT_BIT) != 0;
						}
					} while (!gaveUp && ((word >> 2) != epoch || (word & 3u) == 0u));
					state = gaveUp ? 2u : (word & 3u); // (aborted: pretend a zero prefix so that the walk ends)
					if (!gaveUp) val = scanLoadAgent(state == 2u ? &pre[t] : &agg[t]);
				}
				// the nearest tile whose inclusive prefix is known ends the walk
				const unsigned long long known = __ballot(state == 2u);
				const int stop = known ? __ffsll((long long)known) - 1 : 64;
				if (lane > stop) scanZero(val);
				running = scanAdd(running, scanWaveSum(val));
				if (stop < 64) break;
				hi -= 64;
			}
		}
		if (lane == 0)
		{
			scanStoreAgent(&pre[tile], scanAdd(running, total));
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__hip_atomic_store(flags + tile, (int)((epoch << 2) | 2u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			s_prefix = running;
		}
	}
	__syncthreads();
	T run = scanAdd(s_prefix, excl);
	for (int k = 0; k < ITEMS; ++k)
	{
		const int i = base + threadIdx.x * ITEMS + k;
		if (i < n) out[i] = run;
		run = scanAdd(run, v[k]);
		if (i == n - 1) out[n] = run;
	}
}

// Host helper: ONE launch on `stream`. capN bounds the grid; nPtr is the live count in device memory; `work` holds
// 2 * (capN / SCAN_TILE + 4) elements, `flags` capN / SCAN_TILE + 4 ints (zeroed when allocated).
// The epoch belongs to the flag array (ScanFlags: one per world): it tags the array's status words, 30 bits wide, and when it
// wraps the array is zeroed on the stream before the next scan, so that no word left behind by a scan 2^30 launches ago can
// read as "published" (a process-wide counter over per-world arrays could alias in a rarely used tile).
struct ScanFlags
{
	int* words = nullptr;   // device
	size_t count = 0;
	unsigned epoch = 0;     // of the last launch over `words`
	int* abortWord = nullptr; // device: SCAN_ABORT_BIT is raised here when a look-back gives up
};
template <typename T>
static inline void deviceExclusiveScan(hipStream_t stream, const T* in, T* out, T* work, ScanFlags& sf, const int* nPtr, int capN)
{
	int* flags = sf.words;
	int blocks = (capN + SCAN_TILE - 1) / SCAN_TILE;
	if (blocks < 1) blocks = 1;
	const int blocksWide = (capN + SCAN_TILE_WIDE - 1) / SCAN_TILE_WIDE;
	hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
	(void)hipStreamIsCapturing(stream, &capturing);
	if (capturing != hipStreamCaptureStatusNone || blocksWide > SCAN_CHAIN_MAX_TILES)
	{
		// a captured launch would replay its epoch, and past a few hundred tiles the look-back chain costs more than the two
		// launches it saves (1.4 M elements: 25 us against 15): the three-kernel form (reduce, scan of the tile sums, final)
		hipLaunchKernelGGL(k_scan_reduce<T>, dim3(blocks), dim3(SCAN_THREADS), 0, stream, in, work, nPtr);
		hipLaunchKernelGGL(k_scan_blocksums<T>, dim3(1), dim3(SCAN_THREADS), 0, stream, work, nPtr, (T*)nullptr);
		hipLaunchKernelGGL(k_scan_final<T>, dim3(blocks), dim3(SCAN_THREADS), 0, stream, in, out, work, nPtr);
		return;
	}
	if (sf.epoch >= 0x3fffffffu)
	{
		(void)hipMemsetAsync(flags, 0, sf.count * sizeof(int), stream);
		sf.epoch = 0u;
	}
	const unsigned epoch = ++sf.epoch;
	if (blocks <= SCAN_CHAIN_MAX_TILES)
		hipLaunchKernelGGL((k_scan_chain<T, SCAN_ITEMS>), dim3(blocks), dim3(SCAN_THREADS), 0, stream, in, out, work, flags, blocks + 4, nPtr, epoch, sf.abortWord);
	else
		hipLaunchKernelGGL((k_scan_chain<T, SCAN_ITEMS_WIDE>), dim3(blocksWide), dim3(SCAN_THREADS), 0, stream, in, out, work, flags, blocksWide + 4, nPtr, epoch, sf.abortWord);
}

// ---------------------------------------------------------------------------------------------
// Stable LSD radix sort of (uint64 key, int2 payload), up to RADIX_BITS = 11 bits per pass (round 5; 8 before: the pair
// update's keys are two proxy keys of 17 - 21 bits - six passes of three launches each, every one a dependent launch of a
// step that is priced by those; now four).
// Per pass: (1) per-tile digit histogram, (2) scan of the digit-major histogram matrix,
// (3) stable scatter using wave64 ballots to rank equal digits inside a wave.
#define RADIX_THREADS 256
#define RADIX_ITEMS 8
#define RADIX_TILE (RADIX_THREADS * RADIX_ITEMS)
#define RADIX_BITS 11
#define RADIX_DIGITS (1 << RADIX_BITS)

// `width` <= RADIX_BITS: the bits this pass looks at (the digit is (key >> shift) & ((1 << width) - 1); all RADIX_DIGITS rows of
// the matrix are written, the unused ones with zeros)
__global__ __launch_bounds__(RADIX_THREADS) void k_radix_hist(const uint64_t* __restrict__ keys, int* __restrict__ hist,
	const int* nPtr, int minN, int shift, int width, int numTilesCap)
{
	__shared__ int lh[RADIX_DIGITS];
	int n = *nPtr;
	if (n <= minN) return;
	int numTiles = (n + RADIX_TILE - 1) / RADIX_TILE;
	if (numTiles > (int)gridDim.x) numTiles = (int)gridDim.x; // (a sort sized for fewer keys than there are: k_radix_count has flagged it)
	int tile = blockIdx.x;
	if (tile >= numTiles) return;
	for (int q = threadIdx.x; q < RADIX_DIGITS; q += RADIX_THREADS) lh[q] = 0;
	__syncthreads();
	const uint32_t mask = (1u << width) - 1u;
	int base = tile * RADIX_TILE;
	for (int k = 0; k < RADIX_ITEMS; ++k)
	{
		int i = base + k * RADIX_THREADS + threadIdx.x;
		if (i < n)
		{
			int d = (int)((uint32_t)(keys[i] >> shift) & mask);
			atomicAdd(&lh[d], 1);
		}
	}
	__syncthreads();
	// digit-major so that a plain scan yields global offsets
	for (int q = threadIdx.x; q < RADIX_DIGITS; q += RADIX_THREADS) hist[q * numTiles + tile] = lh[q];
	(void)numTilesCap;
}

// histCount = RADIX_DIGITS * numTiles, written to device memory for the scan utility
// tilesCap: the tiles the launches of this sort were sized for. A sort that was queued without the host having seen the count
// (the pair update of a world that has been sorting with the radix passes lately) may meet more: the passes then cover a part
// only and `overflow` gets `overflowBit` - whoever consumes the result looks at it (createBlocked).
__global__ void k_radix_count(const int* nPtr, int minN, int* histCount, int tilesCap, int* overflow, int overflowBit)
{
	int n = *nPtr;
	int numTiles = (n + RADIX_TILE - 1) / RADIX_TILE;
	if (numTiles > tilesCap)
	{
		if (overflow != nullptr) atomicOr(overflow, overflowBit);
		numTiles = tilesCap;
	}
	*histCount = (n <= minN) ? 0 : RADIX_DIGITS * numTiles;
}

__global__ __launch_bounds__(RADIX_THREADS) void k_radix_scatter(const uint64_t* __restrict__ keysIn, const int2* __restrict__ valsIn,
	uint64_t* __restrict__ keysOut, int2* __restrict__ valsOut, const int* __restrict__ histScan, const int* nPtr, int minN, int shift, int width)
{
	// per wave: how many of the current round's keys of the three waves before it carry a digit - only the digits that OCCUR in
	// the round are touched (a round is 256 keys: at most 256 of the 2 048 counters), set by the first lane of a digit group
	// and taken back by it after the round
	__shared__ int waveCount[4][RADIX_DIGITS];
	__shared__ int digitBase[RADIX_DIGITS]; // running global offset per digit for this tile
	int n = *nPtr;
	if (n <= minN) return;
	int numTiles = (n + RADIX_TILE - 1) / RADIX_TILE;
	if (numTiles > (int)gridDim.x) { numTiles = (int)gridDim.x; n = numTiles * RADIX_TILE; } // (see k_radix_hist)
	int tile = blockIdx.x;
	if (tile >= numTiles) return;
	int tid = threadIdx.x;
	int lane = tid & 63;
	int wave = tid >> 6;
	for (int q = tid; q < RADIX_DIGITS; q += RADIX_THREADS)
	{
		digitBase[q] = histScan[q * numTiles + tile];
		waveCount[0][q] = 0; waveCount[1][q] = 0; waveCount[2][q] = 0; waveCount[3][q] = 0;
	}
	__syncthreads();
	const uint32_t mask = (1u << width) - 1u;
	int base = tile * RADIX_TILE;
	// (all of the tile's keys and payloads first: eight independent loads in flight, not one load's latency in front of every
	// round - the kernel moves 2 MB and was 16 us of dependent steps)
	uint64_t keyR[RADIX_ITEMS];
	int2 valR[RADIX_ITEMS];
#pragma unroll
	for (int k = 0; k < RADIX_ITEMS; ++k)
	{
		const int i = base + k * RADIX_THREADS + tid;
		keyR[k] = i < n ? keysIn[i] : 0;
		valR[k] = i < n ? valsIn[i] : make_int2(0, 0);
	}
#pragma unroll
	for (int k = 0; k < RADIX_ITEMS; ++k)
	{
		int i = base + k * RADIX_THREADS + tid;
		bool valid = i < n;
		uint64_t key = keyR[k];
		int d = valid ? (int)((uint32_t)(key >> shift) & mask) : -1;
		// lanes of this wave holding the same digit
		unsigned long long peers = __ballot(valid);
		for (int b = 0; b < RADIX_BITS; ++b)
		{
			unsigned long long m = __ballot(valid && ((d >> b) & 1));
			peers &= ((d >> b) & 1) ? m : ~m;
		}
		if (!valid) peers = 0;
		unsigned long long lower = peers & ((1ull << lane) - 1ull);
		int rankInWave = __popcll(lower);
		const bool first = valid && lower == 0; // first lane of each digit group
		const int groupSize = __popcll(peers);
		if (first) waveCount[wave][d] = groupSize;
		__syncthreads();
		int total = 0;
		if (valid)
		{
			int off = digitBase[d];
			for (int w = 0; w < wave; ++w) off += waveCount[w][d];
			int dst = off + rankInWave;
			keysOut[dst] = key;
			valsOut[dst] = valR[k];
			if (first) total = waveCount[0][d] + waveCount[1][d] + waveCount[2][d] + waveCount[3][d];
		}
		__syncthreads();
		// the digit's running offset moves on by the round's keys with that digit: added once per digit - by the first lane of
		// the digit's group in the LOWEST wave that has the digit - and the round's counters go back to zero
		if (first)
		{
			bool lowest = true;
			for (int w = 0; w < wave; ++w) lowest = lowest && waveCount[w][d] == 0;
			if (lowest) digitBase[d] += total;
		}
		__syncthreads();
		if (first) waveCount[wave][d] = 0;
		__syncthreads();
	}
}

#endif
