// What an atomic on ONE word costs on MI355X when many workgroups issue it ("atomics"): the measurement
// behind b2dTreeArrive. Standalone: hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_cost tools/microbench/atomic_cost.hip
// Every kernel streams over the same array (so that there is a kernel to speak of) and then, per workgroup:
//   mode 0: nothing          mode 1: one non-returning atomicAdd on one word      mode 2: the same, returning (the value is used)
//   mode 3: two-level arrival (slot b % 32, the completing workgroup goes on to the root: b2d_world.h)
//   mode 4: one atomicAdd per WAVE on one word
//   mode 5: one add per workgroup on word b % 32 of ONE 128-byte line     mode 6: on word b % 32 of 32 different lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_probe(const float4* in, float* out, int n, int mode, int* word, unsigned long long* tree)
{
	float acc = 0.0f;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const float4 v = in[i];
		acc += v.x + v.y + v.z + v.w;
	}
	if (acc == 12345.678f) out[0] = acc; // (keeps the loads alive)
	if (mode == 4) { if ((threadIdx.x & 63) == 0) atomicAdd(word, 1); return; }
	__syncthreads();
	if (threadIdx.x != 0) return;
	if (mode == 5) { atomicAdd(word + 64 + (blockIdx.x & 31), 1); return; }
	if (mode == 6) { atomicAdd(word + 128 + 32 * (blockIdx.x & 31), 1); return; }
	if (mode == 1) atomicAdd(word, 1);
	else if (mode == 2) { const int k = atomicAdd(word, 1); if (k == 0x7fffffff) out[1] = 1.0f; }
	else if (mode == 3)
	{
		const unsigned nb = gridDim.x, groups = nb < 32u ? nb : 32u, g = blockIdx.x % groups;
		const unsigned members = nb / groups + (g < nb % groups ? 1u : 0u);
		unsigned long long* slot = tree + (size_t)g * 16;
		unsigned long long seen = __hip_atomic_fetch_add(slot, (1ull << 48) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + ((1ull << 48) | 1ull);
		if ((unsigned)(seen >> 48) != members) return;
		__hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		unsigned long long* root = tree + 32 * 16;
		const unsigned long long add2 = (1ull << 48) | (seen & 0xffffffffffffull);
		seen = __hip_atomic_fetch_add(root, add2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add2;
		if ((unsigned)(seen >> 48) != groups) return;
		__hip_atomic_store(root, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		atomicAdd(word, (int)(seen & 0xffffffu));
	}
}

int main()
{
	const int n = 4 << 20; // 64 MB of float4: ~20 us of streaming
	float4* in; float* out; int* word; unsigned long long* tree;
	CHECK(hipMalloc(&in, (size_t)n * sizeof(float4)));
	CHECK(hipMalloc(&out, 64));
	CHECK(hipMalloc(&word, 8192));
	CHECK(hipMalloc(&tree, 33 * 16 * 8));
	CHECK(hipMemset(in, 0, (size_t)n * sizeof(float4)));
	CHECK(hipMemset(word, 0, 8192));
	CHECK(hipMemset(tree, 0, 33 * 16 * 8));
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	const char* names[7] = { "no atomic", "1 add / workgroup", "1 returning add / workgroup", "two-level arrival", "1 add / wave", "32 words of one line", "32 words on 32 lines" };
	printf("%-10s", "workgroups");
	for (int m = 0; m < 7; ++m) printf(" %28s", names[m]);
	printf("   (us per launch, mean of 50; 64 MB streamed per launch)\n");
	for (int grid : { 256, 512, 1024, 2048, 4096, 8192, 16384 })
	{
		printf("%-10d", grid);
		for (int mode = 0; mode < 7; ++mode)
		{
			for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k_probe, dim3(grid), dim3(256), 0, 0, in, out, n, mode, word, tree);
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(a, 0));
			for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(k_probe, dim3(grid), dim3(256), 0, 0, in, out, n, mode, word, tree);
			CHECK(hipEventRecord(b, 0));
			CHECK(hipEventSynchronize(b));
			float ms = 0.0f;
			CHECK(hipEventElapsedTime(&ms, a, b));
			printf(" %28.1f", ms * 1000.0f / 50.0f);
		}
		printf("\n");
	}
	int total = 0;
	CHECK(hipMemcpy(&total, word, 4, hipMemcpyDeviceToHost));
	printf("(word = %d)\n", total);
	return 0;
}
