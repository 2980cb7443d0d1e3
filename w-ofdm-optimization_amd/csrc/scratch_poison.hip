// Test helper (libwofdm_poison.so, not part of the product): fill the queue's scratch (private-segment)
// memory with a NaN-like pattern.  A kernel whose per-lane array is forced into scratch writes the
// pattern over every wave slot; a frame kernel that reloads a register-spill slot it has not written in
// the SAME launch (a spill under a partial exec mask reloaded under a fuller one -- DESIGN.md section 4)
// then reads 0x7FC0DEAD instead of whatever an earlier launch happened to leave there, and its error
// counters go wrong on EVERY launch instead of on the first one in a fresh process.
// tests/test_gpu_parity.py::test_every_spilling_production_kernel poisons before each run;
// scratch_peek is the tool's own check (an unwritten scratch array must read back as the pattern).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(1024) poison_kernel(uint32_t *sink, uint32_t pat, int n)
{
    volatile uint32_t a[96];
    for (int i = 0; i < 96; ++i) a[i] = pat + (n > 1000000 ? i : 0);
    uint32_t s = 0;
    for (int i = 0; i < 96; ++i) s += a[(i * 7 + n) % 96];
    if (s == 12345u) sink[threadIdx.x] = s;
}

extern "C" int scratch_poison(uint32_t pat)
{
    uint32_t *sink = nullptr;
    if (hipMalloc(&sink, 4096) != hipSuccess) return -1;
    // enough workgroups to occupy every wave slot of every CU several times over
    hipLaunchKernelGGL(poison_kernel, dim3(256 * 8), dim3(1024), 0, nullptr, sink, pat, 3);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    (void)hipFree(sink);
    return rc;
}

// self-check of the tool: a kernel that reads scratch it has not written
__global__ void __launch_bounds__(1024) peek_kernel(uint32_t *out, int n)
{
    volatile uint32_t a[16];
    if (n > 1000000) for (int i = 0; i < 16; ++i) a[i] = (uint32_t)i;      // (never taken: keeps the array in scratch)
    out[blockIdx.x * 1024 + threadIdx.x] = a[(threadIdx.x + n) % 16];
}
extern "C" int scratch_peek(uint32_t *host_out, int blocks)
{
    uint32_t *d = nullptr;
    if (hipMalloc(&d, (size_t)blocks * 1024 * 4) != hipSuccess) return -1;
    hipLaunchKernelGGL(peek_kernel, dim3(blocks), dim3(1024), 0, nullptr, d, 3);
    int rc = hipMemcpy(host_out, d, (size_t)blocks * 1024 * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
    (void)hipFree(d);
    return rc;
}
