// Test helper (libwofdm_poison.so, not part of the product): fill what survives on chip from one kernel to the
// next -- the queue's scratch (private-segment) memory, every CU's LDS, every SIMD's vector registers -- with a
// chosen pattern, so that a frame kernel that reads any of it before writing it goes wrong on EVERY launch
// instead of on the first one after a different kernel.
//   scratch_poison / scratch_peek   a per-lane array forced into scratch writes the pattern over every wave slot
//                                   (a spill under a partial exec mask reloaded under a fuller one -- DESIGN.md
//                                   section 4 -- then reads 0x7FC0DEAD); peek = the tool's own check
//   lds_poison / lds_peek / lds_poison_range
//   reg_poison
// tests/test_gpu_parity.py (test_every_spilling_production_kernel, test_no_kernel_reads_what_an_earlier_kernel_
// left_on_chip) and tools/poison_probe.py use them.
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

// The same for the LDS: a workgroup's LDS allocation is not cleared at launch, so a frame kernel that reads
// an LDS word before writing it sees what the previous workgroup on that CU left there -- after an earlier
// launch of the SAME plan that is a plausible (often the correct) value.  lds_poison fills all 160 KB of
// every CU's LDS with `pat`; lds_peek counts, per workgroup, the words of a fresh allocation that still
// hold it (the tool's own check).
__global__ void __launch_bounds__(256) lds_poison_kernel(uint32_t pat, int words, uint32_t *sink)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = pat;
    __syncthreads();
    // (a read-back the compiler cannot drop; long enough that the grid spreads over all CUs)
    uint32_t s = 0;
    for (int r = 0; r < 4; ++r)
        for (int i = threadIdx.x; i < words; i += 256) s += lds[i] ^ (uint32_t)r;
    if (s == 0x12345u) sink[0] = s;
}
__global__ void __launch_bounds__(256) lds_peek_kernel(uint32_t pat, int words, uint32_t *hits)
{
    extern __shared__ uint32_t lds[];
    uint32_t n = 0;
    for (int i = threadIdx.x; i < words; i += 256) n += lds[i] == pat;
    atomicAdd(&hits[blockIdx.x], n);
}
static const int LDS_BYTES = 160 * 1024;
extern "C" int lds_poison(uint32_t pat)
{
    uint32_t *sink = nullptr;
    if (hipMalloc(&sink, 64) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(lds_poison_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -3;
    hipLaunchKernelGGL(lds_poison_kernel, dim3(256 * 8), dim3(256), LDS_BYTES, nullptr, pat, LDS_BYTES / 4, sink);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    (void)hipFree(sink);
    return rc;
}
extern "C" int lds_peek(uint32_t pat, uint32_t *host_hits, int blocks)
{
    uint32_t *d = nullptr;
    if (hipMalloc(&d, (size_t)blocks * 4) != hipSuccess || hipMemset(d, 0, (size_t)blocks * 4) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(lds_peek_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -3;
    hipLaunchKernelGGL(lds_peek_kernel, dim3(blocks), dim3(256), LDS_BYTES, nullptr, pat, LDS_BYTES / 4, d);
    int rc = hipMemcpy(host_hits, d, (size_t)blocks * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
    (void)hipFree(d);
    return rc;
}

// Registers: a wave's VGPRs, AGPRs and SGPRs hold what the previous wave on that SIMD left in them.  A kernel
// that reads a register before writing it (an exec-masked definition read under a fuller mask, an MFMA
// accumulator never cleared) then depends on the kernel that ran before it.  reg_poison fills the vector
// registers of every SIMD (v0-v251, a0-a255) and s36-s99 with `pat`.
__global__ void __launch_bounds__(256) reg_poison_kernel(uint32_t pat, uint32_t *sink)
{
    uint32_t v;
    asm volatile(
        "v_mov_b32 v0, %1\n\t"
        "v_mov_b32 v1, %1\n\t"
        "v_mov_b32 v2, %1\n\t"
        "v_mov_b32 v3, %1\n\t"
        "v_mov_b32 v4, %1\n\t"
        "v_mov_b32 v5, %1\n\t"
        "v_mov_b32 v6, %1\n\t"
        "v_mov_b32 v7, %1\n\t"
        "v_mov_b32 v8, %1\n\t"
        "v_mov_b32 v9, %1\n\t"
        "v_mov_b32 v10, %1\n\t"
        "v_mov_b32 v11, %1\n\t"
        "v_mov_b32 v12, %1\n\t"
        "v_mov_b32 v13, %1\n\t"
        "v_mov_b32 v14, %1\n\t"
        "v_mov_b32 v15, %1\n\t"
        "v_mov_b32 v16, %1\n\t"
        "v_mov_b32 v17, %1\n\t"
        "v_mov_b32 v18, %1\n\t"
        "v_mov_b32 v19, %1\n\t"
        "v_mov_b32 v20, %1\n\t"
        "v_mov_b32 v21, %1\n\t"
        "v_mov_b32 v22, %1\n\t"
        "v_mov_b32 v23, %1\n\t"
        "v_mov_b32 v24, %1\n\t"
        "v_mov_b32 v25, %1\n\t"
        "v_mov_b32 v26, %1\n\t"
        "v_mov_b32 v27, %1\n\t"
        "v_mov_b32 v28, %1\n\t"
        "v_mov_b32 v29, %1\n\t"
        "v_mov_b32 v30, %1\n\t"
        "v_mov_b32 v31, %1\n\t"
        "v_mov_b32 v32, %1\n\t"
        "v_mov_b32 v33, %1\n\t"
        "v_mov_b32 v34, %1\n\t"
        "v_mov_b32 v35, %1\n\t"
        "v_mov_b32 v36, %1\n\t"
        "v_mov_b32 v37, %1\n\t"
        "v_mov_b32 v38, %1\n\t"
        "v_mov_b32 v39, %1\n\t"
        "v_mov_b32 v40, %1\n\t"
        "v_mov_b32 v41, %1\n\t"
        "v_mov_b32 v42, %1\n\t"
        "v_mov_b32 v43, %1\n\t"
        "v_mov_b32 v44, %1\n\t"
        "v_mov_b32 v45, %1\n\t"
        "v_mov_b32 v46, %1\n\t"
        "v_mov_b32 v47, %1\n\t"
        "v_mov_b32 v48, %1\n\t"
        "v_mov_b32 v49, %1\n\t"
        "v_mov_b32 v50, %1\n\t"
        "v_mov_b32 v51, %1\n\t"
        "v_mov_b32 v52, %1\n\t"
        "v_mov_b32 v53, %1\n\t"
        "v_mov_b32 v54, %1\n\t"
        "v_mov_b32 v55, %1\n\t"
        "v_mov_b32 v56, %1\n\t"
        "v_mov_b32 v57, %1\n\t"
        "v_mov_b32 v58, %1\n\t"
        "v_mov_b32 v59, %1\n\t"
        "v_mov_b32 v60, %1\n\t"
        "v_mov_b32 v61, %1\n\t"
        "v_mov_b32 v62, %1\n\t"
        "v_mov_b32 v63, %1\n\t"
        "v_mov_b32 v64, %1\n\t"
        "v_mov_b32 v65, %1\n\t"
        "v_mov_b32 v66, %1\n\t"
        "v_mov_b32 v67, %1\n\t"
        "v_mov_b32 v68, %1\n\t"
        "v_mov_b32 v69, %1\n\t"
        "v_mov_b32 v70, %1\n\t"
        "v_mov_b32 v71, %1\n\t"
        "v_mov_b32 v72, %1\n\t"
        "v_mov_b32 v73, %1\n\t"
        "v_mov_b32 v74, %1\n\t"
        "v_mov_b32 v75, %1\n\t"
        "v_mov_b32 v76, %1\n\t"
        "v_mov_b32 v77, %1\n\t"
        "v_mov_b32 v78, %1\n\t"
        "v_mov_b32 v79, %1\n\t"
        "v_mov_b32 v80, %1\n\t"
        "v_mov_b32 v81, %1\n\t"
        "v_mov_b32 v82, %1\n\t"
        "v_mov_b32 v83, %1\n\t"
        "v_mov_b32 v84, %1\n\t"
        "v_mov_b32 v85, %1\n\t"
        "v_mov_b32 v86, %1\n\t"
        "v_mov_b32 v87, %1\n\t"
        "v_mov_b32 v88, %1\n\t"
        "v_mov_b32 v89, %1\n\t"
        "v_mov_b32 v90, %1\n\t"
        "v_mov_b32 v91, %1\n\t"
        "v_mov_b32 v92, %1\n\t"
        "v_mov_b32 v93, %1\n\t"
        "v_mov_b32 v94, %1\n\t"
        "v_mov_b32 v95, %1\n\t"
        "v_mov_b32 v96, %1\n\t"
        "v_mov_b32 v97, %1\n\t"
        "v_mov_b32 v98, %1\n\t"
        "v_mov_b32 v99, %1\n\t"
        "v_mov_b32 v100, %1\n\t"
        "v_mov_b32 v101, %1\n\t"
        "v_mov_b32 v102, %1\n\t"
        "v_mov_b32 v103, %1\n\t"
        "v_mov_b32 v104, %1\n\t"
        "v_mov_b32 v105, %1\n\t"
        "v_mov_b32 v106, %1\n\t"
        "v_mov_b32 v107, %1\n\t"
        "v_mov_b32 v108, %1\n\t"
        "v_mov_b32 v109, %1\n\t"
        "v_mov_b32 v110, %1\n\t"
        "v_mov_b32 v111, %1\n\t"
        "v_mov_b32 v112, %1\n\t"
        "v_mov_b32 v113, %1\n\t"
        "v_mov_b32 v114, %1\n\t"
        "v_mov_b32 v115, %1\n\t"
        "v_mov_b32 v116, %1\n\t"
        "v_mov_b32 v117, %1\n\t"
        "v_mov_b32 v118, %1\n\t"
        "v_mov_b32 v119, %1\n\t"
        "v_mov_b32 v120, %1\n\t"
        "v_mov_b32 v121, %1\n\t"
        "v_mov_b32 v122, %1\n\t"
        "v_mov_b32 v123, %1\n\t"
        "v_mov_b32 v124, %1\n\t"
        "v_mov_b32 v125, %1\n\t"
        "v_mov_b32 v126, %1\n\t"
        "v_mov_b32 v127, %1\n\t"
        "v_mov_b32 v128, %1\n\t"
        "v_mov_b32 v129, %1\n\t"
        "v_mov_b32 v130, %1\n\t"
        "v_mov_b32 v131, %1\n\t"
        "v_mov_b32 v132, %1\n\t"
        "v_mov_b32 v133, %1\n\t"
        "v_mov_b32 v134, %1\n\t"
        "v_mov_b32 v135, %1\n\t"
        "v_mov_b32 v136, %1\n\t"
        "v_mov_b32 v137, %1\n\t"
        "v_mov_b32 v138, %1\n\t"
        "v_mov_b32 v139, %1\n\t"
        "v_mov_b32 v140, %1\n\t"
        "v_mov_b32 v141, %1\n\t"
        "v_mov_b32 v142, %1\n\t"
        "v_mov_b32 v143, %1\n\t"
        "v_mov_b32 v144, %1\n\t"
        "v_mov_b32 v145, %1\n\t"
        "v_mov_b32 v146, %1\n\t"
        "v_mov_b32 v147, %1\n\t"
        "v_mov_b32 v148, %1\n\t"
        "v_mov_b32 v149, %1\n\t"
        "v_mov_b32 v150, %1\n\t"
        "v_mov_b32 v151, %1\n\t"
        "v_mov_b32 v152, %1\n\t"
        "v_mov_b32 v153, %1\n\t"
        "v_mov_b32 v154, %1\n\t"
        "v_mov_b32 v155, %1\n\t"
        "v_mov_b32 v156, %1\n\t"
        "v_mov_b32 v157, %1\n\t"
        "v_mov_b32 v158, %1\n\t"
        "v_mov_b32 v159, %1\n\t"
        "v_mov_b32 v160, %1\n\t"
        "v_mov_b32 v161, %1\n\t"
        "v_mov_b32 v162, %1\n\t"
        "v_mov_b32 v163, %1\n\t"
        "v_mov_b32 v164, %1\n\t"
        "v_mov_b32 v165, %1\n\t"
        "v_mov_b32 v166, %1\n\t"
        "v_mov_b32 v167, %1\n\t"
        "v_mov_b32 v168, %1\n\t"
        "v_mov_b32 v169, %1\n\t"
        "v_mov_b32 v170, %1\n\t"
        "v_mov_b32 v171, %1\n\t"
        "v_mov_b32 v172, %1\n\t"
        "v_mov_b32 v173, %1\n\t"
        "v_mov_b32 v174, %1\n\t"
        "v_mov_b32 v175, %1\n\t"
        "v_mov_b32 v176, %1\n\t"
        "v_mov_b32 v177, %1\n\t"
        "v_mov_b32 v178, %1\n\t"
        "v_mov_b32 v179, %1\n\t"
        "v_mov_b32 v180, %1\n\t"
        "v_mov_b32 v181, %1\n\t"
        "v_mov_b32 v182, %1\n\t"
        "v_mov_b32 v183, %1\n\t"
        "v_mov_b32 v184, %1\n\t"
        "v_mov_b32 v185, %1\n\t"
        "v_mov_b32 v186, %1\n\t"
        "v_mov_b32 v187, %1\n\t"
        "v_mov_b32 v188, %1\n\t"
        "v_mov_b32 v189, %1\n\t"
        "v_mov_b32 v190, %1\n\t"
        "v_mov_b32 v191, %1\n\t"
        "v_mov_b32 v192, %1\n\t"
        "v_mov_b32 v193, %1\n\t"
        "v_mov_b32 v194, %1\n\t"
        "v_mov_b32 v195, %1\n\t"
        "v_mov_b32 v196, %1\n\t"
        "v_mov_b32 v197, %1\n\t"
        "v_mov_b32 v198, %1\n\t"
        "v_mov_b32 v199, %1\n\t"
        "v_mov_b32 v200, %1\n\t"
        "v_mov_b32 v201, %1\n\t"
        "v_mov_b32 v202, %1\n\t"
        "v_mov_b32 v203, %1\n\t"
        "v_mov_b32 v204, %1\n\t"
        "v_mov_b32 v205, %1\n\t"
        "v_mov_b32 v206, %1\n\t"
        "v_mov_b32 v207, %1\n\t"
        "v_mov_b32 v208, %1\n\t"
        "v_mov_b32 v209, %1\n\t"
        "v_mov_b32 v210, %1\n\t"
        "v_mov_b32 v211, %1\n\t"
        "v_mov_b32 v212, %1\n\t"
        "v_mov_b32 v213, %1\n\t"
        "v_mov_b32 v214, %1\n\t"
        "v_mov_b32 v215, %1\n\t"
        "v_mov_b32 v216, %1\n\t"
        "v_mov_b32 v217, %1\n\t"
        "v_mov_b32 v218, %1\n\t"
        "v_mov_b32 v219, %1\n\t"
        "v_mov_b32 v220, %1\n\t"
        "v_mov_b32 v221, %1\n\t"
        "v_mov_b32 v222, %1\n\t"
        "v_mov_b32 v223, %1\n\t"
        "v_mov_b32 v224, %1\n\t"
        "v_mov_b32 v225, %1\n\t"
        "v_mov_b32 v226, %1\n\t"
        "v_mov_b32 v227, %1\n\t"
        "v_mov_b32 v228, %1\n\t"
        "v_mov_b32 v229, %1\n\t"
        "v_mov_b32 v230, %1\n\t"
        "v_mov_b32 v231, %1\n\t"
        "v_mov_b32 v232, %1\n\t"
        "v_mov_b32 v233, %1\n\t"
        "v_mov_b32 v234, %1\n\t"
        "v_mov_b32 v235, %1\n\t"
        "v_mov_b32 v236, %1\n\t"
        "v_mov_b32 v237, %1\n\t"
        "v_mov_b32 v238, %1\n\t"
        "v_mov_b32 v239, %1\n\t"
        "v_mov_b32 v240, %1\n\t"
        "v_mov_b32 v241, %1\n\t"
        "v_mov_b32 v242, %1\n\t"
        "v_mov_b32 v243, %1\n\t"
        "v_mov_b32 v244, %1\n\t"
        "v_mov_b32 v245, %1\n\t"
        "v_mov_b32 v246, %1\n\t"
        "v_mov_b32 v247, %1\n\t"
        "v_mov_b32 v248, %1\n\t"
        "v_mov_b32 v249, %1\n\t"
        "v_mov_b32 v250, %1\n\t"
        "v_mov_b32 v251, %1\n\t"
        "v_accvgpr_write_b32 a0, %1\n\t"
        "v_accvgpr_write_b32 a1, %1\n\t"
        "v_accvgpr_write_b32 a2, %1\n\t"
        "v_accvgpr_write_b32 a3, %1\n\t"
        "v_accvgpr_write_b32 a4, %1\n\t"
        "v_accvgpr_write_b32 a5, %1\n\t"
        "v_accvgpr_write_b32 a6, %1\n\t"
        "v_accvgpr_write_b32 a7, %1\n\t"
        "v_accvgpr_write_b32 a8, %1\n\t"
        "v_accvgpr_write_b32 a9, %1\n\t"
        "v_accvgpr_write_b32 a10, %1\n\t"
        "v_accvgpr_write_b32 a11, %1\n\t"
        "v_accvgpr_write_b32 a12, %1\n\t"
        "v_accvgpr_write_b32 a13, %1\n\t"
        "v_accvgpr_write_b32 a14, %1\n\t"
        "v_accvgpr_write_b32 a15, %1\n\t"
        "v_accvgpr_write_b32 a16, %1\n\t"
        "v_accvgpr_write_b32 a17, %1\n\t"
        "v_accvgpr_write_b32 a18, %1\n\t"
        "v_accvgpr_write_b32 a19, %1\n\t"
        "v_accvgpr_write_b32 a20, %1\n\t"
        "v_accvgpr_write_b32 a21, %1\n\t"
        "v_accvgpr_write_b32 a22, %1\n\t"
        "v_accvgpr_write_b32 a23, %1\n\t"
        "v_accvgpr_write_b32 a24, %1\n\t"
        "v_accvgpr_write_b32 a25, %1\n\t"
        "v_accvgpr_write_b32 a26, %1\n\t"
        "v_accvgpr_write_b32 a27, %1\n\t"
        "v_accvgpr_write_b32 a28, %1\n\t"
        "v_accvgpr_write_b32 a29, %1\n\t"
        "v_accvgpr_write_b32 a30, %1\n\t"
        "v_accvgpr_write_b32 a31, %1\n\t"
        "v_accvgpr_write_b32 a32, %1\n\t"
        "v_accvgpr_write_b32 a33, %1\n\t"
        "v_accvgpr_write_b32 a34, %1\n\t"
        "v_accvgpr_write_b32 a35, %1\n\t"
        "v_accvgpr_write_b32 a36, %1\n\t"
        "v_accvgpr_write_b32 a37, %1\n\t"
        "v_accvgpr_write_b32 a38, %1\n\t"
        "v_accvgpr_write_b32 a39, %1\n\t"
        "v_accvgpr_write_b32 a40, %1\n\t"
        "v_accvgpr_write_b32 a41, %1\n\t"
        "v_accvgpr_write_b32 a42, %1\n\t"
        "v_accvgpr_write_b32 a43, %1\n\t"
        "v_accvgpr_write_b32 a44, %1\n\t"
        "v_accvgpr_write_b32 a45, %1\n\t"
        "v_accvgpr_write_b32 a46, %1\n\t"
        "v_accvgpr_write_b32 a47, %1\n\t"
        "v_accvgpr_write_b32 a48, %1\n\t"
        "v_accvgpr_write_b32 a49, %1\n\t"
        "v_accvgpr_write_b32 a50, %1\n\t"
        "v_accvgpr_write_b32 a51, %1\n\t"
        "v_accvgpr_write_b32 a52, %1\n\t"
        "v_accvgpr_write_b32 a53, %1\n\t"
        "v_accvgpr_write_b32 a54, %1\n\t"
        "v_accvgpr_write_b32 a55, %1\n\t"
        "v_accvgpr_write_b32 a56, %1\n\t"
        "v_accvgpr_write_b32 a57, %1\n\t"
        "v_accvgpr_write_b32 a58, %1\n\t"
        "v_accvgpr_write_b32 a59, %1\n\t"
        "v_accvgpr_write_b32 a60, %1\n\t"
        "v_accvgpr_write_b32 a61, %1\n\t"
        "v_accvgpr_write_b32 a62, %1\n\t"
        "v_accvgpr_write_b32 a63, %1\n\t"
        "v_accvgpr_write_b32 a64, %1\n\t"
        "v_accvgpr_write_b32 a65, %1\n\t"
        "v_accvgpr_write_b32 a66, %1\n\t"
        "v_accvgpr_write_b32 a67, %1\n\t"
        "v_accvgpr_write_b32 a68, %1\n\t"
        "v_accvgpr_write_b32 a69, %1\n\t"
        "v_accvgpr_write_b32 a70, %1\n\t"
        "v_accvgpr_write_b32 a71, %1\n\t"
        "v_accvgpr_write_b32 a72, %1\n\t"
        "v_accvgpr_write_b32 a73, %1\n\t"
        "v_accvgpr_write_b32 a74, %1\n\t"
        "v_accvgpr_write_b32 a75, %1\n\t"
        "v_accvgpr_write_b32 a76, %1\n\t"
        "v_accvgpr_write_b32 a77, %1\n\t"
        "v_accvgpr_write_b32 a78, %1\n\t"
        "v_accvgpr_write_b32 a79, %1\n\t"
        "v_accvgpr_write_b32 a80, %1\n\t"
        "v_accvgpr_write_b32 a81, %1\n\t"
        "v_accvgpr_write_b32 a82, %1\n\t"
        "v_accvgpr_write_b32 a83, %1\n\t"
        "v_accvgpr_write_b32 a84, %1\n\t"
        "v_accvgpr_write_b32 a85, %1\n\t"
        "v_accvgpr_write_b32 a86, %1\n\t"
        "v_accvgpr_write_b32 a87, %1\n\t"
        "v_accvgpr_write_b32 a88, %1\n\t"
        "v_accvgpr_write_b32 a89, %1\n\t"
        "v_accvgpr_write_b32 a90, %1\n\t"
        "v_accvgpr_write_b32 a91, %1\n\t"
        "v_accvgpr_write_b32 a92, %1\n\t"
        "v_accvgpr_write_b32 a93, %1\n\t"
        "v_accvgpr_write_b32 a94, %1\n\t"
        "v_accvgpr_write_b32 a95, %1\n\t"
        "v_accvgpr_write_b32 a96, %1\n\t"
        "v_accvgpr_write_b32 a97, %1\n\t"
        "v_accvgpr_write_b32 a98, %1\n\t"
        "v_accvgpr_write_b32 a99, %1\n\t"
        "v_accvgpr_write_b32 a100, %1\n\t"
        "v_accvgpr_write_b32 a101, %1\n\t"
        "v_accvgpr_write_b32 a102, %1\n\t"
        "v_accvgpr_write_b32 a103, %1\n\t"
        "v_accvgpr_write_b32 a104, %1\n\t"
        "v_accvgpr_write_b32 a105, %1\n\t"
        "v_accvgpr_write_b32 a106, %1\n\t"
        "v_accvgpr_write_b32 a107, %1\n\t"
        "v_accvgpr_write_b32 a108, %1\n\t"
        "v_accvgpr_write_b32 a109, %1\n\t"
        "v_accvgpr_write_b32 a110, %1\n\t"
        "v_accvgpr_write_b32 a111, %1\n\t"
        "v_accvgpr_write_b32 a112, %1\n\t"
        "v_accvgpr_write_b32 a113, %1\n\t"
        "v_accvgpr_write_b32 a114, %1\n\t"
        "v_accvgpr_write_b32 a115, %1\n\t"
        "v_accvgpr_write_b32 a116, %1\n\t"
        "v_accvgpr_write_b32 a117, %1\n\t"
        "v_accvgpr_write_b32 a118, %1\n\t"
        "v_accvgpr_write_b32 a119, %1\n\t"
        "v_accvgpr_write_b32 a120, %1\n\t"
        "v_accvgpr_write_b32 a121, %1\n\t"
        "v_accvgpr_write_b32 a122, %1\n\t"
        "v_accvgpr_write_b32 a123, %1\n\t"
        "v_accvgpr_write_b32 a124, %1\n\t"
        "v_accvgpr_write_b32 a125, %1\n\t"
        "v_accvgpr_write_b32 a126, %1\n\t"
        "v_accvgpr_write_b32 a127, %1\n\t"
        "v_accvgpr_write_b32 a128, %1\n\t"
        "v_accvgpr_write_b32 a129, %1\n\t"
        "v_accvgpr_write_b32 a130, %1\n\t"
        "v_accvgpr_write_b32 a131, %1\n\t"
        "v_accvgpr_write_b32 a132, %1\n\t"
        "v_accvgpr_write_b32 a133, %1\n\t"
        "v_accvgpr_write_b32 a134, %1\n\t"
        "v_accvgpr_write_b32 a135, %1\n\t"
        "v_accvgpr_write_b32 a136, %1\n\t"
        "v_accvgpr_write_b32 a137, %1\n\t"
        "v_accvgpr_write_b32 a138, %1\n\t"
        "v_accvgpr_write_b32 a139, %1\n\t"
        "v_accvgpr_write_b32 a140, %1\n\t"
        "v_accvgpr_write_b32 a141, %1\n\t"
        "v_accvgpr_write_b32 a142, %1\n\t"
        "v_accvgpr_write_b32 a143, %1\n\t"
        "v_accvgpr_write_b32 a144, %1\n\t"
        "v_accvgpr_write_b32 a145, %1\n\t"
        "v_accvgpr_write_b32 a146, %1\n\t"
        "v_accvgpr_write_b32 a147, %1\n\t"
        "v_accvgpr_write_b32 a148, %1\n\t"
        "v_accvgpr_write_b32 a149, %1\n\t"
        "v_accvgpr_write_b32 a150, %1\n\t"
        "v_accvgpr_write_b32 a151, %1\n\t"
        "v_accvgpr_write_b32 a152, %1\n\t"
        "v_accvgpr_write_b32 a153, %1\n\t"
        "v_accvgpr_write_b32 a154, %1\n\t"
        "v_accvgpr_write_b32 a155, %1\n\t"
        "v_accvgpr_write_b32 a156, %1\n\t"
        "v_accvgpr_write_b32 a157, %1\n\t"
        "v_accvgpr_write_b32 a158, %1\n\t"
        "v_accvgpr_write_b32 a159, %1\n\t"
        "v_accvgpr_write_b32 a160, %1\n\t"
        "v_accvgpr_write_b32 a161, %1\n\t"
        "v_accvgpr_write_b32 a162, %1\n\t"
        "v_accvgpr_write_b32 a163, %1\n\t"
        "v_accvgpr_write_b32 a164, %1\n\t"
        "v_accvgpr_write_b32 a165, %1\n\t"
        "v_accvgpr_write_b32 a166, %1\n\t"
        "v_accvgpr_write_b32 a167, %1\n\t"
        "v_accvgpr_write_b32 a168, %1\n\t"
        "v_accvgpr_write_b32 a169, %1\n\t"
        "v_accvgpr_write_b32 a170, %1\n\t"
        "v_accvgpr_write_b32 a171, %1\n\t"
        "v_accvgpr_write_b32 a172, %1\n\t"
        "v_accvgpr_write_b32 a173, %1\n\t"
        "v_accvgpr_write_b32 a174, %1\n\t"
        "v_accvgpr_write_b32 a175, %1\n\t"
        "v_accvgpr_write_b32 a176, %1\n\t"
        "v_accvgpr_write_b32 a177, %1\n\t"
        "v_accvgpr_write_b32 a178, %1\n\t"
        "v_accvgpr_write_b32 a179, %1\n\t"
        "v_accvgpr_write_b32 a180, %1\n\t"
        "v_accvgpr_write_b32 a181, %1\n\t"
        "v_accvgpr_write_b32 a182, %1\n\t"
        "v_accvgpr_write_b32 a183, %1\n\t"
        "v_accvgpr_write_b32 a184, %1\n\t"
        "v_accvgpr_write_b32 a185, %1\n\t"
        "v_accvgpr_write_b32 a186, %1\n\t"
        "v_accvgpr_write_b32 a187, %1\n\t"
        "v_accvgpr_write_b32 a188, %1\n\t"
        "v_accvgpr_write_b32 a189, %1\n\t"
        "v_accvgpr_write_b32 a190, %1\n\t"
        "v_accvgpr_write_b32 a191, %1\n\t"
        "v_accvgpr_write_b32 a192, %1\n\t"
        "v_accvgpr_write_b32 a193, %1\n\t"
        "v_accvgpr_write_b32 a194, %1\n\t"
        "v_accvgpr_write_b32 a195, %1\n\t"
        "v_accvgpr_write_b32 a196, %1\n\t"
        "v_accvgpr_write_b32 a197, %1\n\t"
        "v_accvgpr_write_b32 a198, %1\n\t"
        "v_accvgpr_write_b32 a199, %1\n\t"
        "v_accvgpr_write_b32 a200, %1\n\t"
        "v_accvgpr_write_b32 a201, %1\n\t"
        "v_accvgpr_write_b32 a202, %1\n\t"
        "v_accvgpr_write_b32 a203, %1\n\t"
        "v_accvgpr_write_b32 a204, %1\n\t"
        "v_accvgpr_write_b32 a205, %1\n\t"
        "v_accvgpr_write_b32 a206, %1\n\t"
        "v_accvgpr_write_b32 a207, %1\n\t"
        "v_accvgpr_write_b32 a208, %1\n\t"
        "v_accvgpr_write_b32 a209, %1\n\t"
        "v_accvgpr_write_b32 a210, %1\n\t"
        "v_accvgpr_write_b32 a211, %1\n\t"
        "v_accvgpr_write_b32 a212, %1\n\t"
        "v_accvgpr_write_b32 a213, %1\n\t"
        "v_accvgpr_write_b32 a214, %1\n\t"
        "v_accvgpr_write_b32 a215, %1\n\t"
        "v_accvgpr_write_b32 a216, %1\n\t"
        "v_accvgpr_write_b32 a217, %1\n\t"
        "v_accvgpr_write_b32 a218, %1\n\t"
        "v_accvgpr_write_b32 a219, %1\n\t"
        "v_accvgpr_write_b32 a220, %1\n\t"
        "v_accvgpr_write_b32 a221, %1\n\t"
        "v_accvgpr_write_b32 a222, %1\n\t"
        "v_accvgpr_write_b32 a223, %1\n\t"
        "v_accvgpr_write_b32 a224, %1\n\t"
        "v_accvgpr_write_b32 a225, %1\n\t"
        "v_accvgpr_write_b32 a226, %1\n\t"
        "v_accvgpr_write_b32 a227, %1\n\t"
        "v_accvgpr_write_b32 a228, %1\n\t"
        "v_accvgpr_write_b32 a229, %1\n\t"
        "v_accvgpr_write_b32 a230, %1\n\t"
        "v_accvgpr_write_b32 a231, %1\n\t"
        "v_accvgpr_write_b32 a232, %1\n\t"
        "v_accvgpr_write_b32 a233, %1\n\t"
        "v_accvgpr_write_b32 a234, %1\n\t"
        "v_accvgpr_write_b32 a235, %1\n\t"
        "v_accvgpr_write_b32 a236, %1\n\t"
        "v_accvgpr_write_b32 a237, %1\n\t"
        "v_accvgpr_write_b32 a238, %1\n\t"
        "v_accvgpr_write_b32 a239, %1\n\t"
        "v_accvgpr_write_b32 a240, %1\n\t"
        "v_accvgpr_write_b32 a241, %1\n\t"
        "v_accvgpr_write_b32 a242, %1\n\t"
        "v_accvgpr_write_b32 a243, %1\n\t"
        "v_accvgpr_write_b32 a244, %1\n\t"
        "v_accvgpr_write_b32 a245, %1\n\t"
        "v_accvgpr_write_b32 a246, %1\n\t"
        "v_accvgpr_write_b32 a247, %1\n\t"
        "v_accvgpr_write_b32 a248, %1\n\t"
        "v_accvgpr_write_b32 a249, %1\n\t"
        "v_accvgpr_write_b32 a250, %1\n\t"
        "v_accvgpr_write_b32 a251, %1\n\t"
        "v_accvgpr_write_b32 a252, %1\n\t"
        "v_accvgpr_write_b32 a253, %1\n\t"
        "v_accvgpr_write_b32 a254, %1\n\t"
        "v_accvgpr_write_b32 a255, %1\n\t"
        "s_mov_b32 s36, %2\n\t"
        "s_mov_b32 s37, %2\n\t"
        "s_mov_b32 s38, %2\n\t"
        "s_mov_b32 s39, %2\n\t"
        "s_mov_b32 s40, %2\n\t"
        "s_mov_b32 s41, %2\n\t"
        "s_mov_b32 s42, %2\n\t"
        "s_mov_b32 s43, %2\n\t"
        "s_mov_b32 s44, %2\n\t"
        "s_mov_b32 s45, %2\n\t"
        "s_mov_b32 s46, %2\n\t"
        "s_mov_b32 s47, %2\n\t"
        "s_mov_b32 s48, %2\n\t"
        "s_mov_b32 s49, %2\n\t"
        "s_mov_b32 s50, %2\n\t"
        "s_mov_b32 s51, %2\n\t"
        "s_mov_b32 s52, %2\n\t"
        "s_mov_b32 s53, %2\n\t"
        "s_mov_b32 s54, %2\n\t"
        "s_mov_b32 s55, %2\n\t"
        "s_mov_b32 s56, %2\n\t"
        "s_mov_b32 s57, %2\n\t"
        "s_mov_b32 s58, %2\n\t"
        "s_mov_b32 s59, %2\n\t"
        "s_mov_b32 s60, %2\n\t"
        "s_mov_b32 s61, %2\n\t"
        "s_mov_b32 s62, %2\n\t"
        "s_mov_b32 s63, %2\n\t"
        "s_mov_b32 s64, %2\n\t"
        "s_mov_b32 s65, %2\n\t"
        "s_mov_b32 s66, %2\n\t"
        "s_mov_b32 s67, %2\n\t"
        "s_mov_b32 s68, %2\n\t"
        "s_mov_b32 s69, %2\n\t"
        "s_mov_b32 s70, %2\n\t"
        "s_mov_b32 s71, %2\n\t"
        "s_mov_b32 s72, %2\n\t"
        "s_mov_b32 s73, %2\n\t"
        "s_mov_b32 s74, %2\n\t"
        "s_mov_b32 s75, %2\n\t"
        "s_mov_b32 s76, %2\n\t"
        "s_mov_b32 s77, %2\n\t"
        "s_mov_b32 s78, %2\n\t"
        "s_mov_b32 s79, %2\n\t"
        "s_mov_b32 s80, %2\n\t"
        "s_mov_b32 s81, %2\n\t"
        "s_mov_b32 s82, %2\n\t"
        "s_mov_b32 s83, %2\n\t"
        "s_mov_b32 s84, %2\n\t"
        "s_mov_b32 s85, %2\n\t"
        "s_mov_b32 s86, %2\n\t"
        "s_mov_b32 s87, %2\n\t"
        "s_mov_b32 s88, %2\n\t"
        "s_mov_b32 s89, %2\n\t"
        "s_mov_b32 s90, %2\n\t"
        "s_mov_b32 s91, %2\n\t"
        "s_mov_b32 s92, %2\n\t"
        "s_mov_b32 s93, %2\n\t"
        "s_mov_b32 s94, %2\n\t"
        "s_mov_b32 s95, %2\n\t"
        "s_mov_b32 s96, %2\n\t"
        "s_mov_b32 s97, %2\n\t"
        "s_mov_b32 s98, %2\n\t"
        "s_mov_b32 s99, %2\n\t"
        "v_mov_b32 %0, v251\n\t"
        : "=v"(v) : "v"(pat), "s"(pat) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    if (v == 0x12345u) sink[0] = v;
}
extern "C" int reg_poison(uint32_t pat)
{
    uint32_t *sink = nullptr;
    if (hipMalloc(&sink, 64) != hipSuccess) return -1;
    // (512 registers per wave: one wave per SIMD, one 4-wave workgroup per CU at a time)
    hipLaunchKernelGGL(reg_poison_kernel, dim3(256 * 16), dim3(256), 0, nullptr, pat, sink);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    (void)hipFree(sink);
    return rc;
}

// `pat` in LDS words [lo, hi), `other` elsewhere (to find WHICH words a kernel reads before writing them)
__global__ void __launch_bounds__(256) lds_poison_range_kernel(uint32_t pat, uint32_t other, int lo, int hi, int words, uint32_t *sink)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = (i >= lo && i < hi) ? pat : other;
    __syncthreads();
    uint32_t s = 0;
    for (int r = 0; r < 4; ++r)
        for (int i = threadIdx.x; i < words; i += 256) s += lds[i] ^ (uint32_t)r;
    if (s == 0x12345u) sink[0] = s;
}
extern "C" int lds_poison_range(uint32_t pat, uint32_t other, int lo_word, int hi_word)
{
    uint32_t *sink = nullptr;
    if (hipMalloc(&sink, 64) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(lds_poison_range_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -3;
    hipLaunchKernelGGL(lds_poison_range_kernel, dim3(256 * 8), dim3(256), LDS_BYTES, nullptr, pat, other, lo_word, hi_word,
                       LDS_BYTES / 4, sink);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    (void)hipFree(sink);
    return rc;
}
