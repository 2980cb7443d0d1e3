// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11) and the random-stream layout of the HIP kernels (DESIGN.md "RNG").
//
//   key     = (seed lo, seed hi)
//   counter = (block, frame lo, frame hi, stream << 28 | cell)
//   stream 0: data bits   -- replaces randi([0 1], ...)   matlab/main_BER_calculation.m:246
//   stream 1: unit noise  -- replaces randn + 1j*randn    matlab/main_BER_calculation.m:290
//
// Bits: subcarrier n of symbol s owns a kslot-bit field (kslot = 2, 4, 8 for k = 2, 4, 6) at
// bit offset n*kslot of the symbol's bit stream, i.e. block s*(N*kslot/128) + (n*kslot >> 7),
// word (n*kslot >> 5) & 3, shift n*kslot & 31; the label is the low k bits (bit k-1 = first
// bit of the subcarrier, as qammod's 'bit' input orders them).
// Noise: block p carries the complex unit normals of samples 2p (words 0,1) and 2p+1
// (words 2,3): u1 = fma(w_a, 2^-32, 2^-33), u2 = (w_b >> 9) * 2^-23 (the top 23 bits),
// n = sqrt(-2 ln u1) * (cos 2 pi u2 + j sin 2 pi u2).
#pragma once
#include <stdint.h>

#define WOFDM_STREAM_BITS  0u
#define WOFDM_STREAM_NOISE 1u

struct philox_out { uint32_t w[4]; };

__host__ __device__ __forceinline__ philox_out philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                             uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    philox_out o;
    o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}
