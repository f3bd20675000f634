// Device-side counter hash shared by the generator kernels (see synth.hip).
#pragma once
#include <cstdint>

__host__ __device__ __forceinline__ uint64_t nbx_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Index of the unordered pair (a,b) in a packed lower triangle.
__host__ __device__ __forceinline__ uint64_t nbx_tri(uint64_t a, uint64_t b) {
    const uint64_t hi = a > b ? a : b;
    const uint64_t lo = a > b ? b : a;
    return hi * (hi + 1) / 2 + lo;
}

// nbx_tri for operands known to fit 32 bits (N < 92681): same values, 32 x 32 -> 64-bit
// multiplies instead of 64 x 64 (the generators evaluate one of these per integral).
__host__ __device__ __forceinline__ uint32_t nbx_tri_pair_u32(uint32_t a, uint32_t b) {
    const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
    return hi * (hi + 1u) / 2u + lo;
}
__host__ __device__ __forceinline__ uint64_t nbx_tri_u32(uint32_t a, uint32_t b) {
    const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
    return (uint64_t)hi * ((uint64_t)hi + 1ull) / 2ull + lo;
}

__host__ __device__ __forceinline__ double nbx_synth_val(uint64_t stream, uint64_t k, uint64_t seed) {
    const uint64_t u = nbx_splitmix64(((stream << 48) | k) ^ seed);
    return (double)(u >> 11) * 0x1.0p-53 * 2.0 - 1.0;
}
