// Round-robin tournament geometry shared by the LDS Jacobi eigensolver (eigh_lds.hip), the kernel
// that replays a recorded rotation sequence on the rows of V (eigh_apply_rot_kernel) and the LDS
// one-sided Jacobi SVD (svd.hip).
#pragma once

// ---- tournament ring geometry (NP even, m = NP/2 pairs, R = NP-1 ring positions) ----------
// t = 0: pair k = (top_k, bot_k) = (2k, 2k+1); top_0 = index 0 never moves; ring position r holds
//   top_{r+1} for r <= m-2 and bot_{2m-2-r} for r >= m-1; every step turns the ring by +1.
__host__ __device__ inline int ring_pos_top(int k) { return k - 1; }               // k >= 1
__host__ __device__ inline int ring_pos_bot(int k, int m) { return 2 * m - 2 - k; }
__host__ __device__ inline int ring_index0(int r, int m) {  // original index at ring position r, t = 0
    return r <= m - 2 ? 2 * (r + 1) : 2 * (2 * m - 2 - r) + 1;
}
__host__ __device__ inline int ring_pos_of(int i, int m) {  // inverse of ring_index0, i >= 1
    return (i & 1) ? 2 * m - 2 - (i >> 1) : (i >> 1) - 1;
}

