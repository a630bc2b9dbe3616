// Deterministic block reduction of N per-lane fp32 values on CDNA4 (wave64):
//   1. two DPP quad-permute adds (lanes 4k..4k+3 -> every lane of the quad holds the quad sum): no LDS, no ds_bpermute
//   2. one lane per quad stores its N values to an LDS row (64 rows for a 256-thread block)
//   3. thread j < N sums column j over the rows in a fixed order, IN FP64, and writes out[j] (rounded to fp32 once)
// At step 3 all lanes read the same row -> consecutive LDS addresses, conflict free.
// The fp64 column sum matters: the reduced BA system H_A - H_sc cancels ~100x, so fp32 running sums over 64 rows
// (error ~5e-7) would show up as ~1e-5 in the recovered poses; with fp64 here the block partial is good to one rounding.
// This replaces the reference's SSE-lane + 3-tier accumulators (OptimizationBackend/MatrixAccumulators.h) and the
// per-thread accumulator replicas of IndexThreadReduce; the cross-block finish is fp64 in a separate kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace nalo {

__device__ __forceinline__ float dpp_quad_xor1(float v) {   // quad_perm [1,0,3,2]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {   // quad_perm [2,3,0,1]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}

// sum over the four lanes of a quad. The empty asm statements are opaque to the SLP vectoriser: left alone it packs pairs of these adds into v_pk_add_f32, which
// cannot take a DPP operand, and every add then needs its own v_mov_b32_dpp; scalar adds fold with the moves into v_add_f32_dpp (ba_linearize: 2640 -> 2540
// vector instructions per lane, same operations, same results).
__device__ __forceinline__ float dpp_quad_sum(float v) {
    v += dpp_quad_xor1(v); asm volatile("" : "+v"(v));
    v += dpp_quad_xor2(v); asm volatile("" : "+v"(v));
    return v;
}

// smem: (NT/4) * (N+1) floats. out: N floats (global or LDS). All NT threads must call.
template <int N, int NT>
__device__ __forceinline__ void block_reduce_cols(float (&v)[N], float* smem, float* out) {
    constexpr int NP = N + 1;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_quad_sum(v[i]);
    const int tid = threadIdx.x;
    if ((tid & 3) == 0) {
        float* row = smem + (tid >> 2) * NP;
#pragma unroll
        for (int i = 0; i < N; ++i) row[i] = v[i];
    }
    __syncthreads();
    if (tid < N) {
        double s = 0.0;
#pragma unroll 8
        for (int r = 0; r < NT / 4; ++r) s += (double)smem[r * NP + tid];
        out[tid] = (float)s;
    }
    __syncthreads();
}

}  // namespace nalo
