// Transposed reduction for the multi-query scans (u8_scan_multi_kernel, bin_scan_multi_kernel).
//
// G adjacent lanes hold partial sums of one row for NQ queries each: acc[0..NQ).  What is wanted is
// lane j of the group holding the TOTAL of one query -- a reduce-scatter, not NQ all-reduces: at
// distance d = 1, 2, 4 ... a lane keeps one half of its values (the lower half when its bit d is 0,
// the upper half when it is 1), hands the other half to its partner lane ^ d and adds what the
// partner hands over.  After log2(NQ) steps every lane holds ONE value: the sum over the 2^steps
// lanes that share its upper sub bits of query  j = bitreverse(sub's low log2(NQ) bits); the
// remaining distances up to G/2 are plain all-reduce adds of that one value.  NQ = 8, G = 8: 7 adds
// and 14 selects instead of 24 adds for eight all-reduces -- and the per-query epilogue / metric
// then runs once per lane instead of NQ times.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace qamd {

// value of lane ^ D inside a 16-lane row: D = 1, 2 quad_perm DPP; D = 8 row_ror:8 DPP; D = 4 ds_swizzle
template <int D> __device__ __forceinline__ uint32_t row_xor(uint32_t v) {
    static_assert(D == 1 || D == 2 || D == 4 || D == 8, "distance inside a DPP row");
    if (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    if (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    if (D == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);  // row_ror:8
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);                                 // bit mode: xor 4
}

template <int D, int N> __device__ __forceinline__ void scatter_step(uint32_t (&v)[N], int half, bool upper) {
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
        if (i < half) {
            const uint32_t keep = upper ? v[i + half] : v[i];
            const uint32_t send = upper ? v[i] : v[i + half];
            v[i] = keep + row_xor<D>(send);
        }
    }
}

// In: acc[NQ] partial sums of this lane.  Out: acc[0] = total over the G lanes of the row group for
// query multi_query_of<NQ>(sub).  G, NQ powers of two, NQ <= G <= 16.
template <int G, int NQ> __device__ __forceinline__ uint32_t multi_reduce(uint32_t (&acc)[NQ], int sub) {
    static_assert(NQ == 1 || NQ == 2 || NQ == 4 || NQ == 8, "queries per pass");
    static_assert(G >= NQ && G <= 16, "row group");
    if (NQ >= 2) scatter_step<1, NQ>(acc, NQ / 2, (sub & 1) != 0);
    if (NQ >= 4) scatter_step<2, NQ>(acc, NQ / 4, (sub & 2) != 0);
    if (NQ >= 8) scatter_step<4, NQ>(acc, NQ / 8, (sub & 4) != 0);
    uint32_t v = acc[0];
    if (G >= 2 && NQ < 2) v += row_xor<1>(v);
    if (G >= 4 && NQ < 4) v += row_xor<2>(v);
    if (G >= 8 && NQ < 8) v += row_xor<4>(v);
    if (G >= 16) v += row_xor<8>(v);
    return v;
}

// Which query's total lane `sub` holds after multi_reduce (the lanes with sub >= NQ hold copies).
template <int NQ> __device__ __forceinline__ int multi_query_of(int sub) {
    if (NQ == 8) return ((sub & 1) << 2) | (sub & 2) | ((sub & 4) >> 2);
    if (NQ == 4) return ((sub & 1) << 1) | ((sub & 2) >> 1);
    if (NQ == 2) return sub & 1;
    return 0;
}

}  // namespace qamd
