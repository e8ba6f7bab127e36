// What the v_mfma_f32_32x32x16 sparse-conv kernels share: vector types, the MFMA wrapper and the weight fragment tiling.  Included inside the
// anonymous namespace of each translation unit that holds such kernels (sparse_conv.hip, sparse_conv_st.hip).
#pragma once

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <typename E> struct Mfma32;
template <> struct Mfma32<__bf16> {
    static __device__ __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma32<_Float16> {
    static __device__ __forceinline__ f32x16 run(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

template <int CIN, int COUT> struct Tiling32 {
    static_assert(CIN % 16 == 0, "CIN must be a multiple of 16");
    static_assert(COUT % 16 == 0, "COUT must be a multiple of 16");
    static constexpr int S = CIN / 16;                 // K-steps per kernel offset
    static constexpr int T = (COUT + 31) / 32;         // cout tiles (a 16-cout layer half-fills its tile)
    static constexpr int GROUP_U4 = S * T * 64;        // uint4 per kernel offset: S * T KiB
    static constexpr int GROUP_BYTES = GROUP_U4 * 16;
};
