// fp32 -> three bf16 terms x = h + m + l (8 + 8 + 8 significant bits), exact; shared by the bf16-split products (gemm.hip) and the
// MFMA segment sums of the embedding backward (embed_bwd.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Bf3 {
    __bf16 h, m, l;
};
__device__ __forceinline__ Bf3 split3(float x) {
    Bf3 r;
    r.h = static_cast<__bf16>(x);
    const float r1 = x - static_cast<float>(r.h);
    r.m = static_cast<__bf16>(r1);
    const float r2 = r1 - static_cast<float>(r.m);
    r.l = static_cast<__bf16>(r2);
    return r;
}
#define SPLIT3_INTO(x, H, M, L, idx)       \
    do {                                   \
        const Bf3 s3_ = split3(x);         \
        H[idx] = s3_.h;                    \
        M[idx] = s3_.m;                    \
        L[idx] = s3_.l;                    \
    } while (0)

// Two values at a time: v_cvt_pk_bf16_f32 rounds both, v_pk_add_f32 takes both remainders -- 4.5 VALU per value instead
// of 7 (the split is the VALU work that sits in front of every MFMA group); bit-identical to split3().
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#ifdef SWR_SPLIT_SCALAR
#define SPLIT3_PAIR(x0, x1, H, M, L, idx) do { SPLIT3_INTO(x0, H, M, L, idx); SPLIT3_INTO(x1, H, M, L, (idx) + 1); } while (0)
#else
#define SPLIT3_PAIR(x0, x1, H, M, L, idx)                                   \
    do {                                                                    \
        const f32x2 sx_ = {x0, x1};                                         \
        const bf16x2 sh_ = __builtin_convertvector(sx_, bf16x2);            \
        const f32x2 s1_ = sx_ - __builtin_convertvector(sh_, f32x2);        \
        const bf16x2 sm_ = __builtin_convertvector(s1_, bf16x2);            \
        const f32x2 s2_ = s1_ - __builtin_convertvector(sm_, f32x2);        \
        const bf16x2 sl_ = __builtin_convertvector(s2_, bf16x2);            \
        H[idx] = sh_[0]; H[(idx) + 1] = sh_[1];                             \
        M[idx] = sm_[0]; M[(idx) + 1] = sm_[1];                             \
        L[idx] = sl_[0]; L[(idx) + 1] = sl_[1];                             \
    } while (0)
#endif

#define CVT_PAIR(x0, x1, H, idx)                                            \
    do {                                                                    \
        const f32x2 cx_ = {x0, x1};                                         \
        const bf16x2 ch_ = __builtin_convertvector(cx_, bf16x2);            \
        H[idx] = ch_[0]; H[(idx) + 1] = ch_[1];                             \
    } while (0)

