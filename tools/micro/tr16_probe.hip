// What does ds_read_b64_tr_b16 return?  LDS is filled with its own element index (as 16-bit integers); every lane
// supplies address = 8 * lane (pattern 0), or the row/segment pattern a 4 x 16 matrix per 16-lane group wants
// (pattern 1), and prints the four 16-bit values it receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(int pattern, int pitch, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = static_cast<uint16_t>(i);
    __syncthreads();
    const int l = threadIdx.x;
    int elem;
    if (pattern == 0) elem = 4 * l;
    else {
        const int grp = l >> 4, i = l & 15;
        elem = grp * 4 * pitch + (i >> 2) * pitch + 4 * (i & 3);      // row i/4 of the group's 4 x 16 matrix, columns 4 (i % 4) ..
    }
    typedef __attribute__((address_space(3))) bf16x4* lptr;
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(lds + elem));
    s16x4 r = __builtin_bit_cast(s16x4, v);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = static_cast<uint16_t>(r[j]);
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int pattern = 0; pattern < 2; ++pattern) {
        const int pitch = pattern ? 40 : 0;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, pattern, pitch, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d (pitch %d elements)\n", pattern, pitch);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
