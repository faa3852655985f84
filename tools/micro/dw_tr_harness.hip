// Stand-alone timing harness of dw_tr_kernel (csrc/dw_tr.hip) at config 2's shape with synthetic operands: no python, no torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iscenario-wise-rec_amd/csrc [-DDT_STAMPS ...] tools/micro/dw_tr_harness.hip -o tools/micro/bin/dw_tr_harness
// Prints the mean kernel time over 50 launches (HIP events) and, with -DDT_STAMPS, the shader-clock stamps of workgroup 0 / wave 0
// (prologue end, every stage end, kernel end).  Results are not checked here (tools/micro/dw_probe.py does that through the C ABI).
#include "../../scenario-wise-rec_amd/csrc/dw_tr.hip"

#include <cstdio>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 65536;
    const int K1 = 148, NR = 10, OHW = 128, K2 = 16 * NR + OHW, LD = 160;
    const int n_tiles = static_cast<int>((M + 31) / 32);
    const int rows = 4096, pieces = 2 * NR;
    std::mt19937 rng(1);
    std::vector<float> h(static_cast<size_t>(M) * LD);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) v = nd(rng);
    float *dY, *Z, *coef, *part;
    CK(hipMalloc(&dY, h.size() * 4)); CK(hipMalloc(&Z, h.size() * 4));
    CK(hipMemcpy(dY, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (auto& v : h) v = nd(rng);
    CK(hipMemcpy(Z, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&coef, 4 * 160 * 4));
    CK(hipMemcpy(coef, h.data(), 4 * 160 * 4, hipMemcpyHostToDevice));
    // workspace: [shadow: rows x pieces x 48 bytes of bf16 terms][voff]
    const size_t shadow_bytes = static_cast<size_t>(rows) * pieces * 48;
    std::vector<uint16_t> sh(shadow_bytes / 2);
    for (auto& v : sh) v = static_cast<uint16_t>(0x3C00 + (rng() & 0xFF));          // small finite bf16 values
    std::vector<uint32_t> vo(static_cast<size_t>(n_tiles) * NR * 64);
    for (int T = 0; T < n_tiles; ++T)
        for (int g = 0; g < NR; ++g)
            for (int l = 0; l < 64; ++l) {
                const int s = l >> 5;
                const uint32_t row = rng() % rows;
                vo[(static_cast<size_t>(T) * NR + g) * 64 + l] = static_cast<uint32_t>((static_cast<size_t>(row) * pieces + 2 * g + s) * 48);
            }
    char* ws;
    CK(hipMalloc(&ws, shadow_bytes + vo.size() * 4));
    CK(hipMemcpy(ws, sh.data(), shadow_bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(ws + shadow_bytes, vo.data(), vo.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> mk(static_cast<size_t>(4) * M);
    for (auto& v : mk) v = 1u << (rng() & 31);
    uint32_t* mask_t;
    CK(hipMalloc(&mask_t, mk.size() * 4));
    CK(hipMemcpy(mask_t, mk.data(), mk.size() * 4, hipMemcpyHostToDevice));
    DwTrArgs a;
    a.dY = reinterpret_cast<const char*>(dY); a.Z = reinterpret_cast<const char*>(Z);
    a.lddy_b = LD * 4; a.ldz_b = LD * 4;
    a.ca = coef; a.cb = coef + 160; a.cc = coef + 320; a.mean = coef + 480;
    a.M = M; a.K1 = K1; a.K2 = K2;
    a.ws = ws; a.voff = reinterpret_cast<const uint32_t*>(ws + shadow_bytes); a.mask_t = mask_t; a.NR = NR;
    a.rows_per_split = std::max<int64_t>(128, ((M + 255) / 256 + 31) / 32 * 32);
    a.n_splits = static_cast<int>((M + a.rows_per_split - 1) / a.rows_per_split);
    a.k2p = K2;
    CK(hipMalloc(&part, (static_cast<size_t>(a.n_splits) * K1 * K2 + static_cast<size_t>(a.n_splits) * K1) * 4));
    a.part = part; a.part_cs = part + static_cast<size_t>(a.n_splits) * K1 * K2;
#ifdef DT_STAMPS
    uint64_t* stamps;
    CK(hipMalloc(&stamps, 64 * 8));
    CK(hipMemset(stamps, 0, 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(dt_stamp_buf), &stamps, sizeof(stamps)));
#endif
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int i = 0; i < 5; ++i)
        if (dw_tr_launch(a, st) != SWR_OK) { printf("launch failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 50;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) dw_tr_launch(a, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("dw_tr_kernel M=%lld splits=%d rows/split=%lld: %.2f us per launch\n", static_cast<long long>(M), a.n_splits,
           static_cast<long long>(a.rows_per_split), ms * 1e3 / n);
#ifdef DT_STAMPS
    uint64_t hs[64];
    CK(hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost));
    printf("stamps of workgroup 0, wave 0 (shader cycles since kernel entry):");
    for (int i = 1; i < 16 && hs[i]; ++i) printf(" %llu", static_cast<unsigned long long>(hs[i] - hs[0]));
    printf("\n");
#ifdef DT_STAMPS_FINE
    printf("stage 3 (cycles since its start): issue %llu  first-half %llu  second-half %llu  wait %llu  barrier %llu\n",
           (unsigned long long)(hs[17] - hs[16]), (unsigned long long)(hs[18] - hs[16]), (unsigned long long)(hs[19] - hs[16]),
           (unsigned long long)(hs[20] - hs[16]), (unsigned long long)(hs[21] - hs[16]));
#endif
#endif
    return 0;
}
