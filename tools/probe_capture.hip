// Probe for the frame kernel's "dual-field capture" loop (tools/, not product code): how fast can one wave per frame walk its
// 486 rows when a line costs only the cell gather + four compares + parking the ballots in a lane?
//   variant 0: pure streaming read of the frame (16-byte loads), the memory-side ceiling of the one-wave-per-frame pattern
//   variant 1: direct byte gathers of the two cells a lane owns, rows 2k and 2k+1 together (both fields), D row pairs in flight
//   variant 2: as 1 but one field after the other (rows 0,2,4.., then 1,3,5..) - the access order of the round-1 kernel
// hipcc --offload-arch=gfx950 -O3 -o probe_capture probe_capture.hip ; ./probe_capture [frames] [variant] [depth]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 720, H = 486;

struct Args { const uint8_t *luma; uint4 *out; int n; uint32_t lo, hi; };

// clang has no builtin for v_writelane_b32; the LLVM intrinsic is reachable by its name
extern "C" __device__ uint32_t sdv_llvm_writelane(uint32_t val, uint32_t lane, uint32_t old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t wl(uint32_t old, uint32_t val, int lane) { return sdv_llvm_writelane(val, (uint32_t)lane, old); }

template <int WPE, bool WR = false>
__global__ void __launch_bounds__(64, WPE) k_stream(Args a)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    const uint4 *p = (const uint4 *)(a.luma + (size_t)f * W * H);
    const int nvec = W * H / 16;
    uint4 acc = {0, 0, 0, 0};
    for (int i = lane; i < nvec; i += 64 * 4) {
        uint4 v0 = p[i], v1 = i + 64 < nvec ? p[i + 64] : uint4{0, 0, 0, 0}, v2 = i + 128 < nvec ? p[i + 128] : uint4{0, 0, 0, 0}, v3 = i + 192 < nvec ? p[i + 192] : uint4{0, 0, 0, 0};
        acc.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x; acc.y ^= v0.y ^ v1.y ^ v2.y ^ v3.y; acc.z ^= v0.z ^ v1.z ^ v2.z ^ v3.z; acc.w ^= v0.w ^ v1.w ^ v2.w ^ v3.w;
        if (WR && (i & 0xF00) == 0) a.out[(size_t)f * 1536 + (i >> 12) * 64 * 4 + lane] = acc, a.out[(size_t)f * 1536 + (i >> 12) * 64 * 4 + 64 + lane] = acc, a.out[(size_t)f * 1536 + (i >> 12) * 64 * 4 + 128 + lane] = acc, a.out[(size_t)f * 1536 + (i >> 12) * 64 * 4 + 192 + lane] = acc;
    }
    a.out[(size_t)f * 64 + lane] = acc;
}

// per row: two byte gathers (cells lane and lane+64), four compares, 8 writelanes into the batch registers of "its" lane
template <int D, bool DUAL, int WPE, bool NT_LD = false>
__global__ void __launch_bounds__(64, WPE) k_capture(Args a)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    auto LDB = [](const uint8_t *p) -> uint8_t { return NT_LD ? __builtin_nontemporal_load(p) : *p; };
    const uint8_t *frame = a.luma + (size_t)f * W * H;
    // cell centres: 132 cells between px 12 and 708, data cell b at cell b+3
    const uint32_t psm = ((708 - 12) * 128 + 66) / 132, hpsm = (psm + 1) / 2;
    const int x0 = (int)(((uint32_t)(lane + 3) * psm + hpsm) / 128) + 12, x1 = (int)(((uint32_t)(lane + 67) * psm + hpsm) / 128) + 12;
    uint32_t b0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint4 acc = {0, 0, 0, 0};
    constexpr int NP = H / 2;                       // row pairs
    // DUAL: iteration k = rows 2k, 2k+1.  else: iteration k = two consecutive rows of one field (2 lines per iteration like round 1)
    auto rowA = [&](int k) -> const uint8_t * { if (DUAL) return frame + (size_t)(2 * k) * W; int fld = k >= (NP + 1) / 2 ? 1 : 0; int kk = fld ? k - (NP + 1) / 2 : k; return frame + (size_t)(4 * kk + fld) * W; };
    auto rowB = [&](int k) -> const uint8_t * { if (DUAL) return frame + (size_t)(2 * k + 1) * W; int fld = k >= (NP + 1) / 2 ? 1 : 0; int kk = fld ? k - (NP + 1) / 2 : k; int r = 4 * kk + 2 + fld; if (r >= H) r = H - 1; return frame + (size_t)r * W; };
    uint8_t q[D][4];
#pragma unroll
    for (int d = 0; d < D; d++) { const uint8_t *ra = rowA(d), *rb = rowB(d); q[d][0] = LDB(ra + x0); q[d][1] = LDB(ra + x1); q[d][2] = LDB(rb + x0); q[d][3] = LDB(rb + x1); }
    for (int k0 = 0; k0 < NP; k0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int k = k0 + d;
            if (k < NP) {
                const uint8_t p0 = q[d][0], p1 = q[d][1], p2 = q[d][2], p3 = q[d][3];
                const int kn = k + D < NP ? k + D : NP - 1;
                { const uint8_t *ra = rowA(kn), *rb = rowB(kn); q[d][0] = LDB(ra + x0); q[d][1] = LDB(ra + x1); q[d][2] = LDB(rb + x0); q[d][3] = LDB(rb + x1); }
                const uint64_t aA_lo = __ballot(p0 > a.lo), bA_lo = __ballot(p0 >= a.hi), aA_hi = __ballot(p1 > a.lo), bA_hi = __ballot(p1 >= a.hi);
                const uint64_t aB_lo = __ballot(p2 > a.lo), bB_lo = __ballot(p2 >= a.hi), aB_hi = __ballot(p3 > a.lo), bB_hi = __ballot(p3 >= a.hi);
                const int j = k & 63;
                b0[0] = wl(b0[0], (uint32_t)aA_lo, j); b0[1] = wl(b0[1], (uint32_t)(aA_lo >> 32), j); b0[2] = wl(b0[2], (uint32_t)bA_lo, j); b0[3] = wl(b0[3], (uint32_t)(bA_lo >> 32), j);
                b0[4] = wl(b0[4], (uint32_t)aA_hi, j); b0[5] = wl(b0[5], (uint32_t)(aA_hi >> 32), j); b0[6] = wl(b0[6], (uint32_t)bA_hi, j); b0[7] = wl(b0[7], (uint32_t)(bA_hi >> 32), j);
                b1[0] = wl(b1[0], (uint32_t)aB_lo, j); b1[1] = wl(b1[1], (uint32_t)(aB_lo >> 32), j); b1[2] = wl(b1[2], (uint32_t)bB_lo, j); b1[3] = wl(b1[3], (uint32_t)(bB_lo >> 32), j);
                b1[4] = wl(b1[4], (uint32_t)aB_hi, j); b1[5] = wl(b1[5], (uint32_t)(aB_hi >> 32), j); b1[6] = wl(b1[6], (uint32_t)bB_hi, j); b1[7] = wl(b1[7], (uint32_t)(bB_hi >> 32), j);
                if (j == 63 || k == NP - 1) {
                    // stands in for phase B: ~48 bytes of record per line written by "its" lane, for both batches
                    uint4 r0 = {b0[0] ^ b0[4], b0[1] ^ b0[5], b0[2] ^ b0[6], b0[3] ^ b0[7]}, r1 = {b1[0] ^ b1[4], b1[1] ^ b1[5], b1[2] ^ b1[6], b1[3] ^ b1[7]};
                    uint4 *o = a.out + ((size_t)f * 512 + (size_t)(k & ~63) * 2) * 3;
                    if (lane <= j) { o[lane * 3] = r0; o[lane * 3 + 1] = r1; o[lane * 3 + 2] = r0; o[(64 + lane) * 3] = r1; o[(64 + lane) * 3 + 1] = r0; o[(64 + lane) * 3 + 2] = r1; }
                    acc.x ^= r0.x ^ r1.x;
                }
            }
        }
    }
    if (acc.x == 0x12345678u) a.out[0] = acc;
}

// variant 3: rows 2k, 2k+1 (1440 contiguous bytes) DMA'd into a per-wave LDS ring of D slots by two global_load_lds_dwordx4
// (64 + 26 lanes x 16 bytes), the four cells of a lane read back from LDS as bytes
typedef __attribute__((address_space(3))) void lds_void;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void NT(uint4 v, uint4 *p) { u32x4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (u32x4 *)p); }
template <int D, int WPE, bool WR = true, bool CMP = true, int ST = 0>
__global__ void __launch_bounds__(64, WPE) k_capture_dma(Args a)
{
    constexpr int SLOT = 1536;
    __shared__ __attribute__((aligned(16))) uint8_t ring[D * SLOT];
    const int f = blockIdx.x, lane = threadIdx.x;
    const uint8_t *frame = a.luma + (size_t)f * W * H;
    const uint32_t psm = ((708 - 12) * 128 + 66) / 132, hpsm = (psm + 1) / 2;
    const int x0 = (int)(((uint32_t)(lane + 3) * psm + hpsm) / 128) + 12, x1 = (int)(((uint32_t)(lane + 67) * psm + hpsm) / 128) + 12;
    uint32_t b0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint4 acc = {0, 0, 0, 0};
    uint4 hold[8] = {};
    constexpr int NP = H / 2;
    auto issue = [&](int k, int slot) {
        const uint8_t *src = frame + (size_t)k * (2 * W) + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (lds_void *)(ring + slot * SLOT), 16, 0, 0);
        if (lane < 26) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 1024), (lds_void *)(ring + slot * SLOT + 1024), 16, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; d++) issue(d, d);
    for (int k0 = 0; k0 < NP; k0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int k = k0 + d;
            if (k < NP) {
                // the oldest slot has landed when at most 2 * (D - 1) DMA pieces are still in flight
                if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (D == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (D == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (D == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                const uint8_t *slot = ring + d * SLOT;
                const uint8_t p0 = slot[x0], p1 = slot[x1], p2 = slot[W + x0], p3 = slot[W + x1];
                uint64_t aA_lo, bA_lo, aA_hi, bA_hi, aB_lo, bB_lo, aB_hi, bB_hi;
                if (CMP) { aA_lo = __ballot(p0 > a.lo); bA_lo = __ballot(p0 >= a.hi); aA_hi = __ballot(p1 > a.lo); bA_hi = __ballot(p1 >= a.hi);
                aB_lo = __ballot(p2 > a.lo); bB_lo = __ballot(p2 >= a.hi); aB_hi = __ballot(p3 > a.lo); bB_hi = __ballot(p3 >= a.hi); }
                else { acc.y ^= p0 ^ p1 ^ p2 ^ p3; aA_lo = bA_lo = aA_hi = bA_hi = aB_lo = bB_lo = aB_hi = bB_hi = 0; }
                const int kn = k + D < NP ? k + D : NP - 1;
                issue(kn, d);               // the compares above have consumed the slot's bytes
                const int j = (ST == 4 ? (k + f * 13) : k) & 63;
                if (CMP) {
                b0[0] = wl(b0[0], (uint32_t)aA_lo, j); b0[1] = wl(b0[1], (uint32_t)(aA_lo >> 32), j); b0[2] = wl(b0[2], (uint32_t)bA_lo, j); b0[3] = wl(b0[3], (uint32_t)(bA_lo >> 32), j);
                b0[4] = wl(b0[4], (uint32_t)aA_hi, j); b0[5] = wl(b0[5], (uint32_t)(aA_hi >> 32), j); b0[6] = wl(b0[6], (uint32_t)bA_hi, j); b0[7] = wl(b0[7], (uint32_t)(bA_hi >> 32), j);
                b1[0] = wl(b1[0], (uint32_t)aB_lo, j); b1[1] = wl(b1[1], (uint32_t)(aB_lo >> 32), j); b1[2] = wl(b1[2], (uint32_t)bB_lo, j); b1[3] = wl(b1[3], (uint32_t)(bB_lo >> 32), j);
                b1[4] = wl(b1[4], (uint32_t)aB_hi, j); b1[5] = wl(b1[5], (uint32_t)(aB_hi >> 32), j); b1[6] = wl(b1[6], (uint32_t)bB_hi, j); b1[7] = wl(b1[7], (uint32_t)(bB_hi >> 32), j); }
                if (j == 63 || k == NP - 1) {
                    uint4 r0 = {b0[0] ^ b0[4], b0[1] ^ b0[5], b0[2] ^ b0[6], b0[3] ^ b0[7]}, r1 = {b1[0] ^ b1[4], b1[1] ^ b1[5], b1[2] ^ b1[6], b1[3] ^ b1[7]};
                    uint4 *o = a.out + ((size_t)(ST == 6 ? (f & 31) : f) * 512 + (size_t)(k & ~63) * 2) * 3;
                    if (ST >= 7) {
                        /* round 5: lane-contiguous streaming stores; BPL bytes per line (16 / 32 / 48), a burst every 64 << BSH row pairs */
                        constexpr int BPL = ST == 10 ? 48 : (ST == 11 ? 32 : 16), BSH = ST == 8 ? 1 : (ST == 9 ? 2 : 0), NB = 1 << BSH;
                        const int chunk = k >> 6;
#pragma unroll
                        for (int u = 0; u < NB; u++) if ((chunk & (NB - 1)) == u) { hold[2 * u] = r0; hold[2 * u + 1] = r1; }
                        if (WR && ((chunk & (NB - 1)) == NB - 1 || k == NP - 1)) {
                            uint4 *ob = a.out + (size_t)f * 1536 + (size_t)(chunk & ~(NB - 1)) * 128 * (BPL / 16);
#pragma unroll
                            for (int u = 0; u < 2 * NB; u++)
#pragma unroll
                                for (int w = 0; w < BPL / 16; w++) NT(hold[u], &ob[(u * (BPL / 16) + w) * 64 + lane]);
                        }
                    } else
                    if (WR && lane <= j) {
                        if (ST == 5) { o[lane] = r0; o[64 + lane] = r1; }
                        else if (ST == 0 || ST == 4 || ST == 6) { o[lane * 3] = r0; o[lane * 3 + 1] = r1; o[lane * 3 + 2] = r0; o[(64 + lane) * 3] = r1; o[(64 + lane) * 3 + 1] = r0; o[(64 + lane) * 3 + 2] = r1; }
                        else if (ST == 1) { NT(r0, &o[lane * 3]); NT(r1, &o[lane * 3 + 1]); NT(r0, &o[lane * 3 + 2]);
                                            NT(r1, &o[(64 + lane) * 3]); NT(r0, &o[(64 + lane) * 3 + 1]); NT(r1, &o[(64 + lane) * 3 + 2]); }
                        else if (ST == 2) {     /* lane-contiguous: store i covers 1 KB */
                            o[lane] = r0; o[64 + lane] = r1; o[128 + lane] = r0; o[192 + lane] = r1; o[256 + lane] = r0; o[320 + lane] = r1; }
                        else {                  /* write-through */
                            uint4 *p = &o[lane * 3]; u32x4 t0 = {r0.x, r0.y, r0.z, r0.w}, t1 = {r1.x, r1.y, r1.z, r1.w};
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\tglobal_store_dwordx4 %0, %1, off offset:32 sc1" :: "v"(p), "v"(t0), "v"(t1) : "memory");
                            p = &o[(64 + lane) * 3];
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\tglobal_store_dwordx4 %0, %1, off offset:32 sc1" :: "v"(p), "v"(t1), "v"(t0) : "memory");
                        }
                    }
                    acc.x ^= r0.x ^ r1.x;
                }
            }
        }
    }
    if ((acc.x ^ acc.y) == 0x12345678u) a.out[0] = acc;
}

__global__ void __launch_bounds__(64, 5) k_write_only(Args a)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    uint4 v = {(uint32_t)f, (uint32_t)lane, a.lo, a.hi};
    uint4 *o = a.out + (size_t)f * 1536;
    for (int i = 0; i < 24; i++) o[i * 64 + lane] = v;      // 24 KB per frame
}
template <typename F> static float time_ms(F launch, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10000;
    uint8_t *d_luma; uint4 *d_out;
    const size_t bytes = (size_t)n * W * H;
    CK(hipMalloc(&d_luma, bytes + 4096)); CK(hipMalloc(&d_out, (size_t)n * 512 * 48 + 4096));
    {   // pseudo-random pixels
        std::vector<uint8_t> h(1 << 24); uint32_t s = 12345; for (auto &b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
        for (size_t o = 0; o < bytes; o += h.size()) CK(hipMemcpy(d_luma + o, h.data(), (bytes - o) < h.size() ? (bytes - o) : h.size(), hipMemcpyHostToDevice));
    }
    Args a{d_luma, d_out, n, 100, 130};
    const double gb = (double)bytes / 1e9, gbw = (double)n * 489 * 48 / 1e9;
#define RUN(name, kern) do { float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(n), dim3(64), 0, 0, a); }, 10); \
        printf("%-34s %7.3f ms  %6.2f TB/s read  (%6.2f TB/s read+write)\n", name, ms, gb / ms, (gb + gbw) / ms); } while (0)
    if (argc > 2 && atoi(argv[2]) == 5) {        /* round 5: what the write stream costs by bytes per line and by burst length */
        RUN("dma D=3 w5 no writes", (k_capture_dma<3, 5, false, true>));
        RUN("dma D=3 w5 plain 48 B (3 x 16 per lane)", (k_capture_dma<3, 5, true, true, 0>));
        RUN("dma D=3 w5 nt contiguous 48 B, burst 64", (k_capture_dma<3, 5, true, true, 10>));
        RUN("dma D=3 w5 nt contiguous 32 B, burst 64", (k_capture_dma<3, 5, true, true, 11>));
        RUN("dma D=3 w5 plain contiguous 16 B, burst 64", (k_capture_dma<3, 5, true, true, 5>));
        RUN("dma D=3 w5 nt contiguous 16 B, burst 64", (k_capture_dma<3, 5, true, true, 7>));
        RUN("dma D=3 w5 nt contiguous 16 B, burst 128", (k_capture_dma<3, 5, true, true, 8>));
        RUN("dma D=3 w5 nt contiguous 16 B, burst 256", (k_capture_dma<3, 5, true, true, 9>));
        RUN("dma D=3 w4 nt contiguous 16 B, burst 256", (k_capture_dma<3, 4, true, true, 9>));
        RUN("dma D=3 w5 no writes", (k_capture_dma<3, 5, false, true>));
        return 0;
    }
    RUN("write only 24 KB per frame", k_write_only);
    RUN("stream, 8 waves/SIMD", (k_stream<8>));
    RUN("stream, 5 waves/SIMD", (k_stream<5>));
    RUN("capture dual  D=4 w5 plain loads", (k_capture<4, true, 5, false>));
    RUN("capture dual  D=4 w5 NT loads", (k_capture<4, true, 5, true>));
    RUN("capture dual  D=4 w5 plain loads", (k_capture<4, true, 5, false>));
    RUN("capture dual  D=4 w5 NT loads", (k_capture<4, true, 5, true>));
    RUN("capture dual  D=2 w8", (k_capture<2, true, 8>));
    RUN("capture dual  D=4 w8", (k_capture<4, true, 8>));
    RUN("capture dual  D=8 w8", (k_capture<8, true, 8>));
    RUN("capture dual  D=4 w5", (k_capture<4, true, 5>));
    RUN("capture dual  D=8 w5", (k_capture<8, true, 5>));
    RUN("capture dual  D=8 w4", (k_capture<8, true, 4>));
    RUN("capture field D=4 w8", (k_capture<4, false, 8>));
    RUN("capture field D=8 w8", (k_capture<8, false, 8>));
    RUN("capture field D=8 w5", (k_capture<8, false, 5>));
    RUN("stream + 4KB writes per 64KB w5", (k_stream<5, true>));
    RUN("dma D=3 w5 no writes", (k_capture_dma<3, 5, false, true>));
    RUN("dma D=3 w5 no compares", (k_capture_dma<3, 5, true, false>));
    RUN("dma D=3 w5 no cmp no wr", (k_capture_dma<3, 5, false, false>));
    RUN("dma D=3 w5 nt stores", (k_capture_dma<3, 5, true, true, 1>));
    RUN("dma D=3 w5 lane-contiguous stores", (k_capture_dma<3, 5, true, true, 2>));
    RUN("dma D=3 w5 sc1 stores", (k_capture_dma<3, 5, true, true, 3>));
    RUN("dma D=3 w5 staggered flush", (k_capture_dma<3, 5, true, true, 4>));
    RUN("dma D=3 w5 16 B per line", (k_capture_dma<3, 5, true, true, 5>));
    RUN("dma D=3 w5 stores into 32 slots", (k_capture_dma<3, 5, true, true, 6>));
    RUN("dma D=3 w5 plain stores", (k_capture_dma<3, 5, true, true, 0>));
    RUN("capture dma   D=2 w8", (k_capture_dma<2, 8>));
    RUN("capture dma   D=3 w8", (k_capture_dma<3, 8>));
    RUN("capture dma   D=4 w8", (k_capture_dma<4, 8>));
    RUN("capture dma   D=2 w5", (k_capture_dma<2, 5>));
    RUN("capture dma   D=3 w5", (k_capture_dma<3, 5>));
    RUN("capture dma   D=4 w5", (k_capture_dma<4, 5>));
    RUN("capture dma   D=4 w4", (k_capture_dma<4, 4>));
    RUN("capture dma   D=6 w4", (k_capture_dma<6, 4>));
    return 0;
}
