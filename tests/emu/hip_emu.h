/*
 * hip_emu.h - a tiny single-wavefront SIMT emulator used ONLY by tests (and for local debugging in
 * the GPU-less build container): it lets the *same* HIP kernel source that hipcc compiles for gfx950
 * be compiled with g++ and executed lane by lane on the CPU, so kernel logic can be checked against
 * the oracle without a GPU.  It is test infrastructure, never a product fallback: the product library
 * (libsdvpcm_hip.so) contains no trace of it and fails loudly without a HIP device.
 *
 * Model: one workgroup = one wavefront of 64 lanes, each lane a ucontext fiber.  Wave collectives
 * (__ballot, __shfl, __syncthreads, readfirstlane) are rendezvous points: every lane must reach the
 * same call site (checked) - i.e. collectives must sit in wave-uniform control flow, which is also
 * the discipline the real kernels follow.  Workgroups run one after another.
 */
#ifndef SDV_HIP_EMU_H
#define SDV_HIP_EMU_H

#include <ucontext.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

namespace emu {
enum { WAVE = 64, STACK = 512 * 1024 };
enum Op { OP_NONE = 0, OP_BALLOT, OP_SHFL, OP_SYNC, OP_FIRST };

struct Lane {
    ucontext_t ctx;
    char *stack;
    bool done;
    int op, site;
    uint64_t arg, arg2, result;
};

struct State {
    Lane lanes[WAVE];
    ucontext_t sched;
    int cur;
    dim3 block_idx, grid_dim, block_dim;
    std::function<void()> body;
};

inline State &st() { static State s; return s; }

inline void lane_entry()
{
    State &s = st();
    s.body();
    s.lanes[s.cur].done = true;
    s.lanes[s.cur].op = OP_NONE;
    swapcontext(&s.lanes[s.cur].ctx, &s.sched);
}

inline uint64_t collective(int op, int site, uint64_t arg, uint64_t arg2 = 0)
{
    State &s = st();
    Lane &l = s.lanes[s.cur];
    l.op = op; l.site = site; l.arg = arg; l.arg2 = arg2;
    swapcontext(&l.ctx, &s.sched);
    return l.result;
}

inline void run_block(unsigned bx)
{
    State &s = st();
    s.block_idx = dim3(bx, 0, 0);
    for (int i = 0; i < WAVE; i++) {
        Lane &l = s.lanes[i];
        if (!l.stack) l.stack = (char *)malloc(STACK);
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack; l.ctx.uc_stack.ss_size = STACK; l.ctx.uc_link = &s.sched;
        l.done = false; l.op = OP_NONE;
        makecontext(&l.ctx, (void (*)())lane_entry, 0);
    }
    for (;;) {
        int ndone = 0;
        for (int i = 0; i < WAVE; i++) {
            if (s.lanes[i].done) { ndone++; continue; }
            s.cur = i;
            swapcontext(&s.sched, &s.lanes[i].ctx);
            if (s.lanes[i].done) ndone++;
        }
        if (ndone == WAVE) break;
        if (ndone != 0) { fprintf(stderr, "hip_emu: %d lanes exited while others wait at a collective (site %d)\n", ndone, s.lanes[0].site); abort(); }
        int op = s.lanes[0].op, site = s.lanes[0].site;
        for (int i = 1; i < WAVE; i++)
            if (s.lanes[i].op != op || s.lanes[i].site != site) {
                fprintf(stderr, "hip_emu: divergent collective: lane0 op %d line %d, lane %d op %d line %d\n", op, site, i, s.lanes[i].op, s.lanes[i].site);
                abort();
            }
        if (op == OP_BALLOT) {
            uint64_t m = 0;
            for (int i = 0; i < WAVE; i++) if (s.lanes[i].arg) m |= (1ull << i);
            for (int i = 0; i < WAVE; i++) s.lanes[i].result = m;
        } else if (op == OP_SHFL) {
            uint64_t v[WAVE];
            for (int i = 0; i < WAVE; i++) v[i] = s.lanes[i].arg;
            for (int i = 0; i < WAVE; i++) s.lanes[i].result = v[s.lanes[i].arg2 & 63];
        } else if (op == OP_FIRST) {
            for (int i = 0; i < WAVE; i++) s.lanes[i].result = s.lanes[0].arg;
        }
    }
}

template <class F> inline void launch(unsigned grid, F f)
{
    State &s = st();
    s.grid_dim = dim3(grid); s.block_dim = dim3(WAVE);
    s.body = f;
    for (unsigned b = 0; b < grid; b++) run_block(b);
}

struct TidProxy { unsigned x_() const { return (unsigned)st().cur; } };
struct Idx3 { unsigned x, y, z; };
inline Idx3 tidx() { return Idx3{ (unsigned)st().cur, 0, 0 }; }
inline Idx3 bidx() { return Idx3{ st().block_idx.x, 0, 0 }; }
inline Idx3 bdim() { return Idx3{ (unsigned)WAVE, 1, 1 }; }
inline Idx3 gdim() { return Idx3{ st().grid_dim.x, 1, 1 }; }
} // namespace emu

#define threadIdx (emu::tidx())
#define blockIdx (emu::bidx())
#define blockDim (emu::bdim())
#define gridDim (emu::gdim())

#define __ballot(p) emu::collective(emu::OP_BALLOT, __LINE__, (uint64_t)((p) ? 1 : 0))
#define __shfl(v, src) ((int)(uint32_t)emu::collective(emu::OP_SHFL, __LINE__, (uint64_t)(uint32_t)(v), (uint64_t)(src)))
#define __syncthreads() ((void)emu::collective(emu::OP_SYNC, __LINE__, 0))
#define __builtin_amdgcn_readlane(v, l) ((int)(uint32_t)emu::collective(emu::OP_SHFL, __LINE__, (uint64_t)(uint32_t)(v), (uint64_t)(l)))
#define __builtin_amdgcn_readfirstlane(v) ((int)(uint32_t)emu::collective(emu::OP_FIRST, __LINE__, (uint64_t)(uint32_t)(v)))

static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline unsigned __brev(unsigned v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
static inline unsigned long long __brevll(unsigned long long v) { return ((unsigned long long)__brev((unsigned)v) << 32) | __brev((unsigned)(v >> 32)); }
template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }

#endif
