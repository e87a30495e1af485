// Data-parallel gradient exchange INSIDE the launches that already exist.
//
// The reference is one process and has no collective; the data-parallel form of this package sums the gradient slice of every optimizer
// step over the ranks between `loss.backward()` and `optimizer.step()` (reference call sites: agent/vlsac/vlsac_agent.py:153-154, 183-184,
// 229-230 and siblings).  Those slices are 0.3 - 2 MB and there are seven of them per train(): latency-bound.  Round 4's one-shot form cost
// three dependent launches per all-reduce (push, signal + wait, reduce) on chains whose whole problem is launch count.  Here it costs none:
//
//   * every rank's GRADIENT ARENA lives in a block of device memory that is exported over hipIpc and mapped by every peer (fine-grained
//     memory between GPUs: peers read it over xGMI behind the owner's L2; plain hipMalloc between processes that share one GPU);
//   * the gradient-producing launches are untouched: they write the local arena as on one GPU;
//   * the group's OPTIMIZER launch (elementwise.hip adam_kernel) does the exchange itself: its first block tells every peer "my gradients
//     of epoch e are complete" (they are: the launches that wrote them precede this one in stream order), every optimizer block waits -- bounded --
//     for all peers' READY words, then reads the gradient of its elements from every rank's arena IN RANK ORDER (own rank: the local arena,
//     peers: system-scope loads through the mapped pointers) and sums them where it used to load one gradient: the sums, and therefore the
//     replicas, are bit-identical on every rank; no float atomics, no broadcast;
//   * the last block of the launch to finish (ticket) tells every peer "I have read your epoch e" and waits -- bounded -- for the same word
//     from every peer before the launch ends: nothing that follows in stream order (the next backward) can overwrite gradients a peer is
//     still reading;
//   * the epoch is a DEVICE counter per channel (one channel per optimizer group), advanced by that last block: nothing on the host
//     changes between two calls, so the whole train() is capturable into the same hipGraphs as on one GPU.
//
// A wait that does not complete within `spins` polls sets the rank's bit in the error word and PROCEEDS (the launch always drains; the
// step is then wrong and rlrep_comm_status / the agent's flush() raise).  Deadlock freedom: READY(e) is sent before anything is waited
// for; DONE(e) is sent after a rank's own reads, which need only the peers' READY(e).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RL_DP_MAX_WORLD 16
#define RL_DP_CHANNELS 8                /* 0-3: the optimizer groups; 7: rlrep_comm_allreduce (probe / tests) */

// one per rank, behind its arena in the shared block (zeroed at creation)
struct DpFlags {
    unsigned ready[RL_DP_CHANNELS][RL_DP_MAX_WORLD];       // ready[c][q]: written by rank q -- "my data of channel c is complete for epoch e"
    unsigned done[RL_DP_CHANNELS][RL_DP_MAX_WORLD];        // done[c][q]:  written by rank q -- "I have read YOUR data of channel c, epoch e"
    unsigned epoch[RL_DP_CHANNELS];                        // local: last epoch completed on this rank
    unsigned ticket[RL_DP_CHANNELS];                       // local: blocks of the running launch that have finished
    unsigned pad_[64];
};

struct DpPull {
    int world, rank, channel, nblocks;                     // nblocks: the blocks of the launch that take a ticket
    long long spins;                                       // poll bound per wait
    long long tail_off, tail_n;                            // (optimizer launches) arena-relative range that the trailing block sums too (the temperature gradient's partials); tail_n = 0: none
    unsigned* err;                                         // error word in mapped HOST memory (bit q = a wait for rank q timed out): the host polls it without a device sync
    const float* base[RL_DP_MAX_WORLD];                    // rank q's arena as mapped here (base[rank] = the local arena)
    DpFlags* flags[RL_DP_MAX_WORLD];                       // rank q's flag block as mapped here
};

#ifdef __HIPCC__
typedef float dp_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned dp_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool dp_reached(unsigned v, unsigned e) { return (int)(v - e) >= 0; }         // epochs wrap: signed distance

// 16 / 4 bytes of a PEER's arena: system-scope loads (sc0 sc1: never served from this GPU's caches)
__device__ __forceinline__ dp_f32x4 dp_load4(const float* p) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x27000);
    const dp_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 17);
    return (dp_f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
__device__ __forceinline__ float dp_load1(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

// Called by EVERY thread of a participating block (256 threads, or one wave: `bar` = the block has more than one wave).  `signaller`: the
// block that publishes READY.  Returns the epoch of this launch.
__device__ __forceinline__ unsigned dp_begin(const DpPull& d, bool signaller, bool bar) {
    DpFlags* const mine = d.flags[d.rank];
    const unsigned e = __hip_atomic_load(&mine->epoch[d.channel], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int q = threadIdx.x;
    if (threadIdx.x < 64) {
        const bool peer = q < d.world && q != d.rank;
        if (signaller && peer) {
            __threadfence_system();
            __hip_atomic_store(&d.flags[q]->ready[d.channel][d.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        long long s = 0;
        bool ok = !peer;
        while (true) {
            if (!ok) ok = dp_reached(__hip_atomic_load(&mine->ready[d.channel][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), e);
            if (__all(ok) || ++s > d.spins) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (!ok) atomicOr(d.err, 1u << (q & 15));                       // never hang the GPU: report, proceed, drain
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                   // system scope: nothing read after this comes from a cache line of an earlier epoch
    }
    if (bar) __syncthreads();
    return e;
}

// Called by every thread of a participating block once its reads of the peers' arenas have been consumed.
__device__ __forceinline__ void dp_end(const DpPull& d, unsigned e, bool bar) {
    __shared__ int dp_last_s;
    DpFlags* const mine = d.flags[d.rank];
    if (bar) __syncthreads();
    if (threadIdx.x == 0) dp_last_s = (atomicAdd(&mine->ticket[d.channel], 1u) == (unsigned)(d.nblocks - 1)) ? 1 : 0;
    if (bar) __syncthreads();
    if (!dp_last_s || threadIdx.x >= 64) return;
    const int q = threadIdx.x;
    const bool peer = q < d.world && q != d.rank;
    if (peer) __hip_atomic_store(&d.flags[q]->done[d.channel][d.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    long long s = 0;
    bool ok = !peer;
    while (true) {
        if (!ok) ok = dp_reached(__hip_atomic_load(&mine->done[d.channel][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), e);
        if (__all(ok) || ++s > d.spins) break;
        __builtin_amdgcn_s_sleep(4);
    }
    if (!ok) atomicOr(d.err, 1u << (q & 15));
    if (q == 0) {
        __hip_atomic_store(&mine->ticket[d.channel], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->epoch[d.channel], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// sum over the ranks, in rank order, of the four floats at arena offset `off` (16-byte aligned on every rank: the arenas share one layout).
// Branch-free: eight ranks' loads in flight together (a slot beyond the world re-reads the last rank and is not added); the own arena is read
// through the same system-scope path.
__device__ __forceinline__ dp_f32x4 dp_sum4(const DpPull& d, long long off) {
    dp_f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < d.world; c += 8) {
        dp_f32x4 part[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) part[j] = dp_load4(d.base[min(c + j, d.world - 1)] + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (c + j < d.world) s = (c + j) ? s + part[j] : part[j];
    }
    return s;
}
__device__ __forceinline__ float dp_sum1(const DpPull& d, long long off) {
    float s = 0.f;
    for (int q = 0; q < d.world; ++q) { const float x = dp_load1(d.base[q] + off); s = q ? s + x : x; }
    return s;
}
#endif
