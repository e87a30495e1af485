// Data-parallel exchanges INSIDE the launches that already exist.
//
// The reference is one process and has no collective; the data-parallel form of this package sums the gradient slice of every optimizer
// step over the ranks between `loss.backward()` and `optimizer.step()` (reference call sites: agent/vlsac/vlsac_agent.py:153-154, 183-184,
// 229-230 and siblings).  Those slices are 0.3 - 2 MB and there are seven of them per train(): latency-bound.  Round 4's one-shot form cost
// three dependent launches per all-reduce (push, signal + wait, reduce) on chains whose whole problem is launch count.  Here it costs none:
//
//   * every rank owns ONE block of device memory, [gradient arena | exchange scratch | reduced region | flag words], exported over hipIpc and
//     mapped by every peer (fine-grained memory between GPUs: peers read it over xGMI behind the owner's L2; plain hipMalloc between
//     processes that share one GPU; plain pointers between two comms of one process: the loopback form of tools/exp/dp_loopback.py);
//   * the gradient-producing launches are untouched: they write the local arena as on one GPU;
//   * the group's OPTIMIZER launch (elementwise.hip adam_dp_kernel) does the exchange itself: its first block tells every peer "my gradients
//     of epoch e are complete" (they are: the launches that wrote them precede this one in stream order), every optimizer block waits -- bounded --
//     for all peers' READY words, then
//       ONE-SHOT (world 2, small slices, tails): reads the gradient of its elements from every rank's arena IN RANK ORDER (own rank: the
//         local arena, peers: system-scope loads through the mapped pointers) and sums them where it used to load one gradient:
//         (N - 1) S bytes inbound per GPU;
//       TWO-SHOT (world >= 3 and a slice above ~0.5 MB): the slice is cut into N shards; the launch of rank r first sums shard r
//         over all ranks in rank order into the REDUCED region of its own block (reduce-scatter: (N - 1) S / N inbound), the last of the blocks
//         that did so raises RED(e) at every peer, then every block waits for all RED words and reads the sum of its elements from the
//         shard's owner (all-gather: (N - 1) S / N inbound) -- same launch, same epochs, same error word;
//     either way the sums, and therefore the replicas, are bit-identical on every rank; no float atomics, no broadcast;
//   * the last block of the launch to finish (ticket) tells every peer "I have read your epoch e" and waits -- bounded -- for the same word
//     from every peer before the launch ends: nothing that follows in stream order (the next backward) can overwrite gradients, or reduced
//     shards, that a peer is still reading;
//   * the epoch is a DEVICE counter per channel (one channel per optimizer group), advanced by that last block: nothing on the host
//     changes between two calls, so the whole train() is capturable into the same hipGraphs as on one GPU.
//
// BATCH-COUPLED exchanges of the feature steps (spedersac's Phibar and v, F floats each: agent/spedersac/spedersac_agent.py:197-205; ctrlsac's
// all-gather of mu(s') and reduce-scatter of its gradient: agent/ctrlsac/ctrlsac_agent.py:226-231) use the same block and flags:
//   * DpSlots (PUSH, small vectors): the PRODUCER launch stores its partial into a slot of every rank's scratch ([2 parities][world][n],
//     indexed by the epoch's parity and the producer's rank), its last block raises READY(e) everywhere; the CONSUMER launch -- the next one
//     in stream order that reads the vector -- waits for all READY words and sums the N slots of its OWN block in rank order.  No DONE word:
//     a rank can only push epoch e + 2 after it has consumed e + 1, which needs every peer's push of e + 1, which follows that peer's
//     consumption of e in ITS stream order -- two parities are enough.  Zero extra launches.
//   * ctrlsac's two exchanges are stand-alone launches of comm.hip (push gather / pull reduce-scatter), one each per feature step, capturable.
//
// A wait that does not complete within `timeout` ticks of the 100 MHz wall clock (default two minutes: a watchdog, not a schedule -- a peer
// that evaluates, checkpoints or re-captures a graph simply delays the step, as under RCCL) sets the rank's bit in the error word AND the
// rank's local `poison` word; the launch that saw it, and EVERY later exchange launch of that rank until the host has read and cleared the
// error (rlrep_comm_status), skips its work -- nothing is ever applied from a partial or stale sum -- all launches drain, and
// rlrep_comm_status / the agent's flush() raise.  Deadlock freedom on ONE channel: READY(e) is sent before anything is waited for; RED(e)
// needs only the peers' READY(e); DONE(e) is sent after a rank's own reads, which need only the peers' READY(e) / RED(e).
//
// Progress with SEVERAL channels in flight (the two-chain train(): the feature group's optimizer launch on one stream, the critic's or the
// actor's on the other): rank 0 may sit in channel 0's launch while rank 1 sits in channel 1's, every block of both spinning.  The
// launches they wait for -- rank 0's channel-1 launch, rank 1's channel-0 launch -- are on the OTHER stream of their rank and must become
// resident BESIDE the spinning one.  They do if (1) the two streams map to different hardware queues (GPU_MAX_HW_QUEUES >= 2: HIP's default
// is 4) and (2) the blocks of two optimizer launches fit the chip together: rl_agent_attach_dp asserts
// max-two-groups (blocks + riders) <= CUs x occupancy(adam_dp_kernel) and leaves the largest groups unattached until it holds.  Launches of other kinds that
// share the chip with a spinning one finish on their own (they wait for nothing).  Within a two-shot launch a block spins on RED words that
// depend on phase-A blocks of EVERY rank; phase-A blocks are the lowest-numbered ones and are dispatched first, and a launch that fits the
// chip as a whole (asserted) has all of them resident.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RL_DP_MAX_WORLD 16
#define RL_DP_CHANNELS 8                /* 0-3: the optimizer groups; 4, 5: the feature step's batch-coupled exchanges; 6: rlrep_comm_probe_slots; 7: rlrep_comm_allreduce / _allgather (probe / tests) */
#define RL_DP_TICKS_PER_US 100ll        /* wall_clock64(): the 100 MHz constant clock */
static_assert(RL_DP_MAX_WORLD <= 32, "the late-rank mask is one 32-bit word (1u << rank)");

// one per rank, behind its arena / scratch / reduced region in the shared block (zeroed at creation)
struct DpFlags {
    unsigned ready[RL_DP_CHANNELS][RL_DP_MAX_WORLD];       // ready[c][q]: written by rank q -- "my data of channel c is complete for epoch e"
    unsigned done[RL_DP_CHANNELS][RL_DP_MAX_WORLD];        // done[c][q]:  written by rank q -- "I have read YOUR data of channel c, epoch e"
    unsigned red[RL_DP_CHANNELS][RL_DP_MAX_WORLD];         // red[c][q]:   written by rank q -- "my shard of channel c, epoch e, is summed" (two-shot)
    unsigned epoch[RL_DP_CHANNELS];                        // local: last epoch completed on this rank
    unsigned ticket[RL_DP_CHANNELS];                       // local: blocks of the running launch that have finished
    unsigned ticket2[RL_DP_CHANNELS];                      // local: phase-A blocks of the running two-shot launch that have stored their sums
    unsigned poison;                                       // local: a wait of THIS rank has run out since the host last cleared the error word: every later exchange
                                                           // launch of this rank skips its work (a replica that is out of step applies nothing more until the host has seen it)
    unsigned pad_[39];
};

struct DpPull {
    int world, rank, channel, nblocks;                     // nblocks: the blocks of the launch that take a ticket
    int mode, nblocks_a;                                   // mode 2: two-shot (nblocks_a = the blocks that sum a piece of this rank's shard)
    int no_done;                                           // the launch ends without the DONE handshake: the caller's program guarantees that this rank's data are
                                                           // not overwritten before every peer has passed a LATER handshake (ctrlsac's alternating gather / reduce)
    long long timeout;                                     // bound of every wait, ticks of the 100 MHz wall clock
    long long tail_off, tail_n;                            // (optimizer launches) arena-relative range that the trailing block sums too (the temperature gradient's partials); tail_n = 0: none
    long long shard4;                                      // two-shot: 16-byte elements per shard (the slice's elements / world, rounded up)
    unsigned* err;                                         // error word in mapped HOST memory (bit q = a wait for rank q timed out): the host polls it without a device sync
    const float* base[RL_DP_MAX_WORLD];                    // rank q's arena as mapped here (base[rank] = the local arena)
    float* red[RL_DP_MAX_WORLD];                           // rank q's reduced region as mapped here (same offsets as the arena)
    DpFlags* flags[RL_DP_MAX_WORLD];                       // rank q's flag block as mapped here
};

// small vectors exchanged by PUSH (see the head of this file): slot[q] = rank q's slot area of this channel as mapped here, [2][world][n]
struct DpSlots {
    int world, rank, channel, n;
    int nblocks;                                           // producer: the blocks that take a ticket
    long long timeout;
    unsigned* err;
    float* slot[RL_DP_MAX_WORLD];
    DpFlags* flags[RL_DP_MAX_WORLD];
};

// what rlrep_comm_attach (comm.hip) hands to the agent (engine.hip rl_agent_attach_dp)
struct DpAttach {
    DpPull proto;                                          // world, rank, timeout, err, base / red / flags of every rank
    float* scratch[RL_DP_MAX_WORLD];                       // rank q's exchange scratch as mapped here
    long long arena_floats, scratch_floats, max_floats, two_shot_floats;
};

#ifdef __HIPCC__
typedef float dp_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned dp_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool dp_reached(unsigned v, unsigned e) { return (int)(v - e) >= 0; }         // epochs wrap: signed distance

// 16 / 4 bytes of a PEER's block: system-scope loads (sc0 sc1: never served from this GPU's caches)
__device__ __forceinline__ dp_f32x4 dp_load4(const float* p) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x27000);
    const dp_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 17);
    return (dp_f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
__device__ __forceinline__ float dp_load1(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
// write-through stores (sc0 sc1) of data a peer will read: nothing is left dirty in this XCD's L2 behind the release that follows
__device__ __forceinline__ void dp_store4(float* p, dp_f32x4 v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x27000);
    __builtin_amdgcn_raw_buffer_store_b128((dp_u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, r, 0, 0, 17);
}
__device__ __forceinline__ void dp_store1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Ordering WITHOUT cache maintenance.  Everything a peer reads of this GPU was either written by an EARLIER kernel (the kernel boundary has
// written the XCDs' L2s back) or by write-through stores of this launch (dp_store*: sc0 sc1), and everything read of a peer goes past this GPU's
// caches (dp_load*: sc0 sc1).  So a signal needs only "my stores have left" (vmcnt(0)) in front of it, and a wait only a compiler barrier behind
// it -- a system-scope release / acquire FENCE here is a write-back / invalidate of the L2 by every block of the launch: the optimizer launch
// measured 20.2 us against 7.2 us single-GPU with them, the other chain's kernels losing their L2 lines on top (docs/history/r06.md).
__device__ __forceinline__ void dp_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); }      // vmcnt(0): this wave's stores have been acknowledged
__device__ __forceinline__ void dp_signal(unsigned* word, unsigned e) { __hip_atomic_store(word, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// the first wave of a block waits until word[q] of every peer q has reached epoch e; returns false (and reports the late ranks) on a timeout
__device__ __forceinline__ bool dp_wait_all(const unsigned* words, int world, int rank, unsigned e, long long timeout, unsigned* err, unsigned* poison) {
    const int q = threadIdx.x & 63;
    const bool peer = q < world && q != rank;
    bool ok = !peer;
    const unsigned poisoned = *poison;                 // (a plain load, issued in front of the poll: set by an EARLIER launch's timeout; costs no round trip of its own)
    const long long t0 = wall_clock64();
    while (true) {
        if (!ok) ok = dp_reached(__hip_atomic_load(words + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), e);
        if (__all(ok)) break;
        if (wall_clock64() - t0 > timeout) break;
        __builtin_amdgcn_s_sleep(4);
    }
    if (!ok) {                                                          // never hang the GPU: report, skip, drain
        atomicOr(err, 1u << q);
        __hip_atomic_store(poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::: "memory");                                     // (the reads that follow are issued behind the poll that saw the word; they bypass the caches)
    return __all(ok) && poisoned == 0u;
}

// Called by EVERY thread of a participating block (256 threads, or one wave: `bar` = the block has more than one wave).  `signaller`: the
// block that publishes READY.  Returns the epoch of this launch; *good = no wait of this block has timed out.
__device__ __forceinline__ unsigned dp_begin(const DpPull& d, bool signaller, bool bar, bool* good) {
    __shared__ int dp_good_s;
    DpFlags* const mine = d.flags[d.rank];
    // (a plain load: the word was written by the last block of the PREVIOUS launch of this channel -- an earlier kernel of the same stream; a
    //  coherent load here was a cache-bypassing round trip at the head of every block's critical path)
    const unsigned e = mine->epoch[d.channel] + 1u;
    if (threadIdx.x < 64) {
        const int q = threadIdx.x;
        if (signaller && q < d.world && q != d.rank) dp_signal(&d.flags[q]->ready[d.channel][d.rank], e);      // (the data were written by earlier kernels)
        const bool ok = dp_wait_all(mine->ready[d.channel], d.world, d.rank, e, d.timeout, d.err, &mine->poison);
        if (threadIdx.x == 0) dp_good_s = ok ? 1 : 0;
        if (!bar) *good = ok;
    }
    if (bar) { __syncthreads(); *good = dp_good_s != 0; }
    return e;
}

// Called by every thread of a participating block once its reads of the peers' blocks have been consumed.
__device__ __forceinline__ void dp_end(const DpPull& d, unsigned e, bool bar) {
    __shared__ int dp_last_s;
    DpFlags* const mine = d.flags[d.rank];
    if (bar) __syncthreads();
    if (threadIdx.x == 0) dp_last_s = (atomicAdd(&mine->ticket[d.channel], 1u) == (unsigned)(d.nblocks - 1)) ? 1 : 0;
    if (bar) __syncthreads();
    if (!dp_last_s || threadIdx.x >= 64) return;
    const int q = threadIdx.x;
    if (!d.no_done) {
        if (q < d.world && q != d.rank) dp_signal(&d.flags[q]->done[d.channel][d.rank], e);       // (this rank's reads were consumed before its blocks took their tickets)
        (void)dp_wait_all(mine->done[d.channel], d.world, d.rank, e, d.timeout, d.err, &mine->poison);
    }
    if (q == 0) {
        __hip_atomic_store(&mine->ticket[d.channel], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->ticket2[d.channel], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->epoch[d.channel], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (read by LATER kernels only)
    }
}

// sum over the ranks, in rank order, of the four floats at block offset `off` (16-byte aligned on every rank: the blocks share one layout).
// Exactly `world` loads, all in flight together (wave-uniform switch: a world of 2 issues two loads, not eight); the own arena is read
// through the same system-scope path.
template <int W>
__device__ __forceinline__ dp_f32x4 dp_sum4_w(const float* const* base, long long off) {
    dp_f32x4 part[W];
#pragma unroll
    for (int j = 0; j < W; ++j) part[j] = dp_load4(base[j] + off);
    dp_f32x4 s = part[0];
#pragma unroll
    for (int j = 1; j < W; ++j) s = s + part[j];
    return s;
}
__device__ __forceinline__ dp_f32x4 dp_sum4(const DpPull& d, long long off) {
    switch (d.world) {
    case 2: return dp_sum4_w<2>(d.base, off);
    case 3: return dp_sum4_w<3>(d.base, off);
    case 4: return dp_sum4_w<4>(d.base, off);
    case 8: return dp_sum4_w<8>(d.base, off);
    default: break;
    }
    dp_f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < d.world; c += 4) {
        dp_f32x4 part[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (c + j < d.world) part[j] = dp_load4(d.base[c + j] + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (c + j < d.world) s = (c + j) ? s + part[j] : part[j];
    }
    return s;
}
__device__ __forceinline__ float dp_sum1(const DpPull& d, long long off) {
    float s = 0.f;
    for (int q = 0; q < d.world; ++q) { const float x = dp_load1(d.base[q] + off); s = q ? s + x : x; }
    return s;
}

// TWO-SHOT, phase A + the RED handshake.  Called by every thread of every optimizer block (256 threads) behind dp_begin: block `bid` sums its
// piece of THIS rank's shard of the slice [goff, goff + 4 n4) into the reduced region (nblocks_a blocks have a piece), the last of them
// raises RED(e) at every peer, and every block then waits for all RED words.  Returns false on a timeout.
__device__ __forceinline__ bool dp_reduce_scatter(const DpPull& d, unsigned e, int bid, long long goff, long long n4, bool good) {
    __shared__ int dp_rs_s;
    DpFlags* const mine = d.flags[d.rank];
    if (bid < d.nblocks_a) {
        const long long j = (long long)bid * 256 + threadIdx.x, i4 = (long long)d.rank * d.shard4 + j;
        if (good && j < d.shard4 && i4 < n4) dp_store4(d.red[d.rank] + goff + 4 * i4, dp_sum4(d, goff + 4 * i4));
        dp_drain();                                        // the write-through stores of this thread have reached memory ...
        __syncthreads();                                   // ... and those of the whole block
        if (threadIdx.x == 0) dp_rs_s = (atomicAdd(&mine->ticket2[d.channel], 1u) == (unsigned)(d.nblocks_a - 1)) ? 1 : 0;
        __syncthreads();
        if (dp_rs_s && threadIdx.x < 64) {
            const int q = threadIdx.x;
            if (q < d.world) dp_signal(&d.flags[q]->red[d.channel][d.rank], e);      // (own word too: the local blocks wait on it)
        }
    }
    if (threadIdx.x < 64) {
        // every rank's word, the own one included: rank -1 = "nobody is exempt"
        const bool ok = dp_wait_all(mine->red[d.channel], d.world, -1, e, d.timeout, d.err, &mine->poison);
        if (threadIdx.x == 0) dp_rs_s = ok ? 1 : 0;
    }
    __syncthreads();
    return dp_rs_s != 0;
}
// phase B: the summed gradient of the four floats at slice-relative 16-byte element i4, from the shard's owner
__device__ __forceinline__ dp_f32x4 dp_gather4(const DpPull& d, long long goff, long long i4) {
    const int owner = (int)(i4 / d.shard4);
    return dp_load4(d.red[owner] + goff + 4 * i4);
}

// ---- DpSlots: producer side.  Every thread that owns element f of the partial calls dp_slots_put; then EVERY thread of every ticket-taking
// block calls dp_slots_publish (256 or 1024 threads per block; contains barriers).
__device__ __forceinline__ unsigned dp_slots_epoch(const DpSlots& d) {       // the epoch this producer launch publishes (read before any ticket is taken)
    return d.flags[d.rank]->epoch[d.channel] + 1u;
}
__device__ __forceinline__ void dp_slots_put(const DpSlots& d, unsigned e, int f, float v) {
    const size_t at = ((size_t)(e & 1u) * d.world + d.rank) * d.n + f;
    for (int q = 0; q < d.world; ++q) dp_store1(d.slot[q] + at, v);
}
__device__ __forceinline__ void dp_slots_publish(const DpSlots& d, unsigned e) {
    __shared__ int dp_sl_s;
    DpFlags* const mine = d.flags[d.rank];
    dp_drain();                                            // this thread's pushed (write-through) stores have been acknowledged ...
    __syncthreads();                                       // ... and the block's
    if (threadIdx.x == 0) dp_sl_s = (atomicAdd(&mine->ticket[d.channel], 1u) == (unsigned)(d.nblocks - 1)) ? 1 : 0;
    __syncthreads();
    if (!dp_sl_s || threadIdx.x >= 64) return;
    const int q = threadIdx.x;
    if (q < d.world) dp_signal(&d.flags[q]->ready[d.channel][d.rank], e);
    if (q == 0) {
        __hip_atomic_store(&mine->ticket[d.channel], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->epoch[d.channel], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// ---- consumer side (a LATER launch of the same stream).  Every thread of the block (any size that is a multiple of 64): waits for all ranks'
// READY of the epoch the producer has just published, then sums the N slots of this rank's own block, in rank order, into dst[0 .. n)
// (LDS or global).  Contains barriers.
__device__ __forceinline__ void dp_slots_sum(const DpSlots& d, float* dst) {
    DpFlags* const mine = d.flags[d.rank];
    const unsigned e = mine->epoch[d.channel];            // (written by the producer launch: an earlier kernel of this stream)
    __shared__ int dp_ss_s;
    if (threadIdx.x < 64) {
        const bool ok = dp_wait_all(mine->ready[d.channel], d.world, -1, e, d.timeout, d.err, &mine->poison);
        if (threadIdx.x == 0) dp_ss_s = ok ? 1 : 0;
    }
    __syncthreads();
    if (dp_ss_s) {                                         // (a timed-out or poisoned rank leaves the vector as it is: the optimizer launches behind it skip too)
        const float* sl = d.slot[d.rank] + (size_t)(e & 1u) * d.world * d.n;
        for (int f = threadIdx.x; f < d.n; f += blockDim.x) {
            float s = dp_load1(sl + f);
            for (int q = 1; q < d.world; ++q) s += dp_load1(sl + (size_t)q * d.n + f);
            dst[f] = s;
        }
    }
    __syncthreads();
}
#endif
