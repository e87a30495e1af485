// The shared block of the data-parallel form: allocation, hipIpc export / mapping (or plain pointers between comms of
// one process: the loopback form), attachment to an agent, the stand-alone exchanges (probe, tests, ctrlsac's two batch-coupled exchanges) and
// the error word.  The gradient exchange itself lives in dp_pull.h and runs INSIDE the optimizer launches (elementwise.hip adam_dp_kernel): zero
// launches per all-reduce.  The reference has no collective (one process); the call sites the exchange belongs to are its backward() -> step()
// pairs: agent/vlsac/vlsac_agent.py:153-154, 183-184, 229-230, agent/ctrlsac/ctrlsac_agent.py:243-244, agent/spedersac/spedersac_agent.py:211-212,
// agent/diffsrsac/diffsrsac_agent.py:311-314; the batch-coupled ones: agent/spedersac/spedersac_agent.py:197-205, agent/ctrlsac/ctrlsac_agent.py:226-231.
//
//     block  = [ arena_floats (the caller's gradient arena) | scratch_floats (batch-coupled exchanges) | reduced region (two-shot; world >= 3) | DpFlags ]
//              one allocation, ONE hipIpc handle
//
// Fine-grained device memory when the runtime can export it (peers on other GPUs read it over xGMI behind this GPU's L2); plain hipMalloc
// otherwise -- correct between processes that share ONE GPU (the tests), refused by rlrep_amd/comm.py when the ranks sit on different devices.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/rlrep.h"
#include "dp_pull.h"

void rl_set_error(const char* fmt, ...);
extern long long g_rl_launches;

struct rlrep_comm {
    int rank = 0, world = 1;
    long long arena_floats = 0, scratch_floats = 0, red_floats = 0;
    long long ticks_per_us = RL_DP_TICKS_PER_US;   // wall_clock64() rate of THIS device (hipDeviceAttributeWallClockRate; 100 MHz on gfx950)
    long long timeout_ticks = 120ll * 1000000ll * RL_DP_TICKS_PER_US;       // two minutes: a watchdog (rlrep_comm_set_timeout)
    char* local = nullptr;                         // my block
    char* peer[RL_DP_MAX_WORLD] = {nullptr};       // everyone's block as mapped here (peer[rank] = local)
    bool opened[RL_DP_MAX_WORLD] = {false};
    unsigned* err_host = nullptr;                  // error word: pinned host memory, mapped into the device's address space
    unsigned* err_dev = nullptr;
    size_t scratch_off = 0, red_off = 0, flags_off = 0, bytes = 0;
    bool fine_grained = false, connected = false;
};

// rlrep_agent side (engine.hip)
extern "C" int rl_agent_attach_dp(rlrep_agent* ag, const DpAttach* at, int* attached_mask);

extern "C" void rl_comm_fill_pull(const rlrep_comm* c, DpPull* d) {
    memset(d, 0, sizeof(*d));
    d->world = c->world; d->rank = c->rank; d->timeout = c->timeout_ticks; d->err = c->err_dev; d->mode = 1;
    for (int q = 0; q < c->world; ++q) {
        d->base[q] = reinterpret_cast<const float*>(c->peer[q]);
        d->red[q] = c->red_floats ? reinterpret_cast<float*>(c->peer[q] + c->red_off) : nullptr;
        d->flags[q] = reinterpret_cast<DpFlags*>(c->peer[q] + c->flags_off);
    }
}

// out[i] = sum over the ranks, in rank order, of block_q[off + i]: ONE launch; every block waits for the peers' READY, the last one to finish
// runs the DONE handshake (dp_pull.h).  The probe of rlrep_amd/comm.py, tests/test_comm.py and ctrlsac's reduce-scatter of dmu' (out = this
// rank's own rows, in place: peers read only THEIR rows of this block); the optimizer launches do the same with the gradient of their own elements.
template <int MODE>
__global__ __launch_bounds__(256) void comm_pull_kernel(DpPull d, long long off, long long n, float* __restrict__ out) {
    bool good = true;
    const unsigned e = dp_begin(d, blockIdx.x == 0, true, &good);
    const long long n4 = n >> 2;
    if constexpr (MODE == 2) {
        // two-shot (the launcher guarantees off % 4 == 0, n % 4 == 0, out 16-byte aligned, gridDim.x * 256 >= n4)
        good = dp_reduce_scatter(d, e, blockIdx.x, off, n4, good) && good;
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        if (good && i < n4) reinterpret_cast<dp_f32x4*>(out)[i] = dp_gather4(d, off, i);
    } else if (good) {
        if ((off & 3) == 0 && (((uintptr_t)out) & 15) == 0) {
            for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
                reinterpret_cast<dp_f32x4*>(out)[i] = dp_sum4(d, off + 4 * i);
            for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = dp_sum1(d, off + i);
        } else {
            for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = dp_sum1(d, off + i);
        }
    }
    dp_end(d, e, true);
}

// all-gather by PULL: buffer [world][n] at block offset `off` on every rank; rank q's own segment q is complete when its launch starts (stream
// order); every rank copies every peer's segment into its own buffer (system-scope loads, plain local stores: what later launches read was
// written locally).  ctrlsac's mu(s') of every rank's minibatch (agent/ctrlsac/ctrlsac_agent.py:226-231 with in-batch negatives over the global batch).
__global__ __launch_bounds__(256) void comm_gather_kernel(DpPull d, long long off, long long n) {
    bool good = true;
    const unsigned e = dp_begin(d, blockIdx.x == 0, true, &good);
    if (good) {
        float* mine = const_cast<float*>(d.base[d.rank]) + off;
        const long long n4 = n >> 2, total = n4 * d.world;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const int q = (int)(i / n4);
            if (q == d.rank) continue;
            const long long at = (long long)q * n + 4 * (i - (long long)q * n4);
            *reinterpret_cast<dp_f32x4*>(mine + at) = dp_load4(d.base[q] + off + at);
        }
    }
    dp_end(d, e, true);
}

// the probe's PRODUCER: arena[off + i] = pattern(rank, round, i), written by an ordinary launch with ordinary stores -- what a gradient-producing
// launch does.  Captured into ONE graph right in front of the pull, with no host synchronisation between the two.
__device__ __host__ inline float comm_pattern(int rank, int round, long long i) {
    unsigned h = (unsigned)(i * 2654435761ull) ^ (unsigned)(rank * 40503u + round * 9176u + 12345u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float m = (float)(int)(h & 0xffffu) * (1.0f / 32768.0f) - 1.0f;         // [-1, 1), 16 bits: exact
    const int ex = (int)((h >> 16) % 13u) - 6;                                    // 2^-6 .. 2^6
    return ldexpf(m, ex);
}
__global__ __launch_bounds__(256) void comm_fill_kernel(float* __restrict__ arena, long long off, long long n, int rank, int round) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) arena[off + i] = comm_pattern(rank, round, i);
}

// the pushed-slot exchange (dp_pull.h DpSlots: spedersac's Phibar / v) as a stand-alone producer / consumer pair -- the probe's third form
__global__ __launch_bounds__(256) void comm_slots_put_kernel(DpSlots d, int rank, int round) {
    const unsigned e = dp_slots_epoch(d);
    for (int f = blockIdx.x * 256 + threadIdx.x; f < d.n; f += gridDim.x * 256) dp_slots_put(d, e, f, comm_pattern(rank, round, f));
    dp_slots_publish(d, e);
}
__global__ __launch_bounds__(256) void comm_slots_sum_kernel(DpSlots d, float* __restrict__ out) {
    dp_slots_sum(d, out);         // (one block: every thread of it takes part in the barriers inside)
}

// the consumer half of a pushed exchange as a stage of a step program (spedersac: Phibar, v): out[0 .. d.n) = rank-ordered sum of the slots
extern "C" int rl_launch_slots_sum(const DpSlots* d, float* out, hipStream_t st) {
    if (!d || d->world < 2 || !out || d->n <= 0) return -7;
    hipLaunchKernelGGL(comm_slots_sum_kernel, dim3(1), dim3(256), 0, st, *d, out);
    return (int)hipGetLastError();
}

// no_done (both launchers): see DpPull::no_done -- ctrlsac's step program alternates gather and reduce-scatter on two channels; a peer's READY for one
// is sent after it has completed the other (stream order), so neither needs its own DONE round trip
extern "C" int rl_launch_xchg_gather(const DpPull* proto, int channel, long long off, long long n, int no_done, hipStream_t st) {
    if (!proto || proto->world < 2 || (off & 3) || (n & 3) || n <= 0) return -7;
    DpPull d = *proto; d.channel = channel; d.mode = 1; d.no_done = no_done;
    const long long want = (n / 4 * d.world + 255) / 256;
    d.nblocks = (int)(want < 1 ? 1 : (want > 512 ? 512 : want));
    hipLaunchKernelGGL(comm_gather_kernel, dim3(d.nblocks), dim3(256), 0, st, d, off, n);
    return (int)hipGetLastError();
}
// out (device, may alias this rank's block at `off`) = rank-ordered sum of every rank's block[off .. off + n)
extern "C" int rl_launch_xchg_reduce(const DpPull* proto, int channel, long long off, long long n, float* out, int two_shot, int no_done, hipStream_t st) {
    if (!proto || proto->world < 2 || n <= 0 || !out) return -7;
    DpPull d = *proto; d.channel = channel; d.no_done = no_done && !two_shot;
    const bool two = two_shot && d.world >= 3 && d.red[d.rank] && (off & 3) == 0 && (n & 3) == 0 && (((uintptr_t)out) & 15) == 0 && (n >> 2) <= 256ll * 65535;
    if (two) {
        d.mode = 2;
        d.nblocks = (int)(((n >> 2) + 255) / 256);
        d.shard4 = ((n >> 2) + d.world - 1) / d.world;
        d.nblocks_a = (int)((d.shard4 + 255) / 256);
        hipLaunchKernelGGL(comm_pull_kernel<2>, dim3(d.nblocks), dim3(256), 0, st, d, off, n, out);
    } else {
        d.mode = 1;
        const long long want = (n / 4 + 255) / 256;
        d.nblocks = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
        hipLaunchKernelGGL(comm_pull_kernel<1>, dim3(d.nblocks), dim3(256), 0, st, d, off, n, out);
    }
    return (int)hipGetLastError();
}

extern "C" {

int32_t rlrep_comm_create(int32_t rank, int32_t world, int64_t arena_floats, int64_t scratch_floats, rlrep_comm** out) {
    if (!out || world < 1 || world > RL_DP_MAX_WORLD || rank < 0 || rank >= world || arena_floats <= 0 || scratch_floats < 0) { rl_set_error("comm_create: bad argument"); return RLREP_ERR_ARG; }
    rlrep_comm* c = new rlrep_comm();
    c->rank = rank; c->world = world;
    c->arena_floats = (arena_floats + 63) & ~63ll;
    c->scratch_floats = (scratch_floats + 63) & ~63ll;
    c->red_floats = world >= 3 ? c->arena_floats + c->scratch_floats : 0;         // (the reduced region mirrors arena + scratch: same offsets)
    c->scratch_off = (size_t)c->arena_floats * sizeof(float);
    c->red_off = c->scratch_off + (size_t)c->scratch_floats * sizeof(float);
    c->flags_off = c->red_off + (size_t)c->red_floats * sizeof(float);
    c->bytes = c->flags_off + sizeof(DpFlags);
    const size_t bytes = c->bytes;
    bool fine = hipExtMallocWithFlags((void**)&c->local, bytes, hipDeviceMallocFinegrained) == hipSuccess;
    if (fine) {
        hipIpcMemHandle_t probe;
        if (hipIpcGetMemHandle(&probe, c->local) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(c->local); c->local = nullptr; fine = false; }
    } else (void)hipGetLastError();
    if (!fine && hipMalloc((void**)&c->local, bytes) != hipSuccess) { delete c; rl_set_error("comm_create: allocation of %zu bytes failed", bytes); return RLREP_ERR_NOMEM; }
    c->fine_grained = fine;
    if (hipHostMalloc((void**)&c->err_host, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&c->err_dev, c->err_host, 0) != hipSuccess) {
        (void)hipFree(c->local); delete c; rl_set_error("comm_create: cannot allocate the mapped error word"); return RLREP_ERR_NOMEM;
    }
    *c->err_host = 0;
    {
        int dev = 0, khz = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz >= 1000) c->ticks_per_us = khz / 1000;
        else (void)hipGetLastError();
        c->timeout_ticks = 120ll * 1000000ll * c->ticks_per_us;
    }
    (void)hipMemset(c->local, 0, bytes);
    (void)hipDeviceSynchronize();
    c->peer[rank] = c->local;
    c->connected = world == 1;
    *out = c;
    return 0;
}

float* rlrep_comm_arena(rlrep_comm* c) { return c ? reinterpret_cast<float*>(c->local) : nullptr; }
float* rlrep_comm_scratch(rlrep_comm* c) { return c && c->scratch_floats ? reinterpret_cast<float*>(c->local + c->scratch_off) : nullptr; }

int32_t rlrep_comm_handle_bytes(void) { return (int32_t)sizeof(hipIpcMemHandle_t); }

int32_t rlrep_comm_handle(rlrep_comm* c, void* out, int32_t cap) {
    if (!c || !out || cap < (int32_t)sizeof(hipIpcMemHandle_t)) { rl_set_error("comm_handle: bad argument"); return RLREP_ERR_ARG; }
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, c->local);
    if (e != hipSuccess) { rl_set_error("comm_handle: hipIpcGetMemHandle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e)); return RLREP_ERR_HIP; }
    memcpy(out, &h, sizeof(h));
    return 0;
}

// handles: world x rlrep_comm_handle_bytes(), in rank order (every rank's own entry is ignored)
int32_t rlrep_comm_connect(rlrep_comm* c, const void* handles) {
    if (!c || !handles) { rl_set_error("comm_connect: bad argument"); return RLREP_ERR_ARG; }
    for (int q = 0; q < c->world; ++q) {
        if (q == c->rank || c->opened[q]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)q * sizeof(h), sizeof(h));
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { rl_set_error("comm_connect: hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(e)); return RLREP_ERR_HIP; }
        c->peer[q] = (char*)p; c->opened[q] = true;
    }
    c->connected = true;
    return 0;
}

// LOOPBACK: all `world` comms live in THIS process (same device, or devices with peer access enabled by the caller): plain pointers, no IPC.
// peers[q] = the comm created with rank q (peers[rank] == comm).  What tools/exp/dp_loopback.py measures the exchange with -- N agents on N stream
// pairs of one GPU, no process time-slicing in the number -- and what the one-process multi-rank tests run on.
int32_t rlrep_comm_connect_local(rlrep_comm* c, rlrep_comm* const* peers) {
    if (!c || !peers) { rl_set_error("comm_connect_local: bad argument"); return RLREP_ERR_ARG; }
    for (int q = 0; q < c->world; ++q) {
        const rlrep_comm* o = peers[q];
        if (!o || o->world != c->world || o->rank != q || o->bytes != c->bytes || o->arena_floats != c->arena_floats || o->scratch_floats != c->scratch_floats) {
            rl_set_error("comm_connect_local: peers[%d] is not rank %d of a comm with this geometry", q, q); return RLREP_ERR_ARG;
        }
        if (q == c->rank && o != c) { rl_set_error("comm_connect_local: peers[rank] must be the comm itself"); return RLREP_ERR_ARG; }
        c->peer[q] = o->local;
    }
    c->connected = true;
    return 0;
}

// bound of every device-side wait from now on (attachments made LATER carry it; rlrep_comm_allreduce takes its own).  Default: 120 s.
int32_t rlrep_comm_set_timeout(rlrep_comm* c, int64_t timeout_us) {
    if (!c || timeout_us <= 0) { rl_set_error("comm_set_timeout: bad argument"); return RLREP_ERR_ARG; }
    c->timeout_ticks = timeout_us * c->ticks_per_us;
    return 0;
}

// out_dev[0 .. n) = sum over the ranks (rank order) of block[off .. off + n) (off < arena + scratch) -- stream-ordered, ONE launch, capturable (the
// epoch is a device counter).  Every rank must call it with the same (off, n, mode) in the same order, and must not overwrite that range of its
// block before the call has completed on its stream.  mode: 1 one-shot pull, 2 two-shot (world >= 3, off and n multiples of 4; else one-shot),
// 0 = by size as the optimizer launches choose.  timeout_us bounds every wait (0: the comm's); on a timeout the error word is set
// (rlrep_comm_status) and out_dev is not written.
int32_t rlrep_comm_allreduce(rlrep_comm* c, int64_t off, int64_t n, float* out_dev, int32_t mode, int64_t timeout_us, void* stream) {
    if (!c || !out_dev || n <= 0 || off < 0 || off + n > c->arena_floats + c->scratch_floats) { rl_set_error("comm_allreduce: bad argument (off = %lld, n = %lld, block = %lld floats)", (long long)off, (long long)n, c ? c->arena_floats + c->scratch_floats : 0ll); return RLREP_ERR_ARG; }
    if (!c->connected) { rl_set_error("comm_allreduce before rlrep_comm_connect"); return RLREP_ERR_STATE; }
    DpPull d; rl_comm_fill_pull(c, &d);
    if (timeout_us > 0) d.timeout = timeout_us * c->ticks_per_us;
    const int two = mode == 2 || (mode == 0 && n >= (1 << 17));
    const int rc = rl_launch_xchg_reduce(&d, 7, off, n, out_dev, two, 0, (hipStream_t)stream);
    ++g_rl_launches;
    if (rc != 0) { rl_set_error("comm_allreduce: launch failed (%d)", rc); return RLREP_ERR_HIP; }
    return 0;
}

// block[off + q * n .. + n) of every rank q gathered into every rank's block (all-gather by pull; one launch, capturable; channel 7)
int32_t rlrep_comm_allgather(rlrep_comm* c, int64_t off, int64_t n, void* stream) {
    if (!c || n <= 0 || off < 0 || (off & 3) || (n & 3) || off + n * c->world > c->arena_floats + c->scratch_floats) { rl_set_error("comm_allgather: bad argument"); return RLREP_ERR_ARG; }
    if (!c->connected) { rl_set_error("comm_allgather before rlrep_comm_connect"); return RLREP_ERR_STATE; }
    DpPull d; rl_comm_fill_pull(c, &d);
    const int rc = rl_launch_xchg_gather(&d, 7, off, n, 0, (hipStream_t)stream);      // (channel 7, like rlrep_comm_allreduce: the same READY / DONE protocol, one epoch sequence)
    ++g_rl_launches;
    if (rc != 0) { rl_set_error("comm_allgather: launch failed (%d)", rc); return RLREP_ERR_HIP; }
    return 0;
}

// the probe's producer launch: block[off + i] = pattern(rank, round, i), i < n (plain stores of an ordinary kernel); rlrep_comm_probe_value is the
// same function on the host
int32_t rlrep_comm_probe_fill(rlrep_comm* c, int64_t off, int64_t n, int32_t round, void* stream) {
    if (!c || n <= 0 || off < 0 || off + n > c->arena_floats + c->scratch_floats) { rl_set_error("comm_probe_fill: bad argument"); return RLREP_ERR_ARG; }
    const long long want = (n + 255) / 256;
    hipLaunchKernelGGL(comm_fill_kernel, dim3((unsigned)(want > 1024 ? 1024 : want)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float*>(c->local), (long long)off, (long long)n, c->rank, round);
    ++g_rl_launches;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("comm_probe_fill: %s", hipGetErrorString(e)); return RLREP_ERR_HIP; }
    return 0;
}
float rlrep_comm_probe_value(int32_t rank, int32_t round, int64_t i) { return comm_pattern(rank, round, i); }

// the probe of the PUSHED exchange (what carries spedersac's Phibar and v): a producer launch stores pattern(rank, round, .) of n floats into every
// rank's slot area (the head of the exchange scratch: needs 2 * world * n floats of it) and raises READY everywhere; the NEXT launch waits for all
// ranks' READY and sums the slots of its own block in rank order into out_dev[0 .. n).  Channel 6; two launches, capturable.
int32_t rlrep_comm_probe_slots(rlrep_comm* c, int64_t n, int32_t round, float* out_dev, int64_t timeout_us, void* stream) {
    if (!c || !out_dev || n <= 0 || n > 65536 || 2 * (int64_t)c->world * n > c->scratch_floats) { rl_set_error("comm_probe_slots: bad argument (needs 2 * world * n floats of exchange scratch)"); return RLREP_ERR_ARG; }
    if (!c->connected) { rl_set_error("comm_probe_slots before rlrep_comm_connect"); return RLREP_ERR_STATE; }
    DpSlots d; memset(&d, 0, sizeof(d));
    d.world = c->world; d.rank = c->rank; d.channel = 6; d.n = (int)n; d.err = c->err_dev;
    d.timeout = timeout_us > 0 ? timeout_us * c->ticks_per_us : c->timeout_ticks;
    for (int q = 0; q < c->world; ++q) { d.slot[q] = reinterpret_cast<float*>(c->peer[q] + c->scratch_off); d.flags[q] = reinterpret_cast<DpFlags*>(c->peer[q] + c->flags_off); }
    const int nb = (int)((n + 255) / 256 > 64 ? 64 : (n + 255) / 256);
    d.nblocks = nb;
    hipLaunchKernelGGL(comm_slots_put_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, d, c->rank, round);
    hipLaunchKernelGGL(comm_slots_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d, out_dev);
    g_rl_launches += 2;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("comm_probe_slots: %s", hipGetErrorString(e)); return RLREP_ERR_HIP; }
    return 0;
}

// Optimizer groups of `agent` whose gradient slice holds at most max_floats floats sum their gradients over the ranks INSIDE their optimizer
// launch from now on (the agent must have been created with this comm's arena as its gradient arena and hyper.world_size = world); slices of
// at least two_shot_floats floats take the two-shot form when world >= 3 (0: never).  With scratch of at least rlrep_layout_info.exchange_floats
// the agent's batch-coupled feature exchanges move into its launches too (rlrep_feature_exchange_count() drops to 0).  The step programs are
// REBUILT: call it before capturing graphs.  *attached_mask: bit g = group g is attached.  Larger groups keep the caller's all-reduce.
int32_t rlrep_comm_attach(rlrep_agent* agent, rlrep_comm* c, int64_t max_floats, int64_t two_shot_floats, int32_t* attached_mask) {
    if (!agent || !c) { rl_set_error("comm_attach: bad argument"); return RLREP_ERR_ARG; }
    if (!c->connected) { rl_set_error("comm_attach before rlrep_comm_connect"); return RLREP_ERR_STATE; }
    DpAttach at; memset(&at, 0, sizeof(at));
    rl_comm_fill_pull(c, &at.proto);
    for (int q = 0; q < c->world; ++q) at.scratch[q] = c->scratch_floats ? reinterpret_cast<float*>(c->peer[q] + c->scratch_off) : nullptr;
    at.arena_floats = c->arena_floats; at.scratch_floats = c->scratch_floats; at.max_floats = max_floats; at.two_shot_floats = two_shot_floats;
    int mask = 0;
    const int rc = rl_agent_attach_dp(agent, &at, &mask);
    if (attached_mask) *attached_mask = mask;
    return rc;
}

// 0 if no wait has timed out since the comm was created (or since the last call with clear != 0), else RLREP_ERR_STATE and bit q of *mask:
// a wait for rank q ran out.  Reads a word in mapped host memory: NO device synchronisation (poll it as often as you like); a timeout that
// happens in a launch still in flight is seen by a later call.
int32_t rlrep_comm_status(rlrep_comm* c, uint32_t* mask, int32_t clear) {
    if (!c) { rl_set_error("null comm"); return RLREP_ERR_ARG; }
    const unsigned w = __atomic_load_n(c->err_host, __ATOMIC_ACQUIRE);
    if (mask) *mask = w;
    if (w && clear) {
        __atomic_store_n(c->err_host, 0u, __ATOMIC_RELEASE);
        // ... and the device-side poison word that made this rank's exchange launches skip (rare path: a blocking copy is fine here)
        const unsigned zero = 0;
        DpFlags* f = reinterpret_cast<DpFlags*>(c->local + c->flags_off);
        (void)hipMemcpy(&f->poison, &zero, sizeof(zero), hipMemcpyHostToDevice);
    }
    if (w) { rl_set_error("data-parallel exchange: a peer did not arrive in time (late-rank mask 0x%x): the affected step was skipped on this rank, the replicas are no longer in step", w); return RLREP_ERR_STATE; }
    return 0;
}

int32_t rlrep_comm_fine_grained(rlrep_comm* c) { return c && c->fine_grained ? 1 : 0; }

// DEBUG / measurement: mark every peer as "arrived" and "has read" for the next `ahead` epochs of `channel` in this rank's flag block, so that the next
// one-shot launch of this rank on that channel runs without a live peer (synchronous; the peers' blocks must hold the data already).  What lets
// one stream play eight ranks one after the other: the arithmetic and the read fan-in of a full node on a one-GPU box (HIP gives a process four
// concurrent hardware queues: eight waiting launches cannot all be resident).  Never part of a train().
int32_t rlrep_comm_debug_preset(rlrep_comm* c, int32_t channel, int32_t ahead) {
    if (!c || channel < 0 || channel >= RL_DP_CHANNELS || ahead < 1) { rl_set_error("comm_debug_preset: bad argument"); return RLREP_ERR_ARG; }
    if (hipDeviceSynchronize() != hipSuccess) return RLREP_ERR_HIP;
    DpFlags* f = reinterpret_cast<DpFlags*>(c->local + c->flags_off);
    unsigned e = 0;
    if (hipMemcpy(&e, &f->epoch[channel], sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return RLREP_ERR_HIP;
    unsigned words[RL_DP_MAX_WORLD];
    for (int q = 0; q < RL_DP_MAX_WORLD; ++q) words[q] = e + (unsigned)ahead;          // (ahead = 1 << 30: the peers have "arrived" for the next 2^30 epochs)
    if (hipMemcpy(f->ready[channel], words, sizeof(words), hipMemcpyHostToDevice) != hipSuccess) return RLREP_ERR_HIP;
    if (hipMemcpy(f->done[channel], words, sizeof(words), hipMemcpyHostToDevice) != hipSuccess) return RLREP_ERR_HIP;
    return 0;
}

void rlrep_comm_destroy(rlrep_comm* c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    for (int q = 0; q < c->world; ++q) if (c->opened[q] && c->peer[q]) (void)hipIpcCloseMemHandle(c->peer[q]);
    if (c->local) (void)hipFree(c->local);
    if (c->err_host) (void)hipHostFree(c->err_host);
    delete c;
}

}  // extern "C"
