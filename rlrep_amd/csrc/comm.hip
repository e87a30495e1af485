// One-shot gradient all-reduce over peer-mapped inboxes: the latency-shaped collective for the <= 2 MB gradient
// slices of the update path.  The reference has no collective at all (single process); RCCL's ring is bandwidth-shaped -- 2 (N - 1) serial
// hops, each bounded by one xGMI link -- while these messages are latency-bound.  Here every rank owns an INBOX in its own HBM:
//
//     inbox  = [2 buffers (epoch parity)][world slots][slot_floats]      slot q = what rank q sent
//     flags  = [world] last epoch rank q has finished sending
//
// exported with hipIpcGetMemHandle and mapped by every peer (hipIpcOpenMemHandle; over xGMI between GPUs of a node, the same HBM between
// two processes on one GPU).  One all-reduce of n floats is three stream-ordered launches on every rank:
//     push     copy my n floats into slot[my rank] of EVERY rank's inbox (7 links written concurrently, plus my own), system-scope fence
//     signal   one small block: store the epoch into flags[my rank] of every rank (release, system scope), then wait -- bounded -- until all
//              world flags of MY inbox have reached the epoch (acquire); a timeout sets the error word instead of hanging
//     reduce   data[i] = slot[0][i] + slot[1][i] + ... in RANK ORDER on every rank: the sums are bit-identical everywhere, so replicas
//              stay bit-identical without any broadcast
// Two buffers by epoch parity are enough: a rank can start all-reduce e + 1 (writing buffer (e + 1) & 1 of its peers) while a peer still
// reduces e, but not e + 2 -- its own reduce of e + 1 needs that peer's flag e + 1, which the peer sets after it has finished e.
// No float atomics.  Opt-in (RLREP_ONESHOT_ALLREDUCE=1, rlrep_amd/comm.py): its timing needs a multi-GPU node; its arithmetic and protocol
// are tested with two and three processes on one GPU (tests/test_comm.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/rlrep.h"

void rl_set_error(const char* fmt, ...);
extern long long g_rl_launches;

#define RL_COMM_MAX_WORLD 16

struct rlrep_comm {
    int rank = 0, world = 1;
    long long slot_floats = 0;
    unsigned epoch = 0;
    char* local = nullptr;                         // my inbox + flags (one allocation: exported as ONE handle)
    char* peer[RL_COMM_MAX_WORLD] = {nullptr};     // everyone's inbox as mapped here (peer[rank] = local)
    bool opened[RL_COMM_MAX_WORLD] = {false};
    unsigned* err = nullptr;                       // device error word (timeouts)
    size_t flags_off = 0, bytes = 0;
    bool fine_grained = false;
};

struct CommPtrs { float* slot[RL_COMM_MAX_WORLD]; unsigned* flag[RL_COMM_MAX_WORLD]; };

typedef float f32x4c __attribute__((ext_vector_type(4)));

// my data -> slot[rank] of every inbox.  grid.y = destination rank.
__global__ __launch_bounds__(256) void comm_push_kernel(const float* __restrict__ data, long long n, CommPtrs dst) {
    float* __restrict__ out = dst.slot[blockIdx.y];
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        reinterpret_cast<f32x4c*>(out)[i] = reinterpret_cast<const f32x4c*>(data)[i];
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = data[i];
    __threadfence_system();                        // the signal launch that follows publishes these stores to the peers
}

// one block: publish my epoch to everyone, then wait (bounded) for everyone's epoch in MY flags
__global__ __launch_bounds__(64) void comm_signal_wait_kernel(CommPtrs peers_flag_of_me, const unsigned* __restrict__ my_flags, int world, unsigned epoch,
                                                              unsigned* __restrict__ err, long long spins) {
    const int q = threadIdx.x;
    if (q < world) {
        __threadfence_system();
        __hip_atomic_store(peers_flag_of_me.flag[q], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        long long s = 0;
        // (epochs are compared modulo 2^32 as signed distances: the counter may wrap)
        while ((int)(__hip_atomic_load(my_flags + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            if (++s > spins) { atomicOr(err, 1u << (q & 15)); break; }       // never hang the GPU: report instead
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

// data = sum over ranks of slot[q], in rank order
__global__ __launch_bounds__(256) void comm_reduce_kernel(float* __restrict__ data, long long n, CommPtrs src, int world) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4c v = reinterpret_cast<const f32x4c*>(src.slot[0])[i];
        for (int q = 1; q < world; ++q) v += reinterpret_cast<const f32x4c*>(src.slot[q])[i];
        reinterpret_cast<f32x4c*>(data)[i] = v;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = src.slot[0][i];
        for (int q = 1; q < world; ++q) v += src.slot[q][i];
        data[i] = v;
    }
}

extern "C" {

int32_t rlrep_comm_create(int32_t rank, int32_t world, int64_t slot_floats, rlrep_comm** out) {
    if (!out || world < 1 || world > RL_COMM_MAX_WORLD || rank < 0 || rank >= world || slot_floats <= 0) { rl_set_error("comm_create: bad argument"); return RLREP_ERR_ARG; }
    rlrep_comm* c = new rlrep_comm();
    c->rank = rank; c->world = world; c->slot_floats = (slot_floats + 3) & ~3ll;
    const size_t inbox = (size_t)2 * world * c->slot_floats * sizeof(float);
    c->flags_off = (inbox + 255) & ~(size_t)255;
    c->bytes = c->flags_off + 256 + 256;                  // flags [world] (one 256-byte line), error word (another)
    // FINE-GRAINED device memory: peers write it over xGMI behind this GPU's L2, and the reduce launch must not be served stale lines from an
    // earlier epoch (coarse-grained memory is only coherent at this device's own kernel boundaries).  If the runtime cannot export such an
    // allocation over IPC, fall back to plain hipMalloc (correct between processes that share one GPU, e.g. the tests).
    bool fine = hipExtMallocWithFlags((void**)&c->local, c->bytes, hipDeviceMallocFinegrained) == hipSuccess;
    if (fine) {
        hipIpcMemHandle_t probe;
        if (hipIpcGetMemHandle(&probe, c->local) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(c->local); c->local = nullptr; fine = false; }
    }
    if (!fine && hipMalloc((void**)&c->local, c->bytes) != hipSuccess) { delete c; rl_set_error("comm_create: allocation of %zu bytes failed", c->bytes); return RLREP_ERR_NOMEM; }
    c->fine_grained = fine;
    (void)hipMemset(c->local, 0, c->bytes);
    (void)hipDeviceSynchronize();
    c->err = reinterpret_cast<unsigned*>(c->local + c->flags_off + 256);
    c->peer[rank] = c->local;
    *out = c;
    return 0;
}

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
    return 0;
}

// In-place sum over the ranks of data[0 .. n) (n <= slot_floats), stream-ordered, three launches.  Every rank must call it with the same n, in
// the same order.  timeout_spins bounds the wait for the slowest peer (0: a default of ~2 s); on a timeout the error word is set -- see
// rlrep_comm_status -- and the reduce runs on whatever has arrived (the caller must treat the step as failed).
int32_t rlrep_comm_allreduce(rlrep_comm* c, float* data, int64_t n, int64_t timeout_spins, void* stream) {
    if (!c || !data || n <= 0 || n > c->slot_floats) { rl_set_error("comm_allreduce: bad argument (n = %lld, slot = %lld floats)", (long long)n, c ? c->slot_floats : 0ll); return RLREP_ERR_ARG; }
    for (int q = 0; q < c->world; ++q) if (!c->peer[q]) { rl_set_error("comm_allreduce before rlrep_comm_connect"); return RLREP_ERR_STATE; }
    if ((((uintptr_t)data) & 15) != 0) { rl_set_error("comm_allreduce: data must be 16-byte aligned"); return RLREP_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const unsigned epoch = ++c->epoch;
    const size_t buf = (size_t)(epoch & 1) * c->world * c->slot_floats;
    CommPtrs push; memset(&push, 0, sizeof(push));
    CommPtrs mine; memset(&mine, 0, sizeof(mine));
    for (int q = 0; q < c->world; ++q) {
        push.slot[q] = reinterpret_cast<float*>(c->peer[q]) + buf + (size_t)c->rank * c->slot_floats;        // my slot in rank q's inbox
        push.flag[q] = reinterpret_cast<unsigned*>(c->peer[q] + c->flags_off) + c->rank;                       // my flag in rank q's inbox
        mine.slot[q] = reinterpret_cast<float*>(c->local) + buf + (size_t)q * c->slot_floats;                 // rank q's slot in my inbox
    }
    const int blocks = (int)((n / 4 + 255) / 256 < 1 ? 1 : ((n / 4 + 255) / 256 > 256 ? 256 : (n / 4 + 255) / 256));
    hipLaunchKernelGGL(comm_push_kernel, dim3(blocks, c->world), dim3(256), 0, st, (const float*)data, (long long)n, push);
    hipLaunchKernelGGL(comm_signal_wait_kernel, dim3(1), dim3(64), 0, st, push, reinterpret_cast<const unsigned*>(c->local + c->flags_off), c->world, epoch, c->err,
                       (long long)(timeout_spins > 0 ? timeout_spins : (1ll << 22)));
    hipLaunchKernelGGL(comm_reduce_kernel, dim3(blocks), dim3(256), 0, st, data, (long long)n, mine, c->world);
    g_rl_launches += 3;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("comm_allreduce: %s", hipGetErrorString(e)); return RLREP_ERR_HIP; }
    return 0;
}

// Synchronises `stream`, returns 0 if no wait has timed out since the comm was created, else RLREP_ERR_STATE (bit q of *mask: rank q was late)
int32_t rlrep_comm_status(rlrep_comm* c, uint32_t* mask, void* stream) {
    if (!c) { rl_set_error("null comm"); return RLREP_ERR_ARG; }
    unsigned w = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(&w, c->err, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) { rl_set_error("comm_status: device error"); return RLREP_ERR_HIP; }
    if (mask) *mask = w;
    if (w) { rl_set_error("one-shot all-reduce: a peer did not arrive in time (late-rank mask 0x%x)", w); return RLREP_ERR_STATE; }
    return 0;
}

int32_t rlrep_comm_fine_grained(rlrep_comm* c) { return c && c->fine_grained ? 1 : 0; }

void rlrep_comm_destroy(rlrep_comm* c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    for (int q = 0; q < c->world; ++q) if (c->opened[q] && c->peer[q]) (void)hipIpcCloseMemHandle(c->peer[q]);
    if (c->local) (void)hipFree(c->local);
    delete c;
}

}  // extern "C"
