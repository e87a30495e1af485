// The shared gradient arena of the data-parallel form: allocation, hipIpc export / mapping, attachment to an agent,
// the stand-alone pull all-reduce (probe and tests) and the error word.  The exchange itself lives in dp_pull.h and runs INSIDE the optimizer
// launches (elementwise.hip adam_kernel): zero launches per all-reduce.  The reference has no collective (one process); the call sites the
// exchange belongs to are its backward() -> step() pairs: agent/vlsac/vlsac_agent.py:153-154, 183-184, 229-230,
// agent/ctrlsac/ctrlsac_agent.py:243-244, agent/spedersac/spedersac_agent.py:211-212, agent/diffsrsac/diffsrsac_agent.py:311-314.
//
//     block  = [ arena_floats floats (the caller's gradient arena) | DpFlags ]            one allocation, ONE hipIpc handle
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
    long long arena_floats = 0;
    char* local = nullptr;                         // my block
    char* peer[RL_DP_MAX_WORLD] = {nullptr};       // everyone's block as mapped here (peer[rank] = local)
    bool opened[RL_DP_MAX_WORLD] = {false};
    unsigned* err_host = nullptr;                  // error word: pinned host memory, mapped into the device's address space
    unsigned* err_dev = nullptr;
    size_t flags_off = 0, bytes = 0;
    bool fine_grained = false, connected = false;
};

// rlrep_agent side (engine.hip): the optimizer launches of groups attached to a comm carry a DpPull
extern "C" int rl_agent_attach_dp(rlrep_agent* ag, const DpPull* proto, long long arena_floats, long long max_floats, int* attached_mask);

extern "C" void rl_comm_fill_pull(const rlrep_comm* c, DpPull* d) {
    memset(d, 0, sizeof(*d));
    d->world = c->world; d->rank = c->rank; d->spins = 1ll << 24; d->err = c->err_dev;
    for (int q = 0; q < c->world; ++q) {
        d->base[q] = reinterpret_cast<const float*>(c->peer[q]);
        d->flags[q] = reinterpret_cast<DpFlags*>(c->peer[q] + c->flags_off);
    }
}

// out[i] = sum over the ranks, in rank order, of arena_q[off + i]: ONE launch (channel 7); every block waits for the peers' READY, the last one
// to finish runs the DONE handshake (dp_pull.h).  The probe of rlrep_amd/comm.py and tests/test_comm.py; the optimizer launches do the same
// with the gradient of their own elements.
__global__ __launch_bounds__(256) void comm_pull_kernel(DpPull d, long long off, long long n, float* __restrict__ out) {
    const unsigned e = dp_begin(d, blockIdx.x == 0, true);
    const long long n4 = n >> 2;
    if ((off & 3) == 0 && (((uintptr_t)out) & 15) == 0) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
            reinterpret_cast<dp_f32x4*>(out)[i] = dp_sum4(d, off + 4 * i);
        for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = dp_sum1(d, off + i);
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = dp_sum1(d, off + i);
    }
    dp_end(d, e, true);
}

extern "C" {

int32_t rlrep_comm_create(int32_t rank, int32_t world, int64_t arena_floats, rlrep_comm** out) {
    if (!out || world < 1 || world > RL_DP_MAX_WORLD || rank < 0 || rank >= world || arena_floats <= 0) { rl_set_error("comm_create: bad argument"); return RLREP_ERR_ARG; }
    rlrep_comm* c = new rlrep_comm();
    c->rank = rank; c->world = world; c->arena_floats = (arena_floats + 63) & ~63ll;
    c->flags_off = (size_t)c->arena_floats * sizeof(float);
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
    (void)hipMemset(c->local, 0, bytes);
    (void)hipDeviceSynchronize();
    c->peer[rank] = c->local;
    c->connected = world == 1;
    *out = c;
    return 0;
}

float* rlrep_comm_arena(rlrep_comm* c) { return c ? reinterpret_cast<float*>(c->local) : nullptr; }

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

// out_dev[0 .. n) = sum over the ranks (rank order) of arena[off .. off + n) -- stream-ordered, ONE launch, capturable (the epoch is a device
// counter).  Every rank must call it with the same (off, n) in the same order, and must not overwrite that range of its arena before the call
// has completed on its stream.  timeout_spins bounds every wait (0: a default of several seconds); on a timeout the error word is set
// (rlrep_comm_status) and the sum is formed from whatever the arenas hold.
int32_t rlrep_comm_allreduce(rlrep_comm* c, int64_t off, int64_t n, float* out_dev, int64_t timeout_spins, void* stream) {
    if (!c || !out_dev || n <= 0 || off < 0 || off + n > c->arena_floats) { rl_set_error("comm_allreduce: bad argument (off = %lld, n = %lld, arena = %lld floats)", (long long)off, (long long)n, c ? c->arena_floats : 0ll); return RLREP_ERR_ARG; }
    if (!c->connected) { rl_set_error("comm_allreduce before rlrep_comm_connect"); return RLREP_ERR_STATE; }
    DpPull d; rl_comm_fill_pull(c, &d);
    d.channel = 7;
    if (timeout_spins > 0) d.spins = timeout_spins;
    const long long want = (n / 4 + 255) / 256;
    d.nblocks = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
    hipLaunchKernelGGL(comm_pull_kernel, dim3(d.nblocks), dim3(256), 0, (hipStream_t)stream, d, (long long)off, (long long)n, out_dev);
    ++g_rl_launches;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("comm_allreduce: %s", hipGetErrorString(e)); return RLREP_ERR_HIP; }
    return 0;
}

// Optimizer groups of `agent` whose gradient slice holds at most max_floats floats sum their gradients over the ranks INSIDE their optimizer
// launch from now on (the agent must have been created with this comm's arena as its gradient arena and hyper.world_size = world).
// *attached_mask: bit g = group g is attached.  Larger groups keep the caller's all-reduce between backward and apply.
int32_t rlrep_comm_attach(rlrep_agent* agent, rlrep_comm* c, int64_t max_floats, int32_t* attached_mask) {
    if (!agent || !c) { rl_set_error("comm_attach: bad argument"); return RLREP_ERR_ARG; }
    if (!c->connected) { rl_set_error("comm_attach before rlrep_comm_connect"); return RLREP_ERR_STATE; }
    DpPull d; rl_comm_fill_pull(c, &d);
    int mask = 0;
    const int rc = rl_agent_attach_dp(agent, &d, c->arena_floats, max_floats, &mask);
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
    if (w && clear) __atomic_store_n(c->err_host, 0u, __ATOMIC_RELEASE);
    if (w) { rl_set_error("data-parallel gradient exchange: a peer did not arrive in time (late-rank mask 0x%x): the affected step is invalid", w); return RLREP_ERR_STATE; }
    return 0;
}

int32_t rlrep_comm_fine_grained(rlrep_comm* c) { return c && c->fine_grained ? 1 : 0; }

void rlrep_comm_destroy(rlrep_comm* c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    for (int q = 0; q < c->world; ++q) if (c->opened[q] && c->peer[q]) (void)hipIpcCloseMemHandle(c->peer[q]);
    if (c->local) (void)hipFree(c->local);
    if (c->err_host) (void)hipHostFree(c->err_host);
    delete c;
}

}  // extern "C"
