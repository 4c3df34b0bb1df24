// exchange.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// One-shot exchange of the M-step sums between the ranks of a row-sharded loop (SURVEY.md section 8 e: "each GPU writes
// its 43 KB to its peers over its links, sum in fixed rank order").  OPT-IN (dist.sharded_em_loop(exchange="oneshot")):
// RCCL's all-reduce stays the default.
#ifndef MIXEMT_EXCHANGE_HPP
#define MIXEMT_EXCHANGE_HPP

// ------------------------------------------------------------------------------------------
// Between mxm_em_iter and mxm_m_finalize a row-sharded loop adds up tile x H doubles (43 KB per restart) over the ranks: a
// latency-bound exchange, 15 us as a one-rank RCCL all-reduce here and the largest part of a records shard's 42 us tail.
// Here every rank owns ONE buffer, mapped into every peer's address space (hipIpc):
//     header   epoch (this rank's count of exchanges), ticket (arrival count of the push's workgroups)
//     flags    [2 parities][world]: flags[p][r] = the epoch whose sums rank r has finished writing into THIS buffer
//     slots    [2 parities][world][cap] doubles: slots[p][r] = rank r's sums of an exchange of parity p
//   push   (a kernel on the loop's stream, after the column reduce): epoch e = own epoch + 1; every workgroup writes its
//          part of the rank's sums into slot [e & 1][rank] of EVERY rank's buffer (its own included), makes the writes
//          visible system-wide and arrives at the ticket; the last one publishes e (own epoch) and sets
//          flags[e & 1][rank] = e in every buffer -- one write per peer and link, no reduction tree
//   pull   (the next kernel): waits until all `world` flags of parity e & 1 in its OWN buffer have reached e (bounded spin:
//          on a timeout the sums are poisoned with NaN and mxm_em_state.error = 2 is raised), then
//          colsum[i] = slots[e & 1][0][i] + slots[e & 1][1][i] + ... in RANK ORDER: every rank forms the same bits, so
//          all take the same stop decision -- what the loop needs from its collective
// Two parities suffice: a rank can only start exchange e + 2 after its pull of e + 1, which needs every peer's push of
// e + 1, which each peer issues after ITS pull of e: nobody still reads parity e & 1 when it is written again.  The epoch
// lives on the device (the push increments it), so a burst of iterations can be replayed from a captured hipGraph.
// Memory: the buffer is allocated fine-grained where the runtime offers it (peers write into it while its owner spins on
// it), flags and data are written with system-scope stores / a system-scope release fence and read with system-scope
// acquire loads.  TESTED with several processes on ONE GPU (the only hardware here); over xGMI it is UNMEASURED and the
// cross-device visibility rules above are the design, not an observation.
// ------------------------------------------------------------------------------------------
#define MXM_EXCHANGE_MAX_WORLD 16
#define MXM_EXCHANGE_THREADS 256
#define MXM_EXCHANGE_MAX_GRID 64          // workgroups of a launch that both pushes and waits: all resident at once

// spins are bounded by the wall clock (100 MHz): 3 s by default -- far beyond a straggler, short of a hang; the
// environment's MXM_EXCHANGE_TIMEOUT_MS (read when an exchange is created) sets another bound, e.g. for a rank that is
// being debugged or a host that stalls in a graph instantiation
#define EXCHANGE_TIMEOUT_TICKS_DEFAULT 300000000ull

struct exchange_header {
    unsigned long long epoch;
    unsigned int ticket;
    unsigned int pad_[13];                                   // 64 bytes
    unsigned long long flags[2][MXM_EXCHANGE_MAX_WORLD];     // 256 bytes
};

struct exchange_peers {
    unsigned char *base[MXM_EXCHANGE_MAX_WORLD];
};

struct mxm_exchange {
    int world = 0, rank = 0, device = 0;
    long long cap = 0;                                       // doubles per slot
    size_t bytes = 0;
    unsigned char *own = nullptr;
    bool fine_grained = false;
    unsigned long long timeout_ticks = EXCHANGE_TIMEOUT_TICKS_DEFAULT;
    exchange_peers peers;
    bool opened[MXM_EXCHANGE_MAX_WORLD];
    hipIpcMemHandle_t handle;
};

__device__ __forceinline__ double *exchange_slot(unsigned char *base, int parity, int rank, int world, long long cap) {
    return reinterpret_cast<double *>(base + sizeof(exchange_header)) + ((long long)parity * world + rank) * cap;
}

__global__ __launch_bounds__(MXM_EXCHANGE_THREADS) void exchange_push_kernel(exchange_peers peers, int world, int rank,
                                                                            long long cap,
                                                                            const double *__restrict__ colsum, long long n) {
    exchange_header *own = reinterpret_cast<exchange_header *>(peers.base[rank]);
    // every workgroup reads the epoch before any can publish the next one (the publisher is the LAST to arrive)
    const unsigned long long e = __hip_atomic_load(&own->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
    const int parity = (int)(e & 1ull);
    const long long stride = (long long)gridDim.x * MXM_EXCHANGE_THREADS;
    for (long long i = (long long)blockIdx.x * MXM_EXCHANGE_THREADS + threadIdx.x; i < n; i += stride) {
        const double v = colsum[i];
        for (int p = 0; p < world; ++p)
            __hip_atomic_store(exchange_slot(peers.base[p], parity, rank, world, cap) + i, v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();                                  // this thread's slot writes before the ticket
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int arrived = __hip_atomic_fetch_add(&own->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x - 1) {                      // every workgroup's writes are out
            __hip_atomic_store(&own->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&own->epoch, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            for (int p = 0; p < world; ++p)
                __hip_atomic_store(&reinterpret_cast<exchange_header *>(peers.base[p])->flags[parity][rank], e,
                                   __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}


__global__ __launch_bounds__(MXM_EXCHANGE_THREADS) void exchange_pull_kernel(unsigned char *own_base, int world, long long cap,
                                                                            double *__restrict__ colsum, long long n,
                                                                            mxm_em_state *__restrict__ state, int nb,
                                                                            unsigned long long timeout_ticks) {
    __shared__ int s_late;
    exchange_header *own = reinterpret_cast<exchange_header *>(own_base);
    // the push before this kernel on the same stream has published the epoch
    const unsigned long long e = __hip_atomic_load(&own->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    const int parity = (int)(e & 1ull);
    if (threadIdx.x == 0) s_late = 0;
    __syncthreads();
    if ((int)threadIdx.x < world) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&own->flags[parity][threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
            if (wall_clock64() - t0 > timeout_ticks) {
                s_late = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    const bool late = s_late != 0;                           // uniform
    const long long stride = (long long)gridDim.x * MXM_EXCHANGE_THREADS;
    for (long long i = (long long)blockIdx.x * MXM_EXCHANGE_THREADS + threadIdx.x; i < n; i += stride) {
        double s = 0.0;
        for (int r = 0; r < world; ++r)                      // rank order: the same bits on every rank
            s += __hip_atomic_load(exchange_slot(own_base, parity, r, world, cap) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        colsum[i] = late ? __longlong_as_double(0x7ff8000000000000ll) : s;
    }
    if (late && blockIdx.x == 0 && (int)threadIdx.x < nb && state != nullptr) state[threadIdx.x].error = 2u;
}

// push and pull in ONE launch (mxm_exchange_reduce): the workgroups push their parts, the last to arrive raises the flags,
// and every workgroup then waits for all ranks' flags -- its own rank's among them -- and sums its part of the slots.
// At most MXM_EXCHANGE_MAX_GRID workgroups, all resident at once, so the ones that wait cannot keep the ones that still
// have to push off the chip.  One launch boundary (~4 us) less than push + pull.
__global__ __launch_bounds__(MXM_EXCHANGE_THREADS) void exchange_reduce_kernel(exchange_peers peers, int world, int rank, long long cap,
                                                                              double *__restrict__ colsum, long long n,
                                                                              mxm_em_state *__restrict__ state, int nb,
                                                                              unsigned long long timeout_ticks) {
    __shared__ int s_late;
    unsigned char *own_base = peers.base[rank];
    exchange_header *own = reinterpret_cast<exchange_header *>(own_base);
    const unsigned long long e = __hip_atomic_load(&own->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
    const int parity = (int)(e & 1ull);
    const long long stride = (long long)gridDim.x * MXM_EXCHANGE_THREADS;
    for (long long i = (long long)blockIdx.x * MXM_EXCHANGE_THREADS + threadIdx.x; i < n; i += stride) {
        const double v = colsum[i];
        for (int p = 0; p < world; ++p)
            __hip_atomic_store(exchange_slot(peers.base[p], parity, rank, world, cap) + i, v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    if (threadIdx.x == 0) s_late = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int arrived = __hip_atomic_fetch_add(&own->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x - 1) {
            __hip_atomic_store(&own->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&own->epoch, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            for (int p = 0; p < world; ++p)
                __hip_atomic_store(&reinterpret_cast<exchange_header *>(peers.base[p])->flags[parity][rank], e,
                                   __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if ((int)threadIdx.x < world) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&own->flags[parity][threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
            if (wall_clock64() - t0 > timeout_ticks) {
                s_late = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    const bool late = s_late != 0;
    for (long long i = (long long)blockIdx.x * MXM_EXCHANGE_THREADS + threadIdx.x; i < n; i += stride) {
        double s = 0.0;
        for (int r = 0; r < world; ++r)
            s += __hip_atomic_load(exchange_slot(own_base, parity, r, world, cap) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        colsum[i] = late ? __longlong_as_double(0x7ff8000000000000ll) : s;
    }
    if (late && blockIdx.x == 0 && (int)threadIdx.x < nb && state != nullptr) state[threadIdx.x].error = 2u;
}

static size_t exchange_bytes(int world, long long cap) {
    return sizeof(exchange_header) + (size_t)2 * (size_t)world * (size_t)cap * sizeof(double);
}

extern "C" size_t mxm_exchange_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

extern "C" int mxm_exchange_create(int32_t world, int32_t rank, int64_t n_doubles, mxm_exchange **out, void *handle_out) {
    if (out == nullptr || handle_out == nullptr || world < 1 || world > MXM_EXCHANGE_MAX_WORLD || rank < 0 || rank >= world ||
        n_doubles < 1)
        return fail(-1, "mxm_exchange_create: 1..%s%lld ranks, a rank among them and a slot size required", "",
                    (long long)MXM_EXCHANGE_MAX_WORLD);
    mxm_exchange *x = new mxm_exchange();
    x->world = world;
    x->rank = rank;
    x->cap = (n_doubles + 1) / 2 * 2;
    x->bytes = exchange_bytes(world, x->cap);
    for (int p = 0; p < MXM_EXCHANGE_MAX_WORLD; ++p) {
        x->peers.base[p] = nullptr;
        x->opened[p] = false;
    }
    if (hipGetDevice(&x->device) != hipSuccess) x->device = 0;
    if (const char *ms = getenv("MXM_EXCHANGE_TIMEOUT_MS")) {
        const long long v = atoll(ms);
        if (v > 0) x->timeout_ticks = (unsigned long long)v * 100000ull;            // the wall clock counts at 100 MHz
    }
    // fault injection for the tests of the fall-back (dist.OneShotExchange): MXM_EXCHANGE_FAIL_RANK = k makes rank k's
    // create fail; MXM_EXCHANGE_REFUSE_FINE = 1 behaves as if the runtime refused to export the fine-grained buffer
    if (const char *fr = getenv("MXM_EXCHANGE_FAIL_RANK")) {
        if (fr[0] != '\0' && atoi(fr) == rank) {
            delete x;
            return fail(-2, "mxm_exchange_create: failure injected on rank %s%lld (MXM_EXCHANGE_FAIL_RANK)", "", (long long)rank);
        }
    }
    const char *rf = getenv("MXM_EXCHANGE_REFUSE_FINE");
    const bool refuse_fine = rf != nullptr && rf[0] == '1';
    // fine-grained where the runtime has it (peers write into this buffer while its owner's kernel spins on it) AND lets it
    // be exported; otherwise ordinary device memory -- the kernels' system-scope stores / acquire loads are what the
    // protocol rests on either way (mxm_exchange_info says which one it got)
    const size_t bytes = x->bytes;
    for (int attempt = 0; attempt < 2; ++attempt) {
        void *ptr = nullptr;
        const bool fine = attempt == 0;
        hipError_t e = fine ? hipExtMallocWithFlags(&ptr, bytes, hipDeviceMallocFinegrained) : hipMalloc(&ptr, bytes);
        if (e != hipSuccess || ptr == nullptr) {
            (void)hipGetLastError();
            if (fine) continue;
            delete x;
            return fail(-2, "mxm_exchange_create: cannot allocate %s%lld bytes", "", (long long)bytes);
        }
        e = hipMemset(ptr, 0, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = (fine && refuse_fine) ? hipErrorInvalidValue : hipIpcGetMemHandle(&x->handle, ptr);
        if (e == hipSuccess) {
            x->own = static_cast<unsigned char *>(ptr);
            x->fine_grained = fine;
            break;
        }
        (void)hipGetLastError();
        (void)hipFree(ptr);
        if (!fine) {
            const int rc = fail(-2, "mxm_exchange_create: %s", hipGetErrorString(e));
            delete x;
            return rc;
        }
    }
    x->peers.base[rank] = x->own;
    memcpy(handle_out, &x->handle, sizeof(hipIpcMemHandle_t));
    *out = x;
    return 0;
}

extern "C" int mxm_exchange_connect(mxm_exchange *x, const void *handles) {
    if (x == nullptr || handles == nullptr) return fail(-1, "mxm_exchange_connect: NULL argument%s", "");
    const unsigned char *h = static_cast<const unsigned char *>(handles);
    for (int p = 0; p < x->world; ++p) {
        if (p == x->rank || x->peers.base[p] != nullptr) continue;
        hipIpcMemHandle_t handle;
        memcpy(&handle, h + (size_t)p * sizeof(hipIpcMemHandle_t), sizeof(handle));
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess || ptr == nullptr) {
            (void)hipGetLastError();
            return fail(-2, "mxm_exchange_connect: cannot map rank %s%lld's buffer (%lld)", "", (long long)p, (long long)e);
        }
        x->peers.base[p] = static_cast<unsigned char *>(ptr);
        x->opened[p] = true;
    }
    return 0;
}

extern "C" int mxm_exchange_push(mxm_exchange *x, const double *colsum, int64_t n, void *stream) {
    if (x == nullptr || colsum == nullptr || n < 1 || n > x->cap) return fail(-1, "mxm_exchange_push: bad arguments%s", "");
    for (int p = 0; p < x->world; ++p)
        if (x->peers.base[p] == nullptr) return fail(-1, "mxm_exchange_push: rank %s%lld is not connected", "", (long long)p);
    int grid = (int)((n + MXM_EXCHANGE_THREADS * 4 - 1) / (MXM_EXCHANGE_THREADS * 4));
    if (grid < 1) grid = 1;
    if (grid > 64) grid = 64;
    hipLaunchKernelGGL(exchange_push_kernel, dim3(grid), dim3(MXM_EXCHANGE_THREADS), 0, (hipStream_t)stream, x->peers, x->world,
                       x->rank, x->cap, colsum, (long long)n);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_exchange_pull(mxm_exchange *x, double *colsum, int64_t n, mxm_em_state *state, int32_t nb, void *stream) {
    if (x == nullptr || colsum == nullptr || n < 1 || n > x->cap || nb < 0 || nb > MXM_EXCHANGE_THREADS)
        return fail(-1, "mxm_exchange_pull: bad arguments%s", "");
    int grid = (int)((n + MXM_EXCHANGE_THREADS * 4 - 1) / (MXM_EXCHANGE_THREADS * 4));
    if (grid < 1) grid = 1;
    if (grid > 64) grid = 64;
    hipLaunchKernelGGL(exchange_pull_kernel, dim3(grid), dim3(MXM_EXCHANGE_THREADS), 0, (hipStream_t)stream, x->own, x->world,
                       x->cap, colsum, (long long)n, state, (int)nb, x->timeout_ticks);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_exchange_reduce(mxm_exchange *x, double *colsum, int64_t n, mxm_em_state *state, int32_t nb, void *stream) {
    if (x == nullptr || colsum == nullptr || n < 1 || n > x->cap || nb < 0 || nb > MXM_EXCHANGE_THREADS)
        return fail(-1, "mxm_exchange_reduce: bad arguments%s", "");
    for (int p = 0; p < x->world; ++p)
        if (x->peers.base[p] == nullptr) return fail(-1, "mxm_exchange_reduce: rank %s%lld is not connected", "", (long long)p);
    int grid = (int)((n + MXM_EXCHANGE_THREADS * 4 - 1) / (MXM_EXCHANGE_THREADS * 4));
    if (grid < 1) grid = 1;
    if (grid > MXM_EXCHANGE_MAX_GRID) grid = MXM_EXCHANGE_MAX_GRID;
    hipLaunchKernelGGL(exchange_reduce_kernel, dim3(grid), dim3(MXM_EXCHANGE_THREADS), 0, (hipStream_t)stream, x->peers, x->world,
                       x->rank, x->cap, colsum, (long long)n, state, (int)nb, x->timeout_ticks);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mxm_exchange_info(const mxm_exchange *x, int32_t *fine_grained, int64_t *bytes) {
    if (x == nullptr) return fail(-1, "mxm_exchange_info: NULL handle%s", "");
    if (fine_grained != nullptr) *fine_grained = x->fine_grained ? 1 : 0;
    if (bytes != nullptr) *bytes = (int64_t)x->bytes;
    return 0;
}

extern "C" void mxm_exchange_destroy(mxm_exchange *x) {
    if (x == nullptr) return;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < x->world; ++p)
        if (x->opened[p] && x->peers.base[p] != nullptr) (void)hipIpcCloseMemHandle(x->peers.base[p]);
    if (x->own != nullptr) (void)hipFree(x->own);
    delete x;
}

#endif  // MIXEMT_EXCHANGE_HPP
