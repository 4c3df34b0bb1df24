// common.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// Error plumbing, device query, wave / workgroup reductions shared by every kernel.
#ifndef MIXEMT_COMMON_HPP
#define MIXEMT_COMMON_HPP

typedef double d2 __attribute__((ext_vector_type(2)));

#define MXM_MAX_WG 1024            // upper bound on the persistent grid (workspace sizing)
#define MXM_LINEAR_MIN_H 65        // below this the log-space kernel is used

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *a = "", long long b = 0, long long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}
#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(-2, "HIP error: %s (line %lld)", hipGetErrorString(e_), __LINE__); \
    } while (0)

static int g_num_cu = 0;
static int num_cu() {
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_num_cu;
}

// ------------------------------------------------------------------------------------------
// wave / workgroup reductions (wave = 64 lanes on gfx950)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// all threads get the result; `scratch` holds THREADS/64 doubles; two barriers
template <int THREADS, bool IS_MAX>
__device__ __forceinline__ double block_reduce(double v, double *scratch) {
    constexpr int NW = THREADS / 64;
    v = IS_MAX ? wave_max(v) : wave_sum(v);
    __syncthreads();                       // scratch free (previous use finished)
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = IS_MAX ? fmax(r, scratch[i]) : r + scratch[i];
    return r;
}

__device__ __forceinline__ double logaddexp_f64(double a, double b) {
    // numpy.logaddexp semantics (em.py:156)
    if (a == b) return a + 0.693147180559945309417232121458176568;   // covers +-inf ties
    double d = a - b;
    if (d > 0) return a + log1p(exp(-d));
    if (d <= 0) return b + log1p(exp(d));
    return d;                                                         // NaN
}

// ------------------------------------------------------------------------------------------
// Row schedule of the streaming kernels: rows are DEALT round-robin over the workgroups --
// step q of workgroup b is row b + q * grid -- so at any moment the whole grid reads one
// contiguous window of grid * row_bytes (11 MB at 256 x 5408 fp64) that moves through the
// matrix, evenly spread over the HBM channels whatever the physical placement of the buffer.
// (Contiguous per-workgroup row blocks measured 0-10 % slower depending on the process /
// allocation: 256 far-apart streams whose channel mix depends on the block stride;
// profiles/r01/row_mapping.txt.)  A row is reached through its own buffer descriptor
// (scalar work), so no workgroup ever needs a descriptor range beyond one row.
// Requires gridDim.x <= R.
// ------------------------------------------------------------------------------------------
struct row_deal {
    int64_t nq;                                     // steps (rows) of this workgroup
    __device__ explicit row_deal(int64_t R) : nq((R - blockIdx.x + gridDim.x - 1) / gridDim.x) {}
    __device__ bool live(int64_t q) const { return q < nq; }
    // steps past the end re-read the workgroup's last row (their result is discarded)
    __device__ int64_t row(int64_t q) const {
        return (int64_t)blockIdx.x + (q < nq ? q : nq - 1) * (int64_t)gridDim.x;
    }
};

#endif  // MIXEMT_COMMON_HPP
