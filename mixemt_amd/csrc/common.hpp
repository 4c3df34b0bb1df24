// common.hpp -- part of libmixemt_hip.so (gfx950); included by mixemt_hip.hip only.
// Error plumbing, device query, wave / workgroup reductions shared by every kernel.
#ifndef MIXEMT_COMMON_HPP
#define MIXEMT_COMMON_HPP

typedef double d2 __attribute__((ext_vector_type(2)));

#define MXM_MAX_WG 1024            // upper bound on the persistent grid (workspace sizing)
#define MXM_MAX_BT 4               // restarts sharing one read of the matrix

// Which restarts a launch works on: slot b of the tile is restart s[b] of the [B][H] loop vectors
// and of the state array.  Passed by value, so a tile may name ANY restarts -- the loop driver deals
// full tiles round-robin over the restarts still running without moving their vectors.
struct mxm_slots {
    int s[MXM_MAX_BT];
};
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

// CUs of the calling thread's CURRENT device, cached per device (one host thread per GPU may share the process).
#define MXM_MAX_DEVICES 64
static std::atomic<int> g_num_cu[MXM_MAX_DEVICES];
static int num_cu() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    const bool cached = dev >= 0 && dev < MXM_MAX_DEVICES;
    if (cached) {
        const int have = g_num_cu[dev].load(std::memory_order_relaxed);
        if (have > 0) return have;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    const int n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (cached) g_num_cu[dev].store(n, std::memory_order_relaxed);
    return n;
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

// ------------------------------------------------------------------------------------------
// Second stage of a workgroup-wide sum for the streaming kernel: `red` holds N groups of NW
// per-wave partials (NW = 4 or 8, laid out [group][wave]).  Every wave fetches ONE partial per
// lane (lanes >= N * NW re-read element 0), sums each group of NW lanes with three DPP steps and
// hands the per-group results back in SGPRs (v_readlane) -- 2 VGPRs per wave instead of the
// 2 * N * NW each thread needs when it reads all partials itself (48 at 3 restarts x 8 waves:
// that was the peak of the kernel's register pressure, and what kept the batched shapes from a
// third ring slot).  Fixed tree order -> deterministic.  Needs the full wave active.
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
// Wave-wide sum by DPP only (no LDS-crossbar permutes): four in-row steps leave every lane with
// the sum of its row of 16, row_bcast:15 / row_bcast:31 then fold the four rows; the TOTAL is
// valid in lane 63 only.  ~6 x (2 v_mov_dpp + v_add_f64) against 6 dependent ds_bpermute round
// trips for the xor butterfly -- this sits on the per-row critical path of the streaming kernel.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_old_f64(double old, double v) {   // masked-out rows get `old`
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_lane63(double v) {
    v += dpp_mov_f64<0xB1>(v);                      // lane ^ 1
    v += dpp_mov_f64<0x4E>(v);                      // lane ^ 2
    v += dpp_mov_f64<0x141>(v);                     // row_half_mirror
    v += dpp_mov_f64<0x140>(v);                     // row_mirror: every lane = its row's sum
    v += dpp_mov_old_f64<0x142, 0xA>(0.0, v);       // row_bcast:15 into rows 1 and 3
    v += dpp_mov_old_f64<0x143, 0xC>(0.0, v);       // row_bcast:31 into rows 2 and 3
    return v;
}
// Three wave-wide sums for the price of one and a half (round 6; the batched quad pass reduces three restarts' dot
// products per row): gfx950's v_permlane32_swap / v_permlane16_swap exchange HALVES / ROWS of two registers, so each
// folding step also sorts the values apart -- afterwards the 16 lanes of a row hold one value's partial sums, and the four
// in-row DPP steps finish all three at once.  23 VALU instructions instead of 54 (three DPP ladders) and a third of the
// DPP hazard nops.  Returns z with EVERY lane of row 0 = sum(a), row 1 = sum(c), row 2 = sum(b), row 3 = sum(c).
// Fixed order: ((x[l] + x[l+32]) + (x[l+16] + x[l+48])) over l < 16, then the in-row tree.  Needs the full wave active.
__device__ __forceinline__ double swap32_add(double x, double y) {   // lanes < 32: x[l] + x[l+32]; lanes >= 32: y[l-32] + y[l]
    typedef unsigned int u2s __attribute__((ext_vector_type(2)));
    const u2s lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const u2s hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double swap16_add(double x, double y) {   // rows: x.r0 + x.r1 | y.r0 + y.r1 | x.r2 + x.r3 | y.r2 + y.r3
    typedef unsigned int u2s __attribute__((ext_vector_type(2)));
    const u2s lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const u2s hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double wave_sum3_by_row(double a, double b, double c) {
    const double ab = swap32_add(a, b);             // lanes < 32: a over {l, l+32}; lanes >= 32: b
    const double cc = swap32_add(c, c);             // every lane: c over {l mod 32, l mod 32 + 32}
    double v = swap16_add(ab, cc);                  // rows: a | c | b | c, 16 partial sums each
    v += dpp_mov_f64<0xB1>(v);                      // lane ^ 1
    v += dpp_mov_f64<0x4E>(v);                      // lane ^ 2
    v += dpp_mov_f64<0x141>(v);                     // row_half_mirror
    v += dpp_mov_f64<0x140>(v);                     // row_mirror: every lane = its row's sum
    return v;
}
__device__ __forceinline__ double wave_max_lane63(double v) {
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    v = fmax(v, dpp_mov_old_f64<0x142, 0xA>(v, v));
    v = fmax(v, dpp_mov_old_f64<0x143, 0xC>(v, v));
    return v;
}

__device__ __forceinline__ int wave_min_lane63_i32(int v) {       // the minimum lands in lane 63 (same ladder as above)
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xf, false));
    return v;
}

// Inclusive prefix sum over the wave's 64 lanes by DPP only (row_shr 1 / 2 / 4 / 8 inside the rows of 16, row_bcast
// 15 / 31 across them): six VALU steps instead of six ds_bpermute round trips on the LDS pipe.
__device__ __forceinline__ int wave_inclusive_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}

// w_r / Z_r of a row.  Z_r == 0 means the row is -inf everywhere (no haplogroup can have produced
// the read): the reference's E-step forms -inf - (-inf) = NaN there (em.py:81-83) and its weighted
// column logsumexp (em.py:87) carries the NaN into every proportion -- unless the row's weight is 0,
// which scipy drops.  Same here: the NaN coefficient times the row's zeros poisons every column sum.
__device__ __forceinline__ double weight_over_norm(double wr, double z) {
    if (z > 0.0) return wr / z;
    return (wr != 0.0) ? __builtin_nan("") : 0.0;
}

// Returns c[g] = weight_over_norm(num, sum_g) -- the one fp64 division runs once per wave for
// all groups (they sit in different lanes) instead of once per group.
template <int NW, int N>
__device__ __forceinline__ void group_ratio_to_sgpr(const double *red, int lane, double num, double (&c)[N]) {
    static_assert(NW == 4 || NW == 8 || NW == 16, "waves per workgroup");
    static_assert(N * NW <= 64, "one partial per lane");
    double v = red[lane < N * NW ? lane : 0];
    v += dpp_mov_f64<0xB1>(v);                      // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_mov_f64<0x4E>(v);                      // quad_perm [2,3,0,1]: lane ^ 2
    if constexpr (NW >= 8) v += dpp_mov_f64<0x141>(v);   // row_half_mirror: lane i <-> 7 - i
    if constexpr (NW == 16) v += dpp_mov_f64<0x140>(v);  // row_mirror: lane i <-> 15 - i
    v = weight_over_norm(num, v);
#pragma unroll
    for (int g = 0; g < N; ++g) c[g] = readlane_f64(v, g * NW);
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
// byte E of a code word, times 8 (the byte offset of a table entry): one SDWA shift -- the compiler takes a bit-field
// extract and a shift for it
template <int E>
__device__ __forceinline__ unsigned int code_byte_x8(unsigned int word) {
    const unsigned int three = 3;
    unsigned int r;
    if constexpr (E == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(word));
    else if constexpr (E == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(word));
    else if constexpr (E == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(word));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(word));
    return r;
}

struct row_deal {
    int64_t nq;                                     // steps (rows) of this workgroup
    int64_t bid, grid;                              // the workgroup's place in the dealing (its block index and the grid's
                                                    // size, or virtual ones: a kernel whose workgroups do two jobs)
    __device__ explicit row_deal(int64_t R) : row_deal(R, blockIdx.x, gridDim.x) {}
    __device__ row_deal(int64_t R, int64_t b, int64_t g) : nq((R - b + g - 1) / g), bid(b), grid(g) {}
    __device__ bool live(int64_t q) const { return q < nq; }
    // steps past the end re-read the workgroup's last row (their result is discarded)
    __device__ int64_t row(int64_t q) const { return bid + (q < nq ? q : nq - 1) * grid; }
};

#endif  // MIXEMT_COMMON_HPP
