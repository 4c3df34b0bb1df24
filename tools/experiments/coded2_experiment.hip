// coded2_experiment.hip -- NOT part of libmixemt_hip.so.  Timing experiment for VERDICT r3 #3: the records row pass
// (coded_kernels.hpp, coded_row_pass) with TWO restarts per pass in one thread -- the code words, the byte shifts and
// the table lookups of a row are shared by both restarts; each restart keeps its own proportions, accumulators, wave
// sums and division.  Byte-coded rows only (wide rows and rows without a record are skipped: weight 0), no
// one-launch / resident variants: the main loop's time is what is compared.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I mixemt_amd/csrc \
//         tools/experiments/coded2_experiment.hip -o /tmp/libcoded2.so
// Driver: tools/experiments/time_coded2.py
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <cmath>
#include <type_traits>
#include "mixemt_hip.h"
#include "common.hpp"

#define C2_MAX_CODES 256

// KEEPV = false: the row's values are not kept between the dot product and the accumulation (48 registers) but looked
// up a second time -- the shape that fits three workgroups per CU.
template <int THREADS, int NCH, int NRUN, int NBUF, int WG_PER_CU, bool KEEPV = true>
__global__ __launch_bounds__(THREADS, WG_PER_CU *THREADS / 256) void coded2_kernel(
    const uint8_t *__restrict__ rec, const int64_t *__restrict__ rec_off, const int32_t *__restrict__ ndist, int ldc,
    const double *__restrict__ w, const double *__restrict__ props, int64_t R, int H, double *__restrict__ partial,
    int64_t ldpart) {
    constexpr int NW = THREADS / 64;
    __shared__ double s_tbl[NBUF][C2_MAX_CODES];
    __shared__ __attribute__((aligned(16))) double red[NBUF][NRUN][NW];
    __shared__ long long s_off[2][THREADS];
    __shared__ double s_wr[2][THREADS];
    __shared__ int s_nd[2][THREADS];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nword = ldc >> 2;
    const row_deal deal(R);
    const int voff = t * 4;
    int last_w = t + (NCH - 1) * THREADS;
    if (last_w > nword - 1) last_w = nword - 1;
    const int voff_last = last_w * 4;
    const bool tbl_thread = t < C2_MAX_CODES;
    const int tslot = t & (C2_MAX_CODES - 1);
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));

    double p[NRUN][NCH][4], acc[NRUN][NCH][4];
#pragma unroll
    for (int b = 0; b < NRUN; ++b)
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * (t + k * THREADS) + e;
                p[b][k][e] = (c < H) ? props[(int64_t)b * H + c] : 0.0;
                acc[b][k][e] = 0.0;
            }

    auto fetch_meta = [&](int half, int64_t q0) {
        const int64_t q = q0 + t;
        const int64_t r = deal.row(q);
        int nd = ndist[r];
        if (nd > C2_MAX_CODES) nd = 0;
        s_off[half][t] = rec_off[r];
        s_nd[half][t] = nd;
        s_wr[half][t] = (deal.live(q) && nd > 0) ? (w != nullptr ? w[r] : 1.0) : 0.0;
    };
    unsigned int cw[NBUF][NCH];
    double tring[NBUF];
    int pre_off_lo, pre_off_hi, pre_nd;
    double pre_wr;
    auto read_meta = [&](int64_t q_load, int64_t q_weight) {
        const int half = (int)((q_load / THREADS) & 1), idx = (int)(q_load % THREADS);
        const long long off = s_off[half][idx];
        pre_nd = __builtin_amdgcn_readfirstlane(s_nd[half][idx]);
        pre_off_hi = __builtin_amdgcn_readfirstlane((int)(off >> 32));
        pre_off_lo = __builtin_amdgcn_readfirstlane((int)off);
        pre_wr = s_wr[(q_weight / THREADS) & 1][q_weight % THREADS];
    };
    auto load_row = [&](unsigned int(&cws)[NCH], double &tbl_entry) {
        const int nd = pre_nd;
        const uint8_t *base = rec + (((long long)pre_off_hi << 32) | (unsigned int)pre_off_lo);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, ldc, 0x00020000);
#pragma unroll
        for (int k = 0; k < NCH - 1; ++k) cws[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, k * THREADS * 4, 2);
        cws[NCH - 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff_last, 0, 2);
        if (tbl_thread) {
            const auto rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base + ldc), 0, nd * 8, 0x00020000);
            const u2v v = __builtin_amdgcn_raw_buffer_load_b64(rt, tslot * 8, 0, 2);
            tbl_entry = __hiloint2double((int)v.y, (int)v.x);
        }
    };
    double v[NCH][4];
    auto lookup_row = [&](const char *tb, const unsigned int(&cws)[NCH]) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            v[k][0] = *reinterpret_cast<const double *>(tb + code_byte_x8<0>(cws[k]));
            v[k][1] = *reinterpret_cast<const double *>(tb + code_byte_x8<1>(cws[k]));
            v[k][2] = *reinterpret_cast<const double *>(tb + code_byte_x8<2>(cws[k]));
            v[k][3] = *reinterpret_cast<const double *>(tb + code_byte_x8<3>(cws[k]));
        }
    };
    auto look4 = [&](const char *tb, unsigned int word, double(&out)[4]) {
        out[0] = *reinterpret_cast<const double *>(tb + code_byte_x8<0>(word));
        out[1] = *reinterpret_cast<const double *>(tb + code_byte_x8<1>(word));
        out[2] = *reinterpret_cast<const double *>(tb + code_byte_x8<2>(word));
        out[3] = *reinterpret_cast<const double *>(tb + code_byte_x8<3>(word));
    };
    auto step = [&](auto J, int64_t q) {
        constexpr int j = decltype(J)::value;
        constexpr int jn = (j + 1) % NBUF, jl = (j + NBUF - 1) % NBUF;
        if ((q % THREADS) == 0) fetch_meta((int)((q / THREADS + 1) & 1), q + THREADS);
        load_row(cw[jl], tring[jl]);
        const double wr = pre_wr;
        double s[NRUN];
        const char *tbc = reinterpret_cast<const char *>(&s_tbl[j][0]);       // this row's table (published a step ago)
        if constexpr (!KEEPV) {
            static_assert(NRUN == 1 || KEEPV, "one restart in this form");
            // one chunk (4 cells) of lookups in flight ahead of the chunk being multiplied; scheduling barriers keep the
            // compiler from hoisting all 24 (and the next unrolled step's) into registers of their own
            double s4[4] = {0.0, 0.0, 0.0, 0.0};
            double cur[4], nxt[4];
            look4(tbc, cw[j][0], cur);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (k + 1 < NCH) look4(tbc, cw[j][k + 1], nxt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) s4[e] = fma(cur[e], p[0][k][e], s4[e]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) cur[e] = nxt[e];
            }
            s[0] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        } else {
#pragma unroll
            for (int b = 0; b < NRUN; ++b) {
                double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < NCH; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s4[e] = fma(v[k][e], p[b][k][e], s4[e]);
                s[b] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int b = 0; b < NRUN; ++b) s[b] = wave_sum_lane63(s[b]);
        if (lane == 63) {
#pragma unroll
            for (int b = 0; b < NRUN; ++b) red[j][b][wv] = s[b];
        }
        if (tbl_thread) s_tbl[jn][tslot] = tring[jn];
        __syncthreads();
        read_meta(q + NBUF, q + 1);
        double cf[NRUN];
#pragma unroll
        for (int b = 0; b < NRUN; ++b) {
            double z = 0.0;
            if constexpr (NW == 4) {
                z = (red[j][b][0] + red[j][b][1]) + (red[j][b][2] + red[j][b][3]);
            } else {
                z = ((red[j][b][0] + red[j][b][1]) + (red[j][b][2] + red[j][b][3])) +
                    ((red[j][b][4] + red[j][b][5]) + (red[j][b][6] + red[j][b][7]));
            }
            cf[b] = readlane_f64(weight_over_norm(wr, z), 0);
        }
        __builtin_amdgcn_s_setprio(0);
        const char *tbn = reinterpret_cast<const char *>(&s_tbl[jn][0]);
        auto upd = [&](int k, auto E) {
            constexpr int e = decltype(E)::value;
            if constexpr (!KEEPV) {
                acc[0][k][e] = fma(cf[0], *reinterpret_cast<const double *>(tbc + code_byte_x8<e>(cw[j][k])), acc[0][k][e]);
            } else {
#pragma unroll
                for (int b = 0; b < NRUN; ++b) {
                    acc[b][k][e] = fma(cf[b], v[k][e], acc[b][k][e]);
                    asm volatile("" : "+v"(acc[b][k][e]));
                }
                v[k][e] = *reinterpret_cast<const double *>(tbn + code_byte_x8<e>(cw[jn][k]));
            }
        };
        if constexpr (!KEEPV) {
            double cur[4], nxt[4];
            look4(tbc, cw[j][0], cur);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (k + 1 < NCH) look4(tbc, cw[j][k + 1], nxt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][k][e] = fma(cf[0], cur[e], acc[0][k][e]);
                    asm volatile("" : "+v"(acc[0][k][e]));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) cur[e] = nxt[e];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                upd(k, std::integral_constant<int, 0>{});
                upd(k, std::integral_constant<int, 1>{});
                upd(k, std::integral_constant<int, 2>{});
                upd(k, std::integral_constant<int, 3>{});
            }
        }
    };
    if (deal.nq > 0) {
        fetch_meta(0, 0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NBUF - 1; ++j) {
            read_meta(j, 0);
            load_row(cw[j], tring[j]);
        }
        if (tbl_thread) s_tbl[0][tslot] = tring[0];
        __syncthreads();
        read_meta(NBUF - 1, 0);
        if constexpr (KEEPV) lookup_row(reinterpret_cast<const char *>(&s_tbl[0][0]), cw[0]);
        for (int64_t q = 0; q < deal.nq; q += NBUF) {
            step(std::integral_constant<int, 0>{}, q);
            step(std::integral_constant<int, 1>{}, q + 1);
            step(std::integral_constant<int, 2>{}, q + 2);
            if constexpr (NBUF > 3) step(std::integral_constant<int, 3>{}, q + 3);
        }
    }
#pragma unroll
    for (int b = 0; b < NRUN; ++b) {
        double *dst = partial + ((int64_t)blockIdx.x * NRUN + b) * ldpart;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = 4 * (t + k * THREADS);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < H) dst[c + e] = acc[b][k][e];
        }
    }
}

// variant 0: 256 threads x 24 cells, one restart, two workgroups per CU (the product kernel's shape)
// variant 1: 512 threads x 12 cells, two restarts, one workgroup per CU  (VERDICT r3 #3's shape)
// variant 2: 256 threads x 24 cells, two restarts, one workgroup per CU (one wave per SIMD, up to 512 registers)
// variant 3: 512 threads x 12 cells, one restart, two workgroups per CU
// variant 4: 256 threads x 24 cells, one restart, THREE workgroups per CU, the row's values looked up twice
// variant 5: the same at two workgroups per CU;  variant 6: variant 4 with three rows in flight instead of four
// partial: [grid][NRUN][ldpart]; returns the average kernel time of `reps` launches in ms (negative: error)
extern "C" float coded2_time(int variant, const uint8_t *rec, const int64_t *rec_off, const int32_t *ndist, int ldc,
                             const double *w, const double *props, int64_t R, int H, double *partial, int64_t ldpart,
                             int n_cu, int reps, int *grid_out, int *nrun_out) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.0f;
    int grid = 0, nrun = 1;
    auto launch = [&]() {
        switch (variant) {
            case 0: grid = n_cu * 2; nrun = 1;
                hipLaunchKernelGGL((coded2_kernel<256, 6, 1, 4, 2>), dim3(grid), dim3(256), 0, 0, rec, rec_off, ndist, ldc, w, props, R, H, partial, ldpart); break;
            case 1: grid = n_cu; nrun = 2;
                hipLaunchKernelGGL((coded2_kernel<512, 3, 2, 4, 1>), dim3(grid), dim3(512), 0, 0, rec, rec_off, ndist, ldc, w, props, R, H, partial, ldpart); break;
            case 2: grid = n_cu; nrun = 2;
                hipLaunchKernelGGL((coded2_kernel<256, 6, 2, 4, 1>), dim3(grid), dim3(256), 0, 0, rec, rec_off, ndist, ldc, w, props, R, H, partial, ldpart); break;
            case 3: grid = n_cu * 2; nrun = 1;
                hipLaunchKernelGGL((coded2_kernel<512, 3, 1, 4, 2>), dim3(grid), dim3(512), 0, 0, rec, rec_off, ndist, ldc, w, props, R, H, partial, ldpart); break;
            case 4: grid = n_cu * 3; nrun = 1;
                hipLaunchKernelGGL((coded2_kernel<256, 6, 1, 4, 3, false>), dim3(grid), dim3(256), 0, 0, rec, rec_off, ndist, ldc, w, props, R, H, partial, ldpart); break;
            case 5: grid = n_cu * 2; nrun = 1;
                hipLaunchKernelGGL((coded2_kernel<256, 6, 1, 4, 2, false>), dim3(grid), dim3(256), 0, 0, rec, rec_off, ndist, ldc, w, props, R, H, partial, ldpart); break;
            case 6: grid = n_cu * 3; nrun = 1;
                hipLaunchKernelGGL((coded2_kernel<256, 6, 1, 3, 3, false>), dim3(grid), dim3(256), 0, 0, rec, rec_off, ndist, ldc, w, props, R, H, partial, ldpart); break;
            default: break;
        }
    };
    if (variant < 0 || variant > 6 || ldc / 4 > 1536) return -2.0f;
    launch();
    if (hipDeviceSynchronize() != hipSuccess) return -3.0f;
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    if (hipEventSynchronize(b) != hipSuccess) return -4.0f;
    float ms = 0.0f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    *grid_out = grid;
    *nrun_out = nrun;
    return ms / reps;
}
