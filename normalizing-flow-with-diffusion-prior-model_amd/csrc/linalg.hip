// Batched log|det W| and W^-1 for the invertible 1x1 convolutions, in fp64.
// Reference: InvConv2d.transform  transforms.py:131  torch.slogdet(W.double())[1].float()
//            InvConv2d.invert     transforms.py:144  W.inverse()            (fp32 LAPACK there; fp64 here)
// The reference recomputes both on every call (48 slogdet per forward at L3K16); here ONE launch handles every
// flow's matrix of a level: one 256-thread workgroup per matrix runs Gauss-Jordan with partial pivoting on [W | I]
// held in LDS (C <= 64: 64 KB) or in a caller-provided global workspace (C up to 256, L2-resident).
// Deterministic: pivot ties go to the lowest row, every sum has a fixed order.
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxLdsC = 64;

// Gauss-Jordan on [W | I] with IMPLICIT partial pivoting (no row swaps): step k picks, among the rows not yet used as
// a pivot, the one with the largest |A[r][k]| (lowest r on ties), normalises it and eliminates column k from every
// other row.  After C steps physical row rowof[k] carries e_k on the left and row k of W^-1 on the right.
// Three barriers per step: (1) per-wave shuffle argmax -> LDS, (2) pivot row / column factors staged, (3) update.
template <bool IN_LDS>
__global__ void __launch_bounds__(kBlock) gauss_jordan_kernel(const float* __restrict__ w, int C,
                                                              float* __restrict__ logabsdet, float* __restrict__ winv,
                                                              double* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int W2 = 2 * C;
    double* A = IN_LDS ? sm : ws + (size_t)blockIdx.x * C * W2;
    double* rowk = IN_LDS ? sm + (size_t)C * W2 : sm;             // [2C] normalised pivot row
    double* fcol = rowk + W2;                                      // [C]  elimination factors
    double* wval = fcol + C;                                       // [4]  per-wave argmax value
    int* widx = reinterpret_cast<int*>(wval + 4);                  // [4]  per-wave argmax row
    int* rowof = widx + 4;                                         // [C]  pivot row of step k
    int* used = rowof + C;                                         // [C]
    const float* wm = w + (size_t)blockIdx.x * C * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int e = tid; e < C * W2; e += kBlock) {
        const int i = e / W2, j = e % W2;
        A[e] = j < C ? (double)wm[i * C + j] : (j - C == i ? 1.0 : 0.0);
    }
    for (int i = tid; i < C; i += kBlock) used[i] = 0;
    double logdet = 0.0;
    const int i0 = tid / W2, j0 = tid % W2, di = kBlock / W2, dj = kBlock % W2;
    __syncthreads();

    for (int k = 0; k < C; ++k) {
        // 1. pivot search: row r = tid (C <= 256), shuffle argmax inside each wave, lowest row wins ties
        double best = -1.0;
        int bi = 1 << 30;
        if (tid < C && !used[tid]) { best = fabs(A[tid * W2 + k]); bi = tid; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { wval[wave] = best; widx[wave] = bi; }
        __syncthreads();
        int p = widx[0];
        double pb = wval[0];
#pragma unroll
        for (int q = 1; q < kBlock / 64; ++q)
            if (wval[q] > pb || (wval[q] == pb && widx[q] < p)) { pb = wval[q]; p = widx[q]; }
        // 2. stage the normalised pivot row and the column factors (reads only; nothing is modified yet)
        const double pv = A[p * W2 + k];
        for (int j = tid; j < W2; j += kBlock) rowk[j] = A[p * W2 + j] / pv;
        for (int i = tid; i < C; i += kBlock) fcol[i] = (i == p) ? 0.0 : A[i * W2 + k];
        if (tid == 0) { logdet += log(fabs(pv)); rowof[k] = p; used[p] = 1; }
        __syncthreads();
        // 3. update every element; (i, j) advance incrementally (no runtime div/mod in the hot loop)
        {
            int i = i0, j = j0;
            for (int e = tid; e < C * W2; e += kBlock) {
                if (i == p) A[e] = rowk[j];
                else {
                    const double f = fcol[i];
                    if (f != 0.0) A[e] -= f * rowk[j];
                }
                i += di; j += dj;
                if (j >= W2) { j -= W2; ++i; }
            }
        }
        __syncthreads();
    }
    if (tid == 0) logabsdet[blockIdx.x] = (float)logdet;
    if (winv) {
        float* out = winv + (size_t)blockIdx.x * C * C;
        for (int e = tid; e < C * C; e += kBlock) out[e] = (float)A[rowof[e / C] * W2 + C + (e % C)];
    }
}


// ---- 64 < C <= 192: the matrix in REGISTERS ---------------------------------------------------------------------------
// An fp64 192 x 192 matrix is 295 KB - more than LDS - and the kernel above then works in a global workspace: three barriers
// and a pass over [W | I] in L2 per step, 12.5 ms for the sixteen C = 192 matrices of a 5-level model (a third of its forward
// pass).  Here one 1024-thread workgroup holds the matrix in registers, 2D-cyclic: thread (ti, tj) = (tid / 32, tid % 32) owns
// the elements (ti + 32 a, tj + 32 b), NB x NB doubles.  IN-PLACE Gauss-Jordan inversion (no identity half) with implicit
// partial pivoting: step k takes the unused row p with the largest |A[p][k]| (lowest p on ties), broadcasts row p and column k
// through LDS, and every thread updates its own elements; the array then holds M with W^-1[k][p_j] = M[p_k][j].
// (C <= 128: 32 x 32 threads; C <= 192: 16 x 32 threads with 12 x 6 elements each, 144 of the 256 registers of a wave.)
// Rows / columns beyond C are padded with identity and never chosen.  Two barriers per step (LDS scratch double-buffered).
template <int TI, int NBR, int NBC>       // TI x 32 threads; thread (ti, tj) owns rows ti + TI a (a < NBR), columns tj + 32 b (b < NBC)
__global__ void __launch_bounds__(TI * 32) gauss_jordan_reg_kernel(const float* __restrict__ w, int C, float* __restrict__ logabsdet,
                                                                   float* __restrict__ winv) {
    constexpr int CP = 32 * NBC;
    static_assert(TI * NBR == CP, "square padded matrix");
    __shared__ double cand_v[2][TI];
    __shared__ int cand_i[2][TI];
    __shared__ double rowk[2][CP];
    __shared__ double fcol[2][CP];
    __shared__ int used[CP];
    __shared__ int pivrow[CP];
    __shared__ int kof[CP];
    __shared__ double pivlog[CP];
    __shared__ double pinv[2][2];                                   // the pivot and its reciprocal, by the pivot's owner
    static_assert(TI == 16 || TI == 32, "a wave is two rows of threads; candidates reduced by xor-shuffles");
    const int tid = threadIdx.x, ti = tid >> 5, tj = tid & 31;
    const float* wm = w + (size_t)blockIdx.x * C * C;
    double a[NBR][NBC];
#pragma unroll
    for (int x = 0; x < NBR; ++x)
#pragma unroll
        for (int y = 0; y < NBC; ++y) {
            const int i = ti + TI * x, j = tj + 32 * y;
            a[x][y] = (i < C && j < C) ? (double)wm[i * C + j] : (i == j ? 1.0 : 0.0);
        }
    for (int i = tid; i < CP; i += TI * 32) used[i] = 0;
    __syncthreads();
    for (int k = 0; k < C; ++k) {
        const int buf = k & 1, kb = k >> 5, kt = k & 31;
        // 1. pivot candidates: the TI threads that own column k (tj == k % 32), each over its NBR rows
        if (tj == kt) {
            double best = -1.0;
            int bi = 1 << 30;
#pragma unroll
            for (int x = 0; x < NBR; ++x) {
                const int i = ti + TI * x;
                double v = 0.0;
#pragma unroll
                for (int y = 0; y < NBC; ++y) if (y == kb) v = a[x][y];      // (kb is uniform: scalar branches, no select chain)
                fcol[buf][i] = v;                                   // column k, all rows (read in step 3)
                v = fabs(v);
                if (i < C && !used[i] && (v > best || (v == best && i < bi))) { best = v; bi = i; }
            }
            cand_v[buf][ti] = best;
            cand_i[buf][ti] = bi;
        }
        __syncthreads();
        // the pivot: largest |candidate|, lowest row on ties.  Every wave reduces the TI candidates across its lanes (lane l takes
        // candidate l % TI; five xor-shuffle steps) - with every thread walking all TI candidates through LDS the scan was a quarter
        // of the step's VALU time on each of the 16 waves (the step is VALU-bound: four waves share a SIMD).
        double pb = cand_v[buf][tid % TI];
        int p = cand_i[buf][tid % TI];
#pragma unroll
        for (int o = 1; o < TI; o <<= 1) {
            const double ov = __shfl_xor(pb, o);
            const int oi = __shfl_xor(p, o);
            if (ov > pb || (ov == pb && oi < p)) { pb = ov; p = oi; }
        }
        // bookkeeping by one thread, BEFORE this step's second barrier: the next step's candidate search reads used[] right
        // after it (written at the end of the step it raced with that search: a pivot row could be chosen twice)
        if (tid == 0) { used[p] = 1; pivrow[k] = p; kof[p] = k; }
        // 2. the owners of row p publish it; the owner of the pivot itself also publishes 1 / pivot (ONE fp64 division per step,
        //    not one per thread)
        if (ti == p % TI) {
            const int pxl = p / TI;
#pragma unroll
            for (int y = 0; y < NBC; ++y) {
                double v = 0.0;
#pragma unroll
                for (int x = 0; x < NBR; ++x) if (x == pxl) v = a[x][y];
                rowk[buf][tj + 32 * y] = v;
                if (y == kb && tj == kt) { pinv[buf][0] = v; pinv[buf][1] = 1.0 / v; }
            }
        }
        __syncthreads();
        const double pv = pinv[buf][0];
        const double inv = pinv[buf][1];
        // 3. in-place update of this thread's elements.  Only ONE wave holds row p (a wave is two values of ti): the others skip the
        //    row test (wave-uniform branch)
        const bool wave_has_p = ((p % TI) >> 1) == (tid >> 6);
        // column k of the old matrix counts as zero in the update (its values are the multipliers f): its owners clear it first -
        // kb is uniform, so the column index is static inside a scalar branch and the update itself is a bare fma per element
#pragma unroll
        for (int y = 0; y < NBC; ++y) {
            if (y == kb) {
                if (tj == kt) {
#pragma unroll
                    for (int x = 0; x < NBR; ++x) a[x][y] = 0.0;
                }
            }
        }
        double rr[NBC], ff[NBR];
#pragma unroll
        for (int y = 0; y < NBC; ++y) {
            const int j = tj + 32 * y;
            rr[y] = (j == k) ? inv : rowk[buf][j] * inv;                // new pivot row element
        }
#pragma unroll
        for (int x = 0; x < NBR; ++x) ff[x] = fcol[buf][ti + TI * x];
        if (wave_has_p) {
#pragma unroll
            for (int y = 0; y < NBC; ++y)
#pragma unroll
                for (int x = 0; x < NBR; ++x) a[x][y] = (ti + TI * x == p) ? rr[y] : a[x][y] - ff[x] * rr[y];
        } else {
#pragma unroll
            for (int y = 0; y < NBC; ++y)
#pragma unroll
                for (int x = 0; x < NBR; ++x) a[x][y] = a[x][y] - ff[x] * rr[y];
        }
        if (tid == 0) pivlog[k] = pv;                               // (its log after the loop: an fp64 log by ONE thread in every step kept
                                                                    //  the other 511 waiting at the next barrier)
        // the next step writes the other LDS buffers; its first barrier orders this step's reads before step k+2's writes
    }
    __syncthreads();
    for (int k = tid; k < C; k += TI * 32) pivlog[k] = log(fabs(pivlog[k]));
    __syncthreads();
    if (tid == 0) {
        double sum = 0.0;
        for (int k = 0; k < C; ++k) sum += pivlog[k];               // step order: deterministic
        logabsdet[blockIdx.x] = (float)sum;
    }
    if (winv) {
        float* out = winv + (size_t)blockIdx.x * C * C;
#pragma unroll
        for (int x = 0; x < NBR; ++x)
#pragma unroll
            for (int y = 0; y < NBC; ++y) {
                const int i = ti + TI * x, j = tj + 32 * y;
                if (i < C && j < C) out[kof[i] * C + pivrow[j]] = (float)a[x][y];
            }
    }
}

// ---- log|det| only, matrices of DIFFERENT sizes in one launch (all 1x1 convolutions of a Glow forward pass) --------
// LU with implicit partial pivoting on W alone (no identity half): step k picks the unused row with the largest
// |A[r][k]| (lowest r on ties, as above), and every other unused row i gets A[i][j] -= (A[i][k] / pv) * A[p][j] for
// j > k.  Column k and the pivot row are only READ in step k, so nothing is staged: two barriers per step.  The pivots
// are kept and log|.| is taken by C threads in parallel at the end, summed in step order (deterministic).
constexpr int kMaxLogdetMats = 96;
struct LogdetTable {
    const float* w[kMaxLogdetMats];
    int C[kMaxLogdetMats];
};

__global__ void __launch_bounds__(kBlock) logdet_multi_kernel(const LogdetTable tab, float* __restrict__ logabsdet) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int C = tab.C[blockIdx.x];
    const float* __restrict__ wm = tab.w[blockIdx.x];
    const int LD = C + 1;                                          // odd-ish row stride: column reads spread over banks
    double* A = sm;                                                // [C][LD]
    double* piv = A + (size_t)C * LD;                              // [C] pivots, then their logs
    double* wval = piv + C;                                        // [4]
    int* widx = reinterpret_cast<int*>(wval + 4);                  // [4]
    int* used = widx + 4;                                          // [C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < C * C; e += kBlock) A[(e / C) * LD + (e % C)] = (double)wm[e];
    for (int i = tid; i < C; i += kBlock) used[i] = 0;
    __syncthreads();
    for (int k = 0; k < C; ++k) {
        double best = -1.0;
        int bi = 1 << 30;
        if (tid < C && !used[tid]) { best = fabs(A[tid * LD + k]); bi = tid; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { wval[wave] = best; widx[wave] = bi; }
        __syncthreads();
        int p = widx[0];
        double pb = wval[0];
#pragma unroll
        for (int q = 1; q < kBlock / 64; ++q)
            if (wval[q] > pb || (wval[q] == pb && widx[q] < p)) { pb = wval[q]; p = widx[q]; }
        const double pv = A[p * LD + k];
        const double rp = 1.0 / pv;
        const int nc = C - 1 - k;                                  // columns right of k
        // row p is excluded below by i != p, so marking it used now cannot change what this step's updates do
        if (tid == 0) { piv[k] = pv; used[p] = 1; }
        (void)nc;
        // fixed 2-D mapping (32 column lanes x 8 row groups): no runtime div/mod in the update
        for (int i = tid >> 5; i < C; i += kBlock / 32) {
            if (i == p || used[i]) continue;
            const double f = A[i * LD + k] * rp;
            for (int j = k + 1 + (tid & 31); j < C; j += 32) A[i * LD + j] -= f * A[p * LD + j];
        }
        __syncthreads();                                           // updates + used[] visible before the next argmax
    }
    if (tid < C) piv[tid] = log(fabs(piv[tid]));
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int k = 0; k < C; ++k) s += piv[k];
        logabsdet[blockIdx.x] = (float)s;
    }
}


// max over the 64 lanes of a wave without the LDS crossbar: prefix-max inside each row of 16 lanes with DPP row shifts
// (lanes shifted in from outside a row read 0, the identity), then the four row results through readlane
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));   // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));   // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));   // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));   // row_shr:8
    const unsigned a = __builtin_amdgcn_readlane((int)v, 15), b = __builtin_amdgcn_readlane((int)v, 31);
    const unsigned c = __builtin_amdgcn_readlane((int)v, 47), d = __builtin_amdgcn_readlane((int)v, 63);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ double readlane_f64(double v, int lane_uniform) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane_uniform);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane_uniform);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Register-resident variant for the channel counts Glow produces: lane = row, the row lives in C fp64 registers (the
// loops are fully unrolled, so every index is static), the pivot row is broadcast with v_readlane - no LDS, no barrier
// anywhere in the elimination.  Returns log|det| (valid in lane 0).
template <int C>
__device__ __forceinline__ double lu_logdet_regs(const float* __restrict__ wm, int lane) {
    double a[C];
#pragma unroll
    for (int j = 0; j < C; ++j) a[j] = lane < C ? (double)wm[lane * C + j] : 0.0;
    bool used = lane >= C;
    double mypiv = 1.0;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const unsigned key = used ? 0u : ((__float_as_uint((float)fabs(a[k])) & ~63u) | (unsigned)(63 - lane));
        const int p = 63 - (int)(wave_umax(key) & 63u);
        const double pv = readlane_f64(a[k], p);
        if (lane == k) mypiv = pv;
        if (lane == p) used = true;
        const double f = used ? 0.0 : a[k] * (1.0 / pv);
#pragma unroll
        for (int j = k + 1; j < C; ++j) a[j] -= f * readlane_f64(a[j], p);
    }
    const double lg = lane < C ? log(fabs(mypiv)) : 0.0;
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < C; ++k) sum += readlane_f64(lg, k);        // step order: deterministic
    return sum;
}

// In-place Gauss-Jordan INVERSION with the same register scheme (lane = physical row, C fp64 registers, pivot row broadcast with
// v_readlane, implicit partial pivoting: no row moves).  Step k with pivot row p = p_k and pivot value pv: row p becomes row k of the
// working inverse (a[k] := 1, then the row is divided by pv), every other row r subtracts a_r[k] times it (a_r[k] := 0 first).  With
// B = P A the row-permuted matrix (B[k] = A[p_k]) the registers end as B^-1, logical row i in physical row p_i, and A^-1 = B^-1 P:
//   A^-1[i][p_k] = B^-1[i][k]  ->  lane r (pivot of step `mystep`) writes winv[mystep][p_k] = a[k] for every k.
// The three-barriers-per-step workgroup kernel took ~100 us per level of 16 matrices whatever their size (3 launches on the critical
// path of a training step); this one is one wave per matrix without a barrier.  Returns log|det| (valid in every lane).
template <int C>
__device__ __forceinline__ double gj_inverse_regs(const float* __restrict__ wm, float* __restrict__ winv, int lane, int* prow_s) {
    double a[C];
#pragma unroll
    for (int j = 0; j < C; ++j) a[j] = lane < C ? (double)wm[lane * C + j] : 0.0;
    bool used = lane >= C;
    int mystep = -1;
    double mypiv = 1.0;                                            // lane k keeps the pivot of step k: the logs are taken after the loop,
#pragma unroll                                                     // one per lane (an fp64 log per step sat on the serial chain)
    for (int k = 0; k < C; ++k) {
        const unsigned key = used ? 0u : ((__float_as_uint((float)fabs(a[k])) & ~63u) | (unsigned)(63 - lane));
        const int p = 63 - (int)(wave_umax(key) & 63u);
        if (lane == 0) prow_s[k] = p;                              // (LDS: a register array of the 48 pivot rows spilled)
        const double pv = readlane_f64(a[k], p);
        const double rp = 1.0 / pv;
        if (lane == k) mypiv = pv;
        const bool mine = lane == p;
        if (mine) { used = true; mystep = k; }
        // one update formula for every row, a[j] -= f * pj with pj = a_p[j] / pv the normalised pivot row: the other rows use f = a[k];
        // the pivot row itself f = pv - 1 (a_p[j] - (pv - 1) a_p[j] / pv = a_p[j] / pv), no per-element select.  Column k: the pivot
        // row gets 1 / pv, the others -a[k] / pv.
        const double f = mine ? pv - 1.0 : a[k];
        const double ak = a[k];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const double pj = readlane_f64(a[j], p) * rp;
            a[j] = j == k ? (mine ? rp : -ak * rp) : a[j] - f * pj;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (winv != nullptr && lane < C) {
#pragma unroll
        for (int k = 0; k < C; ++k) winv[mystep * C + prow_s[k]] = (float)a[k];
    }
    const double lg = lane < C ? log(fabs(mypiv)) : 0.0;
    double logdet = 0.0;
#pragma unroll
    for (int k = 0; k < C; ++k) logdet += readlane_f64(lg, k);     // step order: the sum the per-step logs gave
    return logdet;
}

__global__ void __launch_bounds__(64) gj_wave_kernel(const float* __restrict__ w, int C, float* __restrict__ logabsdet,
                                                     float* __restrict__ winv) {
    const float* wm = w + (size_t)blockIdx.x * C * C;
    float* out = winv ? winv + (size_t)blockIdx.x * C * C : nullptr;
    const int lane = threadIdx.x;
    __shared__ int prow_s[64];
    double r = 0.0;
    switch (C) {                                                   // wave-uniform
        case 4: r = gj_inverse_regs<4>(wm, out, lane, prow_s); break;
        case 8: r = gj_inverse_regs<8>(wm, out, lane, prow_s); break;
        case 12: r = gj_inverse_regs<12>(wm, out, lane, prow_s); break;
        case 16: r = gj_inverse_regs<16>(wm, out, lane, prow_s); break;
        case 24: r = gj_inverse_regs<24>(wm, out, lane, prow_s); break;
        case 48: r = gj_inverse_regs<48>(wm, out, lane, prow_s); break;
        default: break;
    }
    if (lane == 0) logabsdet[blockIdx.x] = (float)r;
}
static bool gj_wave_ok(int C) { return C == 4 || C == 8 || C == 12 || C == 16 || C == 24 || C == 48; }

// The same for matrices of DIFFERENT sizes in one launch (the inverses of all levels of a model: a 3-level model's inverse pass ran
// three launches, 94 + 20 + 20 us, one in front of each level; together they take as long as the largest).
constexpr int kMaxGjMats = 96;
struct GjTable {
    const float* w[kMaxGjMats];
    float* winv[kMaxGjMats];
    int C[kMaxGjMats];
};
__global__ void __launch_bounds__(64) gj_wave_multi_kernel(const GjTable tab, float* __restrict__ logabsdet) {
    const float* wm = tab.w[blockIdx.x];
    float* out = tab.winv[blockIdx.x];
    const int C = tab.C[blockIdx.x];
    const int lane = threadIdx.x;
    __shared__ int prow_s[64];
    double r = 0.0;
    switch (C) {                                                   // wave-uniform
        case 4: r = gj_inverse_regs<4>(wm, out, lane, prow_s); break;
        case 8: r = gj_inverse_regs<8>(wm, out, lane, prow_s); break;
        case 12: r = gj_inverse_regs<12>(wm, out, lane, prow_s); break;
        case 16: r = gj_inverse_regs<16>(wm, out, lane, prow_s); break;
        case 24: r = gj_inverse_regs<24>(wm, out, lane, prow_s); break;
        case 48: r = gj_inverse_regs<48>(wm, out, lane, prow_s); break;
        default: break;
    }
    if (lane == 0) logabsdet[blockIdx.x] = (float)r;
}

// One WAVE per matrix (C <= 64): lane = row, so a step needs no workgroup barrier - the argmax is one DPP wave reduction,
// the pivot row is read as LDS broadcasts, every lane updates its own row.  LDS ops of one wave execute in order.
__global__ void __launch_bounds__(64) logdet_wave_kernel(const LogdetTable tab, float* __restrict__ logabsdet) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int C = tab.C[blockIdx.x];
    const float* __restrict__ wm = tab.w[blockIdx.x];
    const int lane = threadIdx.x;
    {
        double r = 0.0;
        bool done = true;
        switch (C) {                                               // wave-uniform
            case 4: r = lu_logdet_regs<4>(wm, lane); break;
            case 8: r = lu_logdet_regs<8>(wm, lane); break;
            case 12: r = lu_logdet_regs<12>(wm, lane); break;
            case 16: r = lu_logdet_regs<16>(wm, lane); break;
            case 24: r = lu_logdet_regs<24>(wm, lane); break;
            case 48: r = lu_logdet_regs<48>(wm, lane); break;
            default: done = false;
        }
        if (done) {
            if (lane == 0) logabsdet[blockIdx.x] = (float)r;
            return;
        }
    }
    const int LD = C + 1;
    double* A = sm;                                                // [C][LD]   (generic sizes: rows in LDS)
    for (int e = lane; e < C * C; e += 64) A[(e / C) * LD + (e % C)] = (double)wm[e];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    bool used = lane >= C;                                         // lanes beyond the matrix never pivot
    double* row = A + (lane < C ? lane : 0) * LD;
    double* piv = A + (size_t)C * LD;                              // [C] pivots: their logs are taken in parallel afterwards
    for (int k = 0; k < C; ++k) {
        // pivot = unused row with the largest |A[r][k]|: one 32-bit wave max over the key (fp32 magnitude bits, low 6
        // bits replaced by 63 - lane: lowest row wins among magnitudes equal to ~1e-5 relative, enough for stability)
        const unsigned key = used ? 0u : ((__float_as_uint((float)fabs(row[k])) & ~63u) | (unsigned)(63 - lane));
        const int p = 63 - (int)(wave_umax(key) & 63u);
        const double* prow = A + p * LD;
        const double pv = prow[k];
        if (lane == p) { used = true; piv[k] = pv; }
        // row and prow alias the same LDS array, so a plain loop waits for every store before the next load: load 8
        // columns of both rows into registers, then update and store (the pivot row itself is never written in step k)
        const double f = used ? 0.0 : row[k] * (1.0 / pv);
        for (int j0 = k + 1; j0 < C; j0 += 8) {
            double r[8], q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u < C ? j0 + u : C - 1;
                r[u] = row[j];
                q[u] = prow[j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (!used && j0 + u < C) row[j0 + u] = r[u] - f * q[u];
        }
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (lane < C) piv[lane] = log(fabs(piv[lane]));                // one log per lane, off the serial chain
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        double sum = 0.0;
        for (int k = 0; k < C; ++k) sum += piv[k];                 // step order: deterministic
        logabsdet[blockIdx.x] = (float)sum;
    }
}

}  // namespace

extern "C" size_t nfdp_invconv_prepare_ws_bytes(int n, int C) {
    if (n <= 0 || C <= kMaxLdsC) return 16;    // LDS-resident; a token allocation keeps callers uniform
    return (size_t)n * C * 2 * C * sizeof(double);
}

extern "C" int nfdp_invconv_prepare_f32(const float* w, int n, int C, float* logabsdet, float* winv, void* ws,
                                        nfdp_stream_t stream) {
    NFDP_REQUIRE(w && logabsdet, NFDP_EINVAL, "invconv_prepare: null pointer");
    NFDP_REQUIRE(n >= 0 && C > 0 && C <= 256, NFDP_ESHAPE, "invconv_prepare: bad shape n=%d C=%d (C<=256)", n, C);
    if (n == 0) return NFDP_OK;
    const size_t red = (size_t)(3 * C + 4) * sizeof(double) + (size_t)(2 * C + 4) * sizeof(int) + 16;
    static const bool wave_off = [] { const char* e = NFDP_DIAG_ENV("NFDP_GJ_WAVE"); return e && e[0] == '0'; }();   // A/B measurements
    if (gj_wave_ok(C) && !wave_off) {
        hipLaunchKernelGGL(gj_wave_kernel, dim3(n), dim3(64), 0, as_stream(stream), w, C, logabsdet, winv);
    } else if (C <= kMaxLdsC) {
        const size_t lds = (size_t)C * 2 * C * sizeof(double) + red;
        hipLaunchKernelGGL((gauss_jordan_kernel<true>), dim3(n), dim3(kBlock), lds, as_stream(stream), w, C, logabsdet,
                           winv, nullptr);
    } else if (C <= 96) {
        hipLaunchKernelGGL((gauss_jordan_reg_kernel<32, 3, 3>), dim3(n), dim3(1024), 0, as_stream(stream), w, C, logabsdet, winv);
    } else if (C <= 128) {
        hipLaunchKernelGGL((gauss_jordan_reg_kernel<32, 4, 4>), dim3(n), dim3(1024), 0, as_stream(stream), w, C, logabsdet, winv);
    } else if (C <= 192) {
        // (32 x 32 threads with 6 x 6 elements each: 5.9 us per pivot step; 16 x 32 threads with 12 x 6: 6.1 - the step is a latency chain
        //  of two 16-wave barriers, the candidate scan and an fp64 division, not its 72 or 36 updates per thread)
        hipLaunchKernelGGL((gauss_jordan_reg_kernel<32, 6, 6>), dim3(n), dim3(1024), 0, as_stream(stream), w, C, logabsdet, winv);
    } else {
        NFDP_REQUIRE(ws, NFDP_EINVAL, "invconv_prepare: workspace required for C=%d", C);
        hipLaunchKernelGGL((gauss_jordan_kernel<false>), dim3(n), dim3(kBlock), red, as_stream(stream), w, C, logabsdet,
                           winv, reinterpret_cast<double*>(ws));
    }
    NFDP_CHECK_LAUNCH("invconv_prepare");
    return NFDP_OK;
}

extern "C" int nfdp_invconv_inverses_supported(int C) { return gj_wave_ok(C) ? 1 : 0; }

extern "C" int nfdp_invconv_inverses_f32(const float* const* w, const int* C, int n, float* logabsdet, float* const* winv,
                                         nfdp_stream_t stream) {
    NFDP_REQUIRE(w && C && logabsdet && winv, NFDP_EINVAL, "invconv_inverses: null pointer");
    NFDP_REQUIRE(n >= 0, NFDP_ESHAPE, "invconv_inverses: bad count");
    // w, C, winv: HOST arrays (device pointers / sizes); the kernel receives them by value, kMaxGjMats per launch
    for (int done = 0; done < n; done += kMaxGjMats) {
        GjTable tab;
        const int m = (n - done) < kMaxGjMats ? (n - done) : kMaxGjMats;
        for (int i = 0; i < m; ++i) {
            NFDP_REQUIRE(w[done + i] && winv[done + i] && gj_wave_ok(C[done + i]), NFDP_ESHAPE,
                         "invconv_inverses: matrix %d: null or C=%d not one of 4, 8, 12, 16, 24, 48", done + i, C[done + i]);
            tab.w[i] = w[done + i];
            tab.winv[i] = winv[done + i];
            tab.C[i] = C[done + i];
        }
        hipLaunchKernelGGL(gj_wave_multi_kernel, dim3(m), dim3(64), 0, as_stream(stream), tab, logabsdet + done);
    }
    NFDP_CHECK_LAUNCH("invconv_inverses");
    return NFDP_OK;
}

extern "C" int nfdp_invconv_logdets_f32(const float* const* w, const int* C, int n, float* logabsdet, nfdp_stream_t stream) {
    NFDP_REQUIRE(w && C && logabsdet, NFDP_EINVAL, "invconv_logdets: null pointer");
    NFDP_REQUIRE(n >= 0, NFDP_ESHAPE, "invconv_logdets: bad count");
    // w, C: HOST arrays (device pointers / sizes); the kernel receives them by value, kMaxLogdetMats per launch
    for (int done = 0; done < n; done += kMaxLogdetMats) {
        LogdetTable tab;
        const int m = (n - done) < kMaxLogdetMats ? (n - done) : kMaxLogdetMats;
        int cmax = 0;
        for (int i = 0; i < m; ++i) {
            NFDP_REQUIRE(w[done + i] && C[done + i] > 0 && C[done + i] <= 96, NFDP_ESHAPE,
                         "invconv_logdets: matrix %d: null or C=%d outside [1,96]", done + i, C[done + i]);
            tab.w[i] = w[done + i];
            tab.C[i] = C[done + i];
            cmax = C[done + i] > cmax ? C[done + i] : cmax;
        }
        for (int i = m; i < kMaxLogdetMats; ++i) { tab.w[i] = nullptr; tab.C[i] = 0; }
        const size_t lds = ((size_t)cmax * (cmax + 1) + cmax + 4) * sizeof(double) + (size_t)(cmax + 4) * sizeof(int) + 16;
        if (cmax <= 64)        // one wave per matrix, no barriers
            hipLaunchKernelGGL(logdet_wave_kernel, dim3(m), dim3(64), lds, as_stream(stream), tab, logabsdet + done);
        else
            hipLaunchKernelGGL(logdet_multi_kernel, dim3(m), dim3(kBlock), lds, as_stream(stream), tab, logabsdet + done);
    }
    NFDP_CHECK_LAUNCH("invconv_logdets");
    return NFDP_OK;
}
