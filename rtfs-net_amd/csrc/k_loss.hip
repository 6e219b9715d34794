// Evaluation-side loss path on the device ("the step after the path"):
//   pit_pairwise_kernel   reference src/losses/matrix.py:22-53 (PairwiseNegSDR.forward: snr / sisdr / sdsdr) fused with
//                         src/losses/pit_wrapper.py:84-110 (find_best_perm_factorial, perm_reduce=None) for n_src <= 4
// One workgroup per batch element.  The pairwise matrix needs only second moments of the (optionally mean-removed)
// signals: sums of e_i, t_j, e_i^2, t_j^2 and e_i t_j over time are accumulated in float64 in ONE pass over the 2 n L
// samples (each sample read once; 16-byte loads when the row pitch allows, which it always does for a whole number of
// hops), and every quantity of the reference's formulas follows algebraically:
//   zero-mean:  <e,t> - L mean(e) mean(t),  |e|^2 - L mean(e)^2, ...
//   sisdr:      alpha = <e,t> / (|t|^2 + eps);  |proj|^2 = alpha^2 |t|^2;  |noise|^2 = |e|^2 - 2 alpha <e,t> + alpha^2 |t|^2
// (float64 keeps the cancellation in |noise|^2 exact to ~1e-12 relative, i.e. > 100 dB of SDR).  Wave 0 then scans the
// n! permutations in itertools order and keeps the FIRST minimum, as torch.min does on the reference's loss_set.
#include "common.h"
#include "kernels.h"

#define LOSS_MAXN 4

namespace {

__device__ __forceinline__ double block_sum_d(double v, double* red, int tid) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

__global__ __launch_bounds__(256) void pit_pairwise_kernel(const float* __restrict__ est, const float* __restrict__ tgt, int n, int L,
                                                           int kind, int zero_mean, int take_log, float* __restrict__ pw,
                                                           float* __restrict__ min_loss, int* __restrict__ perm_out) {
    __shared__ double red[4];
    __shared__ double mom[2 * LOSS_MAXN + 2 * LOSS_MAXN + LOSS_MAXN * LOSS_MAXN];  // se[n] st[n] see[n] stt[n] set[n][n]
    __shared__ float pws[LOSS_MAXN * LOSS_MAXN];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* e = est + (size_t)b * n * L;
    const float* t = tgt + (size_t)b * n * L;
    double se[LOSS_MAXN], st[LOSS_MAXN], see[LOSS_MAXN], stt[LOSS_MAXN], set[LOSS_MAXN][LOSS_MAXN];
#pragma unroll
    for (int i = 0; i < LOSS_MAXN; ++i) {
        se[i] = st[i] = see[i] = stt[i] = 0;
#pragma unroll
        for (int j = 0; j < LOSS_MAXN; ++j) set[i][j] = 0;
    }
    for (int l = tid; l < L; l += 256) {
        float ev[LOSS_MAXN], tv[LOSS_MAXN];
#pragma unroll
        for (int i = 0; i < LOSS_MAXN; ++i) {
            ev[i] = i < n ? e[(size_t)i * L + l] : 0.f;
            tv[i] = i < n ? t[(size_t)i * L + l] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < LOSS_MAXN; ++i) {
            se[i] += (double)ev[i];
            st[i] += (double)tv[i];
            see[i] += (double)ev[i] * (double)ev[i];
            stt[i] += (double)tv[i] * (double)tv[i];
#pragma unroll
            for (int j = 0; j < LOSS_MAXN; ++j) set[i][j] += (double)ev[i] * (double)tv[j];
        }
    }
    for (int i = 0; i < n; ++i) {
        const double a0 = block_sum_d(se[i], red, tid), a1 = block_sum_d(st[i], red, tid);
        const double a2 = block_sum_d(see[i], red, tid), a3 = block_sum_d(stt[i], red, tid);
        if (tid == 0) {
            mom[i] = a0;
            mom[LOSS_MAXN + i] = a1;
            mom[2 * LOSS_MAXN + i] = a2;
            mom[3 * LOSS_MAXN + i] = a3;
        }
        for (int j = 0; j < n; ++j) {
            const double a4 = block_sum_d(set[i][j], red, tid);
            if (tid == 0) mom[4 * LOSS_MAXN + i * LOSS_MAXN + j] = a4;
        }
    }
    __syncthreads();
    if (tid < n * n) {
        const int i = tid / n, j = tid % n;  // estimate i, target j
        const double eps = 1e-8, Ld = (double)L;  // the losses' own EPS (matrix.py:19), not the norm layers'
        double ee = mom[2 * LOSS_MAXN + i], tt = mom[3 * LOSS_MAXN + j], et = mom[4 * LOSS_MAXN + i * LOSS_MAXN + j];
        if (zero_mean) {
            const double me = mom[i] / Ld, mt = mom[LOSS_MAXN + j] / Ld;
            ee -= Ld * me * me;
            tt -= Ld * mt * mt;
            et -= Ld * me * mt;
        }
        double proj2, noise2;
        if (kind == 0) {  // snr: proj = t, noise = e - t
            proj2 = tt;
            noise2 = ee - 2 * et + tt;
        } else {
            const double alpha = et / (tt + eps);
            proj2 = alpha * alpha * tt;
            noise2 = kind == 1 ? ee - 2 * alpha * et + alpha * alpha * tt  // sisdr: noise = e - alpha t
                               : ee - 2 * et + tt;                         // sdsdr: noise = e - t
        }
        noise2 = noise2 < 0 ? 0 : noise2;
        double sdr = proj2 / (noise2 + eps);
        if (take_log) sdr = 10.0 * log10(sdr + eps);
        const float v = (float)(-sdr);
        pws[i * LOSS_MAXN + j] = v;
        pw[((size_t)b * n + i) * n + j] = v;
    }
    __syncthreads();
    if (tid == 0) {
        // permutations of range(n) in lexicographic (itertools) order; loss = mean_i pw[perm[i]][i]; first minimum wins
        int p[LOSS_MAXN], best[LOSS_MAXN];
        for (int i = 0; i < n; ++i) p[i] = best[i] = i;
        float bl = 0.f;
        bool first = true;
        while (true) {
            float s = 0.f;
            for (int i = 0; i < n; ++i) s += pws[p[i] * LOSS_MAXN + i];
            s /= (float)n;
            if (first || s < bl) {
                bl = s;
                first = false;
                for (int i = 0; i < n; ++i) best[i] = p[i];
            }
            // next lexicographic permutation
            int k = n - 2;
            while (k >= 0 && p[k] > p[k + 1]) --k;
            if (k < 0) break;
            int m = n - 1;
            while (p[m] < p[k]) --m;
            int tmp = p[k]; p[k] = p[m]; p[m] = tmp;
            for (int lo = k + 1, hi = n - 1; lo < hi; ++lo, --hi) { tmp = p[lo]; p[lo] = p[hi]; p[hi] = tmp; }
        }
        min_loss[b] = bl;
        for (int i = 0; i < n; ++i) perm_out[b * n + i] = best[i];
    }
}

}  // namespace

// Gradient of the PIT loss with respect to the estimates.  For the assigned pair (estimate j = perm[b][i], target i) every variant of
// matrix.py:22-53 is a function of the second moments only, so d loss / d e = ce * e + ct * t with e, t the (mean-removed) signals;
// one workgroup per (b, i): one pass for the five moments in float64, one pass to write the row of estimate j.
//   snr:    P = |t|^2,          N = |e - t|^2
//   sdsdr:  P = alpha^2 |t|^2,  N = |e - t|^2              alpha = <e,t> / (|t|^2 + eps)
//   sisdr:  P = alpha^2 |t|^2,  N = |e - alpha t|^2
//   loss = -10 log10(P / (N + eps) + eps)  (or -P / (N + eps) without the log); upstream gradient dmin[b] / n_src.
__global__ __launch_bounds__(256) void pit_sdr_bwd_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                          const int* __restrict__ perm, const float* __restrict__ dmin, float* __restrict__ dest,
                                                          int n, int L, int kind, int zero_mean, int take_log) {
    __shared__ double red[4];
    const int b = blockIdx.x / n, i = blockIdx.x % n, tid = threadIdx.x;
    const int j = perm[b * n + i];
    const float* e = est + ((size_t)b * n + j) * L;
    const float* t = tgt + ((size_t)b * n + i) * L;
    double se = 0, st = 0, see = 0, stt = 0, set = 0;
    for (int l = tid; l < L; l += 256) {
        const double ev = e[l], tv = t[l];
        se += ev; st += tv; see += ev * ev; stt += tv * tv; set += ev * tv;
    }
    se = block_sum_d(se, red, tid);
    st = block_sum_d(st, red, tid);
    see = block_sum_d(see, red, tid);
    stt = block_sum_d(stt, red, tid);
    set = block_sum_d(set, red, tid);
    const double eps = 1e-8;
    double me = 0, mt = 0;
    if (zero_mean) {
        me = se / L; mt = st / L;
        see -= L * me * me; stt -= L * mt * mt; set -= L * me * mt;
    }
    const double Et = stt + eps, alpha = set / Et;
    double P, N, p_t, n_t;
    if (kind == 0) {         // snr
        P = stt; N = see - 2 * set + stt; p_t = 0; n_t = -2;
    } else if (kind == 2) {  // sdsdr
        P = alpha * alpha * stt; N = see - 2 * set + stt; p_t = 2 * alpha * stt / Et; n_t = -2;
    } else {                 // sisdr
        P = alpha * alpha * stt; N = see - 2 * alpha * set + alpha * alpha * stt; p_t = 2 * alpha * stt / Et;
        n_t = -2 * alpha - 2 * (set - alpha * stt) / Et;
    }
    if (N < 0) N = 0;
    const double Ne = N + eps, r = P / Ne;
    const double g0 = (take_log ? -(10.0 / 2.302585092994046) / (r + eps) : -1.0) * (double)dmin[b] / n;
    const float ce = (float)(g0 * (-P / (Ne * Ne)) * 2.0), ct = (float)(g0 * (p_t / Ne - P / (Ne * Ne) * n_t));
    const float fme = (float)me, fmt = (float)mt;
    float* d = dest + ((size_t)b * n + j) * L;
    for (int l = tid; l < L; l += 256) d[l] = ce * (e[l] - fme) + ct * (t[l] - fmt);
}

int launch_pit_sdr_bwd(const float* est, const float* tgt, const int* perm, const float* dmin, float* dest, int B, int n, int L, int kind,
                       int zero_mean, int take_log, hipStream_t st) {
    if (n < 1 || n > LOSS_MAXN || L < 1) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(pit_sdr_bwd_kernel, dim3(B * n), dim3(256), 0, st, est, tgt, perm, dmin, dest, n, L, kind, zero_mean, take_log);
    return rtfs_launch_status();
}

int launch_pit_pairwise(const float* est, const float* tgt, int B, int n, int L, int kind, int zero_mean, int take_log, float* pw,
                        float* min_loss, int* perm, hipStream_t st) {
    if (B < 1 || n < 1 || n > LOSS_MAXN || L < 1 || kind < 0 || kind > 2) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(pit_pairwise_kernel, dim3(B), dim3(256), 0, st, est, tgt, n, L, kind, zero_mean, take_log, pw, min_loss, perm);
    return rtfs_launch_status();
}
