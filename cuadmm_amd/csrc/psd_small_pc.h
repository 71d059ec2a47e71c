// Producer/consumer variant of the register-resident projection kernel (see psd_small_reg.h for the
// layout and the phases).  Difference: the implicit-QL scalar recurrence is executed once per block by a
// producer wavefront (one block per lane) instead of redundantly by every lane of the block, and the
// other wavefronts only apply the published plane rotations to their Z rows.
#pragma once
#include <hip/hip_runtime.h>

#include "psd_device.h"
#include "psd_small_reg.h"

namespace cuadmm {

template <int NMAX>
struct PcLayout {
  static constexpr int LD = NMAX + 1;
  static constexpr int kTile = NMAX * LD;
  static constexpr int kVec = 2 * NMAX;        // vv / ww during tridiagonalisation; (c,s) double buffer during QL
  static constexpr int kDE = NMAX + 1;         // entry NMAX of D / E carries the sweep descriptor of buffer 0 / 1
  static constexpr int kMisc = 2;
  static constexpr int kPer = kTile + 2 * kVec + 2 * kDE + kMisc;
};

template <int NMAX, int WAVES, int MODE, class Args>
__device__ __forceinline__ void psd_small_pc_body(const Args& a, double* smem, int* flags) {
  using Gp = SubGroup<NMAX>;
  using Lay = PcLayout<NMAX>;
  constexpr int LD = Lay::LD;
  constexpr int BPW = 64 / NMAX;
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  constexpr int NB = WAVES * BPW;                 // blocks per workgroup
  const int wg_slot0 = (int)blockIdx.x * NB;
  const int slot0 = wg_slot0 + wave * BPW;
  double* wsm = smem + wave * BPW * Lay::kPer;   // this wavefront's block regions
  long long* dbg = a.dbg ? a.dbg + ((long long)blockIdx.x * WAVES + wave) * 8 : nullptr;
#define CUADMM_STAMP(i) do { if (dbg && lane == 0) dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
  CUADMM_STAMP(0);

  // ---- cooperative coalesced load of the wavefront's blocks into their LDS tiles ---------------
  for (int gg = 0; gg < BPW; ++gg) {
    const int slot = slot0 + gg;
    if (slot >= a.count) break;
    const int bi = a.ids ? a.ids[slot] : slot;
    double* Tg = wsm + gg * Lay::kPer;
    if (MODE == 0) {
      const int n = a.bn[bi];
      const double* src = a.in + a.boff[bi];
      const int len = n * (n + 1) / 2;
      for (int e = lane; e < len; e += 64) {
        int i, j;
        tri_decode(e, i, j);
        double v = src[e];
        if (i != j) v *= kSqrt2Inv;
        Tg[j * LD + i] = v;
        Tg[i * LD + j] = v;
      }
    } else {
      const int n = a.n_uniform;
      const double* src = a.in + (long long)bi * n * n;
      for (int idx = lane; idx < n * n; idx += 64) {
        const int c = idx / n, r = idx - c * n;
        if (r >= c) {
          const double v = src[idx];
          Tg[r * LD + c] = v;
          Tg[c * LD + r] = v;
        }
      }
    }
  }
  wave_fence();

  const int g = lane / NMAX;
  const int rank = lane & (NMAX - 1);
  const int slot = slot0 + g;
  const bool valid = slot < a.count;
  const int bi = valid ? (a.ids ? a.ids[slot] : slot) : 0;
  const int n = valid ? ((MODE == 0) ? a.bn[bi] : a.n_uniform) : 0;
  double* T = wsm + g * Lay::kPer;
  double* vv = T + Lay::kTile;
  double* ww = vv + Lay::kVec;
  double* D = ww + Lay::kVec;
  double* E = D + Lay::kDE;
  double* misc = E + Lay::kDE;
  const int half_base = lane & ~(NMAX - 1);

  double ar[NMAX], q[NMAX];
  double eps_abs = 0.0;
  if (valid) {
#pragma unroll
  for (int c = 0; c < NMAX; ++c) {
    ar[c] = (rank < n && c < n) ? T[rank * LD + c] : 0.0;
    q[c] = (c == rank) ? 1.0 : 0.0;
  }
  vv[rank] = 0.0; vv[rank + NMAX] = 0.0;
  ww[rank] = 0.0; ww[rank + NMAX] = 0.0;
  wave_fence();
  CUADMM_STAMP(1);

  // ---- Householder tridiagonalisation, Z accumulated on the fly ---------------------------------
  for (int k = 0; k < n - 2; ++k) {
    const double x = ar[0];                                    // A[rank][k]
    const double xn2 = Gp::sum((rank >= k + 2) ? x * x : 0.0, nullptr);
    const double alpha = __shfl(x, half_base + k + 1, 64);
    if (rank == k) D[k] = x;
    double t = 0.0, beta = alpha, scal = 0.0;
    if (xn2 != 0.0) {
      double nrm, inrm;
      fast_sqrt_rsqrt(alpha * alpha + xn2, nrm, inrm);
      beta = -copysign(nrm, alpha);
      t = (beta - alpha) * (-copysign(inrm, alpha));          // (beta - alpha) / beta
      scal = fast_rcp(alpha - beta);
    }
    if (rank == k + 1) E[k] = beta;
    if (t != 0.0) {
      const double vr = (rank >= k + 2) ? x * scal : ((rank == k + 1) ? 1.0 : 0.0);
      vv[rank] = vr;
      wave_fence();
      // p = t * A(k+1:, k+1:) v   (ar[j] = A[rank][k+j]; vv[c] = 0 for c <= k)
      double p0 = 0.0, p1 = 0.0;
      const double* vs = vv + k;
#pragma unroll
      for (int j = 0; j < NMAX; j += 2) { p0 += ar[j] * vs[j]; p1 += ar[j + 1] * vs[j + 1]; }
      const double pr = (rank >= k + 1) ? (p0 + p1) * t : 0.0;
      const double K = Gp::sum(pr * vr, nullptr) * (-0.5 * t);
      const double wr = pr + K * vr;
      ww[rank] = wr;
      // Z <- Z H_k  (row-owner: s = t * <z_row, v>)
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int c = 0; c < NMAX; c += 2) { s0 += q[c] * vv[c]; s1 += q[c + 1] * vv[c + 1]; }
      const double sr = (s0 + s1) * t;
#pragma unroll
      for (int c = 0; c < NMAX; ++c) q[c] -= sr * vv[c];
      wave_fence();
      const double* wsft = ww + k;
#pragma unroll
      for (int j = 0; j < NMAX; ++j) ar[j] -= vr * wsft[j] + wr * vs[j];
      wave_fence();
    }
#pragma unroll
    for (int j = 0; j + 1 < NMAX; ++j) ar[j] = ar[j + 1];
    ar[NMAX - 1] = 0.0;
  }
  {
    const int kk = n >= 2 ? n - 2 : 0;                          // columns already rotated out
    if (n >= 2) {
      if (rank == kk) D[kk] = ar[0];
      if (rank == kk + 1) { E[kk] = ar[0]; D[kk + 1] = ar[1]; }
    } else if (rank == 0) {
      D[0] = ar[0];
    }
    if (rank == 0) E[n - 1] = 0.0;
  }
  wave_fence();

  CUADMM_STAMP(2);
  // deflation threshold relative to ||T||_F (see psd_device.h)
  {
    const double dv = rank < n ? D[rank] : 0.0, ev = rank < n ? E[rank] : 0.0;
    eps_abs = sqrt(Gp::sum(dv * dv + 2.0 * ev * ev, nullptr)) * 0x1p-53;
  }
  }  // valid (load + tridiagonalisation)

  // ---- implicit QL, producer / consumer -----------------------------------------------------------
  // The scalar recurrence of a sweep does not depend on Z.  The last wavefront of the workgroup runs it
  // for ALL blocks of the workgroup, one block per lane (no redundancy across lanes), and publishes the
  // plane rotations (c, s) of the sweep in LDS; every wavefront then applies them to its own Z rows in
  // registers.  Sweeps advance in lock step, double buffered: while sweep t is applied, sweep t+1 is
  // being produced.  One workgroup barrier per sweep.
  if (valid && rank == 0) { misc[0] = eps_abs; }
  __syncthreads();
  const bool is_producer = (wave == WAVES - 1);
  // producer: LPP consecutive lanes serve one block of the workgroup (they run the recurrence redundantly, which
  // costs nothing extra, and share the search for the deflation point)
  constexpr int LPP = 64 / NB;
  static_assert(LPP >= 1 && LPP * NB == 64, "workgroup blocks must divide the wavefront");
  const int pb = lane / LPP;
  const int pt = lane % LPP;
  const bool p_act = is_producer && (wg_slot0 + pb) < a.count;
  int p_n = 0, p_l = 0, p_sweeps = 0, p_fail = 0;
  bool p_done = true;
  // LDS regions of "my" block as a producer lane (index clamped so that the address space stays LDS for every lane)
  double* const pbase = smem + pb * Lay::kPer;
  double* const pCS0 = pbase + Lay::kTile;
  double* const pCS1 = pCS0 + Lay::kVec;
  double* const pD = pCS1 + Lay::kVec;
  double* const pE = pD + Lay::kDE;
  double p_eps = 0.0;
  int p_bi = 0;
  if (p_act) {
    const int pslot = wg_slot0 + pb;
    p_bi = a.ids ? a.ids[pslot] : pslot;
    p_n = (MODE == 0) ? a.bn[p_bi] : a.n_uniform;
    p_eps = (pE + Lay::kDE)[0];
    p_done = (p_n <= 1);
  }
  long long pt_scan = 0, pt_loop = 0, pn_slots = 0;
  auto produce = [&](int buf) {
    const long long pc0 = dbg ? (long long)__builtin_readcyclecounter() : 0;
    // returns through LDS: descriptor {l_eff, m} of this sweep (empty when l_eff >= m) and CS[i] for l_eff <= i < m
    int* desc = reinterpret_cast<int*>(buf == 0 ? pD + NMAX : pE + NMAX);
    double* CS = buf == 0 ? pCS0 : pCS1;
    int dl_eff = 1, dm = 0;
    if (!p_done) {
      int m;
      for (;;) {
        // first index >= l with a negligible off-diagonal: lane pt tests l+pt, l+pt+LPP, ...; one ballot per round
        m = p_n - 1;
        for (int base = p_l; base < p_n - 1; base += LPP) {
          const int idx = base + pt;
          bool small = false;
          if (idx < p_n - 1) {
            const double ae = fabs(pE[idx]);
            const double dd = fabs(pD[idx]) + fabs(pD[idx + 1]);
            small = (ae <= p_eps || ae + dd == dd);
          }
          const unsigned long long bits = (__ballot(small) >> (pb * LPP)) & ((1ull << LPP) - 1ull);
          if (bits) { m = base + (int)__builtin_ctzll(bits); break; }
        }
        if (m > p_l) break;
        ++p_l; p_sweeps = 0;
        if (p_l >= p_n) { p_done = true; break; }
      }
      const long long pc1 = dbg ? (long long)__builtin_readcyclecounter() : 0;
      pt_scan += pc1 - pc0;
      if (!p_done && p_sweeps++ >= kQlMaxSweepsPerEig) { p_fail = 1; p_done = true; }
      if (!p_done) {
        const int l = p_l;
        const double dl = pD[l], el = pE[l];
        double gq = (pD[l + 1] - dl) * fast_rcp(el + el);
        double r0, r0i;
        fast_sqrt_rsqrt(fma(gq, gq, 1.0), r0, r0i);
        gq = pD[m] - dl + el * fast_rcp(gq + copysign(r0, gq));
        double s = 1.0, c = 1.0, p = 0.0;
        double e_c = pE[m - 1], d_c = pD[m - 1], d1_c = pD[m];
        int i = m - 1;
        bool broke = false;
        const long long pc2 = dbg ? (long long)__builtin_readcyclecounter() : 0;
        pn_slots += m - l;
        // software pipelined by hand: the LDS stores of slot i are issued at the top of slot i-1, together with the
        // prefetch for slot i-2, so that no LDS operation is younger than one full recurrence step when the next
        // step waits on the counter (the compiler waits for lgkmcnt(0) at the loop header)
        double rr_p = 0.0, dn_p = 0.0, c_p = 1.0, s_p = 0.0;
        int i_p = -1;
        for (; i >= l; --i) {
          if (pt == 0 && i_p >= 0) {
            pE[i_p + 1] = rr_p; pD[i_p + 1] = dn_p;
            reinterpret_cast<double2*>(CS)[i_p] = make_double2(c_p, s_p);
          }
          const int ip = i > 0 ? i - 1 : 0;                 // prefetch for slot i-1 (untouched by this sweep)
          const double e_n = pE[ip], d_n = pD[ip];
          const double f = s * e_c, b = c * e_c;
          const double h = fma(f, f, gq * gq);
          if (h == 0.0) {                                   // underflow recovery of the textbook recurrence
            if (pt == 0) { pE[i + 1] = 0.0; pD[i + 1] = d1_c - p; pE[m] = 0.0; }
            broke = true;
            break;
          }
          double rr, rinv;
          fast_sqrt_rsqrt(h, rr, rinv);
          s = f * rinv; c = gq * rinv;
          gq = d1_c - p;
          const double cb = c * b;
          const double r2 = fma(d_c - gq, s, cb + cb);
          p = s * r2;
          rr_p = rr; dn_p = gq + p; c_p = c; s_p = s; i_p = i;
          gq = fma(c, r2, -b);
          d1_c = d_c; e_c = e_n; d_c = d_n;
        }
        if (dbg) pt_loop += (long long)__builtin_readcyclecounter() - pc2;
        if (pt == 0 && i_p >= 0 && !(broke && i_p == i)) {      // flush the last completed slot
          pE[i_p + 1] = rr_p; pD[i_p + 1] = dn_p;
          reinterpret_cast<double2*>(CS)[i_p] = make_double2(c_p, s_p);
        }
        if (!broke && pt == 0) { pD[l] = pD[l] - p; pE[l] = gq; pE[m] = 0.0; }
        dl_eff = i + 1;                                      // == l after a full sweep
        dm = m;
      }
    }
    if (pt == 0) { desc[0] = dl_eff; desc[1] = dm; }
    wave_fence();
  };

  int buf = 0;
  long long t_prod = 0, t_cons = 0, t_bar = 0, n_steps = 0;
  if (is_producer) {
    const bool all = __all(p_done || !p_act);
    if (lane == 0) flags[0] = all ? 1 : 0;
    if (p_act && !all) produce(0);
  }
  __syncthreads();
  while (flags[buf] == 0) {
    long long tt0 = dbg ? (long long)__builtin_readcyclecounter() : 0;
    if (is_producer) {                                        // sweep t+1 into the other buffer
      const bool all = __all(p_done || !p_act);
      if (lane == 0) flags[buf ^ 1] = all ? 1 : 0;
      if (p_act && !all) produce(buf ^ 1);
    }
    long long tt1 = dbg ? (long long)__builtin_readcyclecounter() : 0;
    if (valid) {                                              // apply sweep t to the rows held in registers
      const int* desc = reinterpret_cast<const int*>(buf == 0 ? D + NMAX : E + NMAX);
      const double* CS = buf == 0 ? vv : ww;
      const int l = desc[0], m = desc[1];
      // (c,s) pairs are fetched a chunk ahead of their use (unconditionally: entries outside [l,m) are stale, unused)
      constexpr int CH = 8;
      const double2* CS2 = reinterpret_cast<const double2*>(CS);
#pragma unroll
      for (int hi = NMAX - 2; hi >= 0; hi -= CH) {
        double2 cs[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t) { const int i = hi - t; cs[t] = CS2[i >= 0 ? i : 0]; }
#pragma unroll
        for (int t = 0; t < CH; ++t) {
          const int i = hi - t;
          if (i >= 0 && i < m && i >= l) {
            const double c = cs[t].x, s = cs[t].y;
            const double z0 = q[i], z1 = q[i + 1];
            q[i + 1] = fma(s, z0, c * z1);
            q[i] = fma(c, z0, -(s * z1));
          }
        }
      }
    }
    long long tt2 = dbg ? (long long)__builtin_readcyclecounter() : 0;
    __syncthreads();
    if (dbg) { long long tt3 = (long long)__builtin_readcyclecounter(); t_prod += tt1 - tt0; t_cons += tt2 - tt1; t_bar += tt3 - tt2; ++n_steps; }
    buf ^= 1;
  }
  if (dbg && lane == 0) { dbg[7] = (t_prod << 40) | (t_cons << 20) | t_bar / 16; dbg[6] = n_steps; }
  if (dbg && is_producer && lane == 0) { dbg[0 + 8] = 0; }
  if (dbg && is_producer && lane == 0) { dbg[1] = -pt_scan; dbg[2] = -pt_loop; dbg[3] = -pn_slots; dbg[4] = -n_steps; }
  if (p_act) {
    if (pt == 0) {
      if (MODE == 0) { if (p_fail && a.info) atomicAdd(a.info, 1); }
      else if (a.info) a.info[p_bi] = p_fail;
    }
  }
  if (valid) {
  CUADMM_STAMP(3);
  // ---- output ------------------------------------------------------------------------------------------
  if (MODE == 0) {
    if constexpr (NMAX >= 16) {
      // hand Z (unscaled) and max(d,0) to the matrix-core rebuild that follows (psd_small_reg_rebuild_mfma)
#pragma unroll
      for (int k = 0; k < NMAX; ++k) T[rank * LD + k] = q[k];
      { const double lam = rank < n ? D[rank] : 0.0; vv[rank] = lam > 0.0 ? lam : 0.0; }   // dense_scalar.cu:41-47
    } else {
      // T = Z * diag(max(d,0))   (dense_scalar.cu:41-47, diagonal_batch.cu:11-23)
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        const double lam = (k < n) ? D[k] : 0.0;
        T[rank * LD + k] = q[k] * (lam > 0.0 ? lam : 0.0);
      }
      wave_fence();
      // P = T * Z^T, upper triangle, row a at a time; row a of T is dead once it has been used
      for (int aa = 0; aa < n; ++aa) {
        const double* ta = T + aa * LD;
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int k = 0; k < NMAX; k += 2) { acc0 += ta[k] * q[k]; acc1 += ta[k + 1] * q[k + 1]; }
        T[aa * LD + rank] = acc0 + acc1;                      // P[aa][rank]
      }
    }
    wave_fence();
  } else {
    // ascending order + column-major eigenvectors (cusolver.h:76-95)
    int pos = 0;
    const double lam = rank < n ? D[rank] : 0.0;
    for (int j = 0; j < n; ++j) {
      const double lj = D[j];
      pos += (lj < lam) || (lj == lam && j < rank);
    }
    int* POS = reinterpret_cast<int*>(vv);
    if (rank < n) {
      POS[rank] = pos;
      a.Wout[(long long)bi * n + pos] = lam;
    }
    wave_fence();
    if (rank < n) {
      double* Vout = a.out + (long long)bi * n * n;
#pragma unroll
      for (int k = 0; k < NMAX; ++k)
        if (k < n) Vout[(long long)POS[k] * n + rank] = q[k];
    }
  }
  CUADMM_STAMP(4);
  }  // valid
#undef CUADMM_STAMP
}


}  // namespace cuadmm
