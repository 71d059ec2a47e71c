// REJECTED (round 4), kept for the record -- not compiled.  The one-launch matrix-sign kernel of psd_large.hip by ROW PANELS: one
// barrier per step instead of two.  Correct (tests/test_gpu_psd.py had a test for it: 1e-14 against the oracle, bit-identical across
// barrier modes and repetitions) and SLOWER: PlanarHand_N=1's projection 0.52 -> 0.71 ms (c1 1 030 -> 730 iterations/s).  The tile
// variant spends ~5.9 us per phase (two per step) at ~40 steps per projection; a panel step is one barrier (~6 us) + both products on
// one CU (3.5 us at N = 128: MFMA-bound with one wavefront per SIMD) + a trip for the statistics, the decision, a trip for U and the
// combination (~4 us) = 14 - 16 us.  Two lessons: (1) the iterate must stay EXACTLY symmetric -- a lift step multiplies antisymmetric
// rounding noise by 2.3 like everything small, and after ~40 steps of a rank-deficient block the projection was off by 0.1 until the
// combination mirrored the upper triangle of U; (2) a loop of "global load, LDS store" runs one L2 trip per iteration unless all the
// loads are issued first (64 trips: 1.7 ms per projection before the batch of 36 registers).
// Host side it needed: ClusterArgs {pstat, pstat_half, panel}, LgPath {panel, wtiles}, a [parity][member][16][4] statistics buffer,
// dynamic LDS of 8 (N^2 + 16 N) bytes, and "every one-launch group of the plan has N <= 128" as the condition.
// ---- the same, by ROW PANELS: one barrier per step (round 4) -----------------------------------------------------------------
// The tile variant above meets twice per step (Y = S S, then T from S Y) and a meeting is ~5 dependent trips to the coherence point: a
// PlanarHand projection is ~22 steps x 2 x ~10 us.  S and every polynomial in it commute, so a workgroup that owns the 16 rows R of a
// member can form (S^2)_R = S_R S and (S^3)_R = (S^2)_R S from its own rows and the OLD iterate alone; mu_k needs tr Y and ||Y||_F^2 of
// the whole matrix, but it only scales: S' = 1.5 mu S - 0.5 mu^3 S^3 is an elementwise combination once mu is known.  So per step:
// both products on the panel (the whole iterate sits in LDS: N <= 128, 128 KB, columns swizzled by the parity of the row so that the four
// k-rows of a fragment read fall on disjoint banks without padding; the panel of S^2 goes through a 16 KB staging area to become the A
// operand), the panel of U = S^3 and three partial sums to global memory, ONE barrier, then every workgroup sums the partial sums in
// tile order, runs the schedule's decision for itself (same inputs, same arithmetic: same mu everywhere) and combines the whole next
// iterate into its LDS from its own copy of S and the panels of U.  The schedule is the tile variant's (lagged: a_k, b_k with g of the
// iterate before), the sums are associated differently: results agree to rounding, not bit for bit (test).  U alternates between the
// buffers T and S (the global S is only read at the start), the projection P = 0.5 (X0 + X0 S) goes to Y.  The combination mirrors the
// upper triangle of U, so every workgroup's iterate is exactly symmetric and the A operand can be read through the transpose.
__global__ __launch_bounds__(256) void lg_sign_panel_kernel(ClusterMulti cm) {
  extern __shared__ double pn_smem[];       // S[N][N] (column c of row k at c ^ 16 (k & 1)) | Yst[N][16] (k-major panel: the A operand)
  __shared__ double red[16];
  __shared__ int s_local;
  __shared__ double s_sched_raw[(sizeof(SignSched) + 7) / 8];     // the schedule's state, thread 0's (no initialisers on __shared__ objects)
  __shared__ double s_gprev2;
  SignSched& s_sched = *reinterpret_cast<SignSched*>(s_sched_raw);
  int gi = 0;
  while (gi + 1 < cm.n && (int)blockIdx.x >= cm.wg_begin[gi + 1]) ++gi;
  const ClusterArgs ca = cm.ca[gi];
  const SignArgs sg = cm.sg[gi];
  const int N = ca.N, NT = N / 16;
  const int w = (int)blockIdx.x - cm.wg_begin[gi], slot = w >> 3;
  const int member = ca.spread ? w / NT : (w & 7) + 8 * (slot / NT);
  const int tile = ca.spread ? w % NT : slot % NT;
  if (member >= ca.count) return;               // the whole workgroup, before any barrier
  const int n = sg.st[member].n;                // written by the prologue launch
  const int nt = (n + 15) >> 4, NN = nt * 16;   // row panels / columns that hold data (the rest of N is zero padding)
  if (tile >= nt) return;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kk = lane >> 4;
  const int row0 = tile * 16;
  double* Sl = pn_smem;
  double* Yst = pn_smem + (size_t)N * N;
  unsigned* bar = ca.bar + member;
  unsigned phase = 0;
  int xcc0;
  {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    xcc0 = x & 15;
    if (tid == 0) __hip_atomic_store(ca.xcc + (size_t)member * NT + tile, x & 15, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the iterate into LDS (the prologue launch wrote S = X0 / ||X0||_1), the schedule's fresh state
  const size_t mat = (size_t)member * (size_t)N * (size_t)N;
  {
    const double* Sg = ca.S + mat;
    for (int idx = tid; idx < NN * N; idx += 256) {
      const int r = idx / N, c = idx - r * N;
      Sl[r * N + (c ^ ((r & 1) << 4))] = Sg[idx];
    }
    if (tid == 0) { s_sched = sg.st[member].sched; s_gprev2 = 0.0; }
  }
  if (!lg_member_barrier(bar, (unsigned)nt * ++phase, false)) { if (tid == 0 && ca.fail) atomicAdd(ca.fail, 1); return; }
  if (tid == 0) {
    int same = ca.force_agent ? 0 : 1;
    for (int q = 0; q < nt; ++q)
      same &= __hip_atomic_load(ca.xcc + (size_t)member * NT + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xcc0;
    s_local = same;
  }
  __syncthreads();
  const bool local = s_local != 0;
  const int j0 = wave, j1 = wave + 4;           // this wavefront's column tiles (nt <= 8)
  const bool has0 = j0 < nt, has1 = j1 < nt;
  int step = 0, steps_done = 0x7fffffff, sched_steps = 0;
  for (; step < ca.max_steps; ++step) {
    // Y_R = S_R S: A[r][k] = S[row0 + r][k] read as S[k][row0 + r] (symmetric up to rounding)
    lg_v4f64 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < NN; k0 += 4) {
      const int k = k0 + kk, sw = (k & 1) << 4;
      const double* rowk = Sl + k * N + r16;
      const double af = rowk[(row0) ^ sw];
      if (has0) y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af, rowk[(16 * j0) ^ sw], y0, 0, 0, 0);
      if (has1) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af, rowk[(16 * j1) ^ sw], y1, 0, 0, 0);
    }
    double p_tr = 0.0, p_y2 = 0.0, p_g2 = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = kk + 4 * r;                // row inside the panel; column inside the tile: r16
      if (has0) { const double v = y0[r]; p_y2 += v * v; if (j0 == tile && lr == r16) p_tr += v; Yst[(16 * j0 + r16) * 16 + lr] = v; }
      if (has1) { const double v = y1[r]; p_y2 += v * v; if (j1 == tile && lr == r16) p_tr += v; Yst[(16 * j1 + r16) * 16 + lr] = v; }
    }
    __syncthreads();
    // U_R = Y_R S
    lg_v4f64 u0 = {0.0, 0.0, 0.0, 0.0}, u1 = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < NN; k0 += 4) {
      const int k = k0 + kk, sw = (k & 1) << 4;
      const double* rowk = Sl + k * N + r16;
      const double af = Yst[k * 16 + r16];
      if (has0) u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af, rowk[(16 * j0) ^ sw], u0, 0, 0, 0);
      if (has1) u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af, rowk[(16 * j1) ^ sw], u1, 0, 0, 0);
    }
    double* Ub = ((step & 1) ? ca.S : ca.T) + mat;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + kk + 4 * r, sw = (row & 1) << 4;
      if (has0) { const double d = Sl[row * N + ((16 * j0 + r16) ^ sw)] - u0[r]; p_g2 += d * d; Ub[(size_t)row * N + 16 * j0 + r16] = u0[r]; }
      if (has1) { const double d = Sl[row * N + ((16 * j1 + r16) ^ sw)] - u1[r]; p_g2 += d * d; Ub[(size_t)row * N + 16 * j1 + r16] = u1[r]; }
    }
    p_tr = wave_sum(p_tr); p_y2 = wave_sum(p_y2); p_g2 = wave_sum(p_g2);
    if (lane == 0) { red[wave] = p_tr; red[4 + wave] = p_y2; red[8 + wave] = p_g2; }
    __syncthreads();
    double* ps = ca.pstat + (size_t)(step & 1) * (size_t)ca.pstat_half + ((size_t)member * 16 + tile) * 4;
    if (tid == 0) {
      ps[0] = (red[0] + red[1]) + (red[2] + red[3]);
      ps[1] = (red[4] + red[5]) + (red[6] + red[7]);
      ps[2] = (red[8] + red[9]) + (red[10] + red[11]);
    }
    if (!lg_member_barrier(bar, (unsigned)nt * ++phase, local)) { if (tid == 0 && ca.fail) atomicAdd(ca.fail, 1); return; }
    if (tid == 0) {
      const double* pm = ca.pstat + (size_t)(step & 1) * (size_t)ca.pstat_half + (size_t)member * 16 * 4;
      double a = 0.0, b = 0.0, g2 = 0.0;
      for (int q = 0; q < nt; ++q) { a += pm[4 * q]; b += pm[4 * q + 1]; g2 += pm[4 * q + 2]; }
      SignSched sc = s_sched;
      sc.gprev = step == 0 ? -1.0 : sqrt(s_gprev2 > 0.0 ? s_gprev2 : 0.0);
      bool last;
      const double mu = sc.decide<true>(n, a, b, 0.0, last);
      s_sched = sc;
      s_gprev2 = g2;
      red[12] = mu;
      red[13] = last ? 1.0 : 0.0;
      red[14] = (double)sc.steps;
      if (local) {
        int x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        if ((x & 15) != xcc0 && ca.fail) atomicAdd(ca.fail, 1);
      }
    }
    __syncthreads();
    const double mu = red[12];
    const bool last = red[13] != 0.0;
    const double alpha = -0.5 * mu * mu * mu, beta = 1.5 * mu;
    // the next iterate, whole, from this workgroup's S and everybody's U -- the UPPER triangle decides and is mirrored (as the tile
    // variant does): the iterate must stay EXACTLY symmetric.  A lift step multiplies everything small by 2.3, antisymmetric rounding
    // noise included: after the ~40 steps of a rank-deficient block it would be O(1) (measured: projection errors of 0.1).
    // One element of every upper 16 x 16 tile per thread, ALL loads in flight before the first use (one trip to the L2, not 36).
    {
      const int ntri = nt * (nt + 1) / 2;                         // <= 36
      const int ra = tid >> 4, cb = tid & 15;
      double uv[36];
#pragma unroll
      for (int t = 0; t < 36; ++t) {
        if (t < ntri) {
          int tb, ta;
          tri_decode(t, tb, ta);                                  // tb >= ta
          uv[t] = Ub[(size_t)(16 * ta + ra) * N + 16 * tb + cb];
        }
      }
#pragma unroll
      for (int t = 0; t < 36; ++t) {
        if (t < ntri) {
          int tb, ta;
          tri_decode(t, tb, ta);
          const int r = 16 * ta + ra, c = 16 * tb + cb;
          if (c >= r) {
            const double v = beta * Sl[r * N + (c ^ ((r & 1) << 4))] + alpha * uv[t];
            Sl[r * N + (c ^ ((r & 1) << 4))] = v;
            if (c != r) Sl[c * N + (r ^ ((c & 1) << 4))] = v;
          }
        }
      }
    }
    __syncthreads();
    if (last) { steps_done = step + 1; sched_steps = (int)red[14]; ++step; break; }
  }
  // P_R = 0.5 (X0_R + X0_R S): the panel of X0 through the staging area (k-major)
  const double* X0 = ca.X0 + mat;
  for (int idx = tid; idx < 16 * NN; idx += 256) {
    const int r = idx / NN, k = idx - r * NN;
    Yst[k * 16 + r] = X0[(size_t)(row0 + r) * N + k];
  }
  __syncthreads();
  lg_v4f64 q0 = {0.0, 0.0, 0.0, 0.0}, q1 = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < NN; k0 += 4) {
    const int k = k0 + kk, sw = (k & 1) << 4;
    const double* rowk = Sl + k * N + r16;
    const double af = Yst[k * 16 + r16];
    if (has0) q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af, rowk[(16 * j0) ^ sw], q0, 0, 0, 0);
    if (has1) q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af, rowk[(16 * j1) ^ sw], q1, 0, 0, 0);
  }
  double* Pb = ca.Y + mat;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int lr = kk + 4 * r, row = row0 + lr;
    if (has0) Pb[(size_t)row * N + 16 * j0 + r16] = 0.5 * Yst[(16 * j0 + r16) * 16 + lr] + 0.5 * q0[r];
    if (has1) Pb[(size_t)row * N + 16 * j1 + r16] = 0.5 * Yst[(16 * j1 + r16) * 16 + lr] + 0.5 * q1[r];
  }
  if (tile == 0 && tid == 0) {
    if (steps_done != 0x7fffffff) {
      sg.done[member].steps = sched_steps;
      sg.done[member].done_at = steps_done;
      if (sg.hint) sg.hint[sg.ids[member]] = s_sched.lifts;
      atomicMax(sg.group + 1, sched_steps);
      atomicSub(sg.group, 1);
    }
  }
}

