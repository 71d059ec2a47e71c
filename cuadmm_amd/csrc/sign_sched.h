// Per-block adaptive schedule of the matrix-sign (Newton-Schulz) PSD projection.
//
// Round 1 ran a fixed worst-case schedule for every block: 36 "lift" steps S <- p(mu S), mu = 1.53, followed by 8 plain
// steps (mu = 1), p(x) = 1.5 x - 0.5 x^3, which resolves every eigenvalue with |lambda| >= 1e-13 ||X||_1.  Typical ADMM
// iterates need a quarter of that.  This header is the state machine every sign kernel (one wavefront per block, one
// workgroup per block, batched-GEMM path) runs PER BLOCK, on quantities it already has in registers:
//
//     Y = S^2 :  a = tr Y,  b = ||Y||_F^2          =>  d2 = ||I - Y||_F^2 = n - 2a + b
//     S Y     :  g2 = ||S - S Y||_F^2 = sum_i s_i^2 (1 - s_i^2)^2
//
// (s_i: eigenvalues of the current iterate; every iterate is a polynomial in X, so these sums are exact functions of the
// spectrum.)  With "basin" eigenvalues (close to 1: u_i = 1 - s_i^2 small) and "unresolved" ones (v_i = s_i^2 small):
//
//     d2 - k            =  sum u^2 - 2 sum v + sum v^2           (k = number of unresolved = rint(d2))
//     g2 - (d2 - k)     = -sum u^3 + 3 sum v - ...               => sum v ~ (g2 - (d2 - k)) / 3, polluted at 3rd order only
//
// Phases.  LIFT (mu = 1.53): unresolved eigenvalues grow by 2.295 per step, basin eigenvalues bounce inside [0.5, 1].
// PROBE: two optimally scaled steps for the interval [0.5, 1] (mu = 1.309, 1.0845: basin error 0.5 -> 0.158 -> 0.011),
// then plain steps; after j probe steps the basin error is E_j, so d2 / g2 reveal what is left:
//     d2 < 0.5                      -> nothing unresolved (rigorous: every s_i^2 > 0.29): plain steps until g <= 2.3e-7,
//                                      one more update, done (error after the update <= 0.375 g^2 = 2e-14);
//     unresolved <= tol * G         -> what is left is below the resolution tol = 1e-13 relative to ||X||_1 (G = product of
//                                      the slopes 1.5 mu so far = growth of a tiny eigenvalue): finish the basin and stop;
//     unresolved visible (s_hat)    -> burst of lift steps that carries s_hat to the basin (no probing in between);
//     below the probe's noise floor -> one more plain step (every probe step squares the floor).
// The first step uses mu = 1.53 / t with t = min(1, ||Y||_F^(1/2)) >= spectral radius: a free tighter normalisation than
// ||X||_1.  Any mu in [1, 1.53] keeps [0.5, 1] invariant, so the heuristics only cost steps, never accuracy: the exit
// tests are the rigorous part, and kCap bounds the work (the result is then what the fixed schedule would have given).
//
// MEGA-LIFT (round 5).  A lift step multiplies an unresolved eigenvalue by 2.295, whatever its size: a block whose spectrum has a GAP
// -- eigenvalues of order one and a cluster at 1e-6 ... 1e-11 relative, every moment matrix of a relaxation whose iterate is numerically
// low-rank -- spends 15 ... 30 steps carrying the cluster up.  Once g = ||S - S Y||_F is small the spectrum is SPLIT, rigorously: every
// eigenvalue is <= gb = g (1 + 1.5 g^2) ("tiny") or within eb of 1 ("basin").  The cubic  q_c(x) = (1 + c) x - c x^3  -- the same two
// products, other coefficients -- has q_c(1) = 1, q_c'(1) = 1 - 2 c and q_c(x) = (1 + c) x (1 - O(x^2)) near 0: it multiplies the whole
// tiny cluster by 1 + c in ONE step and pays with the basin's error, eb -> |1 - 2 c| eb.  c is chosen so that the cluster's top lands at
// <= kMegaTiny and the basin stays within kMegaBasin of 1; the two probe steps that follow bring both into [0.5, 1] (q(1.309 x) maps
// [0.95, 1.05] into [0.76, 0.91] and 0.3 to 0.56).  Rounding: c (S - S Y) amplifies the products' roundoff to c eps, which is eps / G in
// the units of the ORIGINAL matrix (G = growth so far >= 1): the perturbation every ordinary step adds, three decades below the
// resolution.  The basin bound: eb = gb / 2 from g alone; after a plain step on a split spectrum 0.375 gb_prev^2 (what the lagged
// variant always used) -- so when the basin is what limits c, one more plain step (needed by the terminal phase anyway) squares it.
// Everything here is a choice of coefficients for steps the kernels run anyway; the exit tests are untouched.
//
// CLEAN MEGA-LIFT (round 6, the batched-GEMM path's groups of padded size <= 512).  The cap above exists because c (S - S Y) amplifies the
// roundoff of the product S Y, and the part of it that couples the basin to the lifted cluster rotates the resolved eigenvectors.  The cubic
// in Y that projects that part out,   S <- S + c (I - Y) R (I - Y),   R = S - S Y   (= S (1 + c (1 - S^2)^3) in exact arithmetic: the tiny
// cluster times 1 + c, a basin eigenvalue 1 - u/2 moved by c u^3 only),   damps the basin's share of R's noise by u on either side and needs no
// cap: one lift takes the cluster to kMegaTiny whatever the gap.  It costs a second step SLOT: slot 1 is an ordinary step whose output is R
// (coefficients -1, 1) instead of the next iterate; slot 2 forms M = R - R Y as a FULL product (R (I - Y) is symmetric only up to the very noise
// it removes: the mirrored upper-triangle product would put it back) and then S + c (M - Y M), mirrored (T R T is symmetric for symmetric R).
// Matrix-level replica (tools/dbg/clean_mega/msim.py, n = 96 ... 120, gaps 1e-6 ... 1e-13): errors 2e-16 ||X|| like the schedule without any
// mega-lift (capped: 1e-14), 42 -> 28 steps at a gap of 1e-9 ... 1e-11 (capped: 37 - 39), 44 -> 27 - 29 at 1e-12 (capped: 44).
//
// LAGGED variant (batched-GEMM path, psd_large.hip): there g2 of iterate k is only complete after the launch that
// applies mu_k, so decisions at step k use (a_k, b_k) and g2 of iterate k-1 (plain phase: g_k <= 0.75 g_{k-1}^2).
//
// DEFERRED variant (one wavefront per block, psd_sign_wave.h): decide<true> for step k is evaluated while the iterate of
// step k - 1 is on its way through LDS, i.e. with (a, b) AND g of iterate k - 1 (step 0 is decided in place: its b sets the
// normalisation).  (a, b) only enter through step 0, through the test d2 < 0.5 -- which, once true, stays true: [0.5, 1]
// is invariant -- and through the heuristics; every exit test works from g of the previous iterate, as in the lagged
// variant.  So the deferral costs at most a step, never accuracy, and takes the reductions and this state machine off the
// critical path of a step (they were a third of it).
//
// Compiles for host and device: tests/test_sign_schedule.py drives it on the CPU through cuadmm_sign_sched_simulate.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define CUADMM_SCHED_HD __host__ __device__ __forceinline__
#else
#define CUADMM_SCHED_HD inline
#endif

namespace cuadmm {

// MEGA = false: the state machine of rounds 2-4, no mega-lift, none of its state (the closed-block kernels of the block-diagonal
// synthetic configurations: their spectra have no gap to jump, and at that kernel's register budget the extra scalars were spills --
// 23 instead of 13 SGPRs in VGPR lanes, scratch traffic in the iteration: C2 lost 2 % before this switch existed).
template <bool MEGA>
struct SignSchedT {
  static constexpr double kMu = 1.53;          // p(1.53) = 0.5
  static constexpr double kSlope = 1.5 * 1.53; // growth of a tiny eigenvalue per lift step
  static constexpr double kMuP1 = 1.309;       // sqrt(3 / (1 + l + l^2)), l = 0.5
  static constexpr double kMuP2 = 1.0845;      // same for l = p(1.309 * 0.5) = 0.8417
  static constexpr double kTol = 1e-13;        // resolution relative to ||X||_1 (the fixed schedule's)
  static constexpr double kGExit = 2.3e-7;     // g at which one more plain update leaves an error <= 2e-14
  static constexpr int kLift0 = 3;
  static constexpr int kCap = 64;
  static constexpr double kMegaTiny = 0.3;     // where a mega-lift puts the top of the tiny cluster
  static constexpr double kMegaBasin = 0.05;   // how far from 1 it may push the basin
  static constexpr double kMegaMin = 5.3;      // worth it from two lift steps' growth on (2.295^2)
  static constexpr double kMegaCMax = 600.0;   // c of one mega-lift, and
  static constexpr double kMegaCSum = 600.0;   // ... summed over the mega-lifts of one projection.  c (S - S Y) amplifies the products' roundoff to
                                               // c eps, and the part of that noise that couples the basin to the lifted cluster ROTATES the resolved
                                               // eigenvectors by ~c eps: an error of ~0.16 c eps ||X|| in the projection (matrix-level replica on
                                               // low-rank + noise spectra: 3.5e-14 at sum c = 2 014; uncapped, c = 2e9 gave 4e-7 on a PSD input).
                                               // 600 keeps it at 1e-14 ||X||, a tenth of the resolution contract; a mega-lift cleaned of the
                                               // coupling, (I - Y)(S - S Y)(I - Y), would lift the cap for two more products (not built)


  double G = 1.0, gprev = -1.0;
  double cm = 0.0;                             // > 0: the step just decided is a mega-lift, S <- (1 + cm) S - cm S Y
  double gbl = -1.0;                           // gb of the PREVIOUS iterate when the step taken from it was plain (else -1)
  int megas = 0, waits = 0;
  double tbk = -1.0, ebk = -1.0;               // after a mega-lift: propagated bounds of the NEXT iterate (tiny <= tbk, basin within ebk of 1)
  double csum = 0.0;                           // sum of the mega-lifts' c so far
  bool chain = false;                          // the next step is decided from (tbk, ebk) alone: a further mega-lift or the first probe
  bool mega_on = true;                         // host model only: +8 on the mode switches it off (the schedule of rounds 2-4, for comparison)
  bool clean = false;                          // the CLEAN mega-lift: S <- S + c (I - Y)(S - S Y)(I - Y), two step slots, no cap on c (see take_mega)
  bool cont = false;                           // clean: the next step slot is the second half of the mega-lift just decided
  int half = 0;                                // clean: 1 = the slot just decided forms R = S - S Y, 2 = it applies S + cmc (I - Y) R (I - Y)
  double cmc = 0.0;
  int cap = kCap;                              // clean: the caller's step limit when it is below kCap (a mega-lift's two slots never straddle it)
  int k = 0, j = 0, fin = 0, steps = 0;
  // Warm start of the SCHEDULE (not of the iterate): consecutive ADMM iterations project nearly the same spectrum, so the
  // number of lift steps a block needed last time can serve as the length of its first lift phase now.  lift0 = hint from the
  // previous projection (kLift0 without one), lifts = lift-type steps taken this time = the next hint.  A wrong hint costs
  // steps only.  Measured gain is small (C2 12.0 -> 11.4 steps; C3 17 -> 18): the engine leaves it off (CUADMM_PSD_HINT=1).
  int lift0 = kLift0, lifts = 0;
  bool plain = false, plain_prev = false;

  // basin error after j probe steps starting from [0.5, 1]
  static CUADMM_SCHED_HD double probe_err(int j) {
    return j <= 2 ? 0.011 : (j == 3 ? 1.8e-4 : (j == 4 ? 5e-8 : 4e-15));
  }

  // lifts that carry an eigenvalue of size sh to (just below) the basin, bounded by the resolution limit
  CUADMM_SCHED_HD int burst_len(double sh) const {
    int c = 0;
    double x = sh > 1e-300 ? sh : 1e-300, g = G;
    while (x * kSlope <= 0.45 && g * kTol < 0.5 && c < kCap) { x *= kSlope; g *= kSlope; ++c; }
    return c > 1 ? c : 1;
  }

  // Does the next decision read its statistics at all?  Inside a lift phase or a burst (k > 0) and on the two probe steps the
  // scale is fixed in advance: the one-wavefront kernels then skip the three wave reductions and call decide with zeros
  // (same decisions, same step counts -- a third of a step's latency on two steps out of three).
  CUADMM_SCHED_HD bool needs_stats() const {
    if (MEGA && chain) return false;
    if (steps == 0 || fin > 0 || plain) return true;
    if (k > 0 && G * kTol < 0.5) return false;
    return !(j == 0 || j == 1);
  }

  // One decision per Newton-Schulz step.  n: true block size; a, b from Y = S^2 of the CURRENT iterate; g2: ||S - SY||_F^2
  // of the current iterate (LAG = false) or of the previous one (LAG = true; ignored at the first step).
  // Returns mu of this step; `last` = this update is the final one.
  // Which statistics does the next decision read?  needs_stats(): any at all; needs_ab(): tr Y and ||Y||_F^2 as well (the first
  // step and the decision after a probe); in the plain and finishing phases only g2 is read -- one wave reduction, not three.
  CUADMM_SCHED_HD bool needs_ab() const { return steps == 0 || (fin == 0 && !plain && !(MEGA && chain)); }

  // takes a mega-lift of (at most) c_want from an iterate whose tiny part is <= tb and whose basin is within eb of 1: sets cm, the
  // propagated bounds and whether the next step continues the chain; false when what the caps leave is not worth a step
  // c_want: what the split allows the CAPPED lift (tiny cluster to kMegaTiny, basin kept within kMegaBasin of 1); c_clean: what would take
  // the cluster to kMegaTiny outright -- the clean lift's (it barely moves the basin), used when `clean` is set and it pays.
  CUADMM_SCHED_HD bool take_mega(int n, double c_want, double c_clean, double tb, double eb) {
    double c = c_want;
    if (clean) {
      // The clean lift costs two step slots, the second one a product and a half: ~2.5 steps for the whole factor 1 + c_clean.  The capped lift
      // costs one step for what the caps leave (c <= 600, 600 per projection) and 2.295-fold steps for the rest: clean pays when the factor left
      // over exceeds 2.295^1.5 ~ 3.5.  Its roundoff: x + c x (1 - x^2)^3 moves the basin by <= 9 c eb^3, and what c amplifies is
      // eps ||S - S Y|| ~ eps tb (the two outer products), i.e. c eps tb <= 0.3 eps: no cap.
      double cc = c_want;
      cc = cc < kMegaCMax ? cc : kMegaCMax;
      cc = cc < kMegaCSum - csum ? cc : kMegaCSum - csum;
      const double capped = 1.0 + cc >= kMegaMin ? 1.0 + cc : 1.0;
      if (1.0 + c_clean > 3.5 * capped && 1.0 + c_clean >= kMegaMin * kSlope && steps + 3 <= cap) {
        cmc = c_clean;
        half = 1;
        ++megas;
        tbk = (1.0 + c_clean) * tb;
        ebk = eb + 9.0 * c_clean * eb * eb * eb;
        cont = true;
        chain = true;
        k = 0; j = 0; waits = 0;
        return true;
      }
    }
    c = c < kMegaCMax ? c : kMegaCMax;
    c = c < kMegaCSum - csum ? c : kMegaCSum - csum;
    if (!(1.0 + c >= kMegaMin)) return false;
    cm = c;
    csum += c;
    ++megas;
    tbk = (1.0 + c) * tb;
    ebk = 2.14 * c * eb + c * 2.3e-16 * sqrt((double)n);
    chain = true;                                        // the next decision reads (tbk, ebk): a further mega-lift or the first probe
    k = 0; j = 0; waits = 0;
    return true;
  }

  // coefficients of the step decide() returned mu for:  S <- alpha S Y + beta S
  CUADMM_SCHED_HD void coefs(double mu, double& alpha, double& beta) const {
    if (MEGA && cm > 0.0) { alpha = -cm; beta = 1.0 + cm; }
    else if (MEGA && half == 1) { alpha = -1.0; beta = 1.0; }                  // first slot of a clean mega-lift: the step's output is R = S - S Y
    else { alpha = -0.5 * mu * mu * mu; beta = 1.5 * mu; }
  }

  template <bool LAG>
  CUADMM_SCHED_HD double decide(int n, double a, double b, double g2, bool& last) {
    // exit tests compare g^2 with kGExit^2 (no square root on the steps that only test); g itself -- this iterate's, or the
    // previous one's in the lagged variant -- is formed where a bound is derived from it
    constexpr double kGExit2 = kGExit * kGExit;
    const double g2c = g2 > 0.0 ? g2 : 0.0;
    double mu = 1.0;
    last = false;
    if (MEGA) cm = 0.0;
    if (MEGA && cont) {                 // second half of a clean mega-lift: nothing is decided, nothing is read
      cont = false;
      half = 2;
      plain_prev = false;
      gbl = -1.0;
      ++steps;
      if (steps >= kCap) last = true;
      return 1.0;
    }
    if (MEGA) half = 0;
    const bool was_plain = plain;
    double g_now = -1.0;                // !LAG: sqrt(g2) once a branch below needed it (kept in gprev for the host model)
    double gb_now = -1.0;               // the split bound of THIS iterate, where a branch formed it
    if (steps == 0) {
      if (!(b > 0.0)) { last = true; }  // zero (or non-finite) block: one harmless update
      else {
        double t = sqrt(sqrt(b));
        t = t < 1.0 ? t : 1.0;
        mu = kMu / t;
        k = (lift0 > 1 ? (lift0 < kCap ? lift0 : kCap) : 1) - 1;
        j = 0;
      }
    } else if (MEGA && chain) {
      // right after a mega-lift: no statistics are read.  What the caps left of the allowed factor is taken now; then the probes.
      chain = false;
      const double f1 = kMegaTiny / tbk, f2 = 1.0 + kMegaBasin / (2.14 * ebk + 1e-300);
      const double f = f1 < f2 ? f1 : f2;
      if (!(G * kTol < 0.5 && tbk > kTol * G && take_mega(n, f - 1.0, f1 - 1.0, tbk, ebk))) { k = 0; mu = kMuP1; j = 1; }
    } else if (fin > 0) {
      --fin;
      last = fin == 0;
      if (!LAG && g2c <= kGExit2) last = true;
    } else if (plain) {
      if (LAG) last = plain_prev && gprev >= 0.0 && 0.75 * gprev * gprev <= kGExit;
      else last = g2c <= kGExit2;
    } else if (k > 0 && G * kTol < 0.5) {
      mu = kMu; --k; j = 0;
    } else if (j == 0) {
      k = 0;
      mu = kMuP1; j = 1;
    } else if (j == 1) {
      mu = kMuP2; j = 2;
    } else if ((double)n - 2.0 * a + b < 0.5) {
      plain = true;
      if (!LAG) last = g2c <= kGExit2;
    } else {
      const double d2 = (double)n - 2.0 * a + b;
      const double g = LAG ? gprev : sqrt(g2c);
      g_now = g;
      // Unresolved eigenvalues exist.  Rigorous facts (no assumption on the spectrum): every eigenvalue satisfies
      // s (1 - s^2) <= g, i.e. it is either <= gb = g (1 + 1.5 g^2) or within gb / 2 of 1 (for g <= 0.3).
      double gb = -1.0, eb = 0.0;          // bounds for the CURRENT iterate: unresolved <= gb, basin error <= eb
      if (!LAG) {
        if (g <= 0.3) { gb = g * (1.0 + 1.5 * g * g); eb = 0.5 * gb; }
      } else if (j >= 3 && gprev >= 0.0 && gprev <= 0.3) {   // previous step was plain (mu = 1)
        const double gq = gprev * (1.0 + 1.5 * gprev * gprev);
        gb = 1.5 * gq;
        eb = 0.375 * gq * gq;
      }
      gb_now = gb;
      const bool at_limit = G * kTol >= 0.5;             // whatever is still unresolved is below the resolution
      // mega-lift: the factor the split allows (tiny cluster to kMegaTiny, basin kept within kMegaBasin of 1)
      double f_tiny = 0.0, f_all = 0.0, ebm_used = eb;
      if (MEGA && mega_on && gb > 0.0 && !at_limit && gb > kTol * G) {
        double ebm = eb;
        if (!LAG && gbl >= 0.0) { const double e2 = 0.4 * gbl * gbl; ebm = e2 < ebm ? e2 : ebm; }   // 1.5 e^2 + 0.5 e^3, e <= gbl / 2 <= 0.16, either side of 1
        f_tiny = kMegaTiny / gb;
        const double f_basin = 1.0 + kMegaBasin / (2.14 * ebm + 1e-300);
        f_all = f_tiny < f_basin ? f_tiny : f_basin;
        ebm_used = ebm;
      }
      if (at_limit || (gb >= 0.0 && gb <= kTol * G)) {
        // finish the basin and stop: plain steps until the error bound is <= 1.15e-7, then the last update
        int c;
        if (gb >= 0.0) {
          c = 1;
          double e = eb;
          while (e > 1.15e-7 && c < 8) { e = 1.5 * e * e; ++c; }
        } else {
          c = (5 - j > 0 ? 5 - j : 0) + 2;               // g carries no bound: basin error table + stragglers
        }
        fin = c - 1;
        last = fin == 0;
      } else {
        // Heuristics from here on (they only cost steps): what is left, and how far below the basin is it?
        const double kk = rint(d2), frac = d2 - kk;
        const double e = probe_err(j), u = 2.0 * e;
        double v, noise;
        if (LAG) { v = -frac * 0.5; noise = (double)n * (0.5 * u * u + 1e-15); }
        else { v = (g2 - frac) * (1.0 / 3.0); noise = (double)n * (u * u * u * (1.0 / 3.0) + 1e-15); }
        v = v > 0.0 ? v : 0.0;
        const bool exact = j >= 5 && gb >= 0.0;
        if (frac > 0.05 || frac < -0.05 || j >= 6) {     // mid-range eigenvalues (s > 0.16) present: keep lifting
          k = 2; mu = kMu; --k; j = 0;
        } else if (exact || v > 2.0 * noise) {
          // something unresolved is VISIBLE (g alone cannot tell a tiny cluster from the basin's own error: exactly rank-deficient
          // blocks must keep taking plain steps until gb <= tol G): the whole cluster in one step where the split allows it
          bool taken = false;
          if (MEGA) {
            const double s_hat = sqrt(v + 2.0 * noise);  // what the cluster weighs by the statistics of THIS iterate
            const bool stale = gb > 4.0 * s_hat && waits < 3;   // the bound still carries the basin's error (the lagged g: a step old)
            if ((f_all >= kMegaMin || (clean && f_tiny >= kMegaMin)) && !stale && take_mega(n, f_all - 1.0, f_tiny - 1.0, gb, ebm_used)) {
              taken = true;                              // (the probes that follow the chain restore [0.5, 1])
            } else if (f_tiny >= kMegaMin || (mega_on && stale && gb >= 0.0 && kMegaTiny / s_hat >= kMegaMin)) {
              ++j; ++waits;                              // a plain step squares the basin's share of the bound (and is not wasted)
              taken = true;
            }
          }
          if (taken) {
          } else {
            k = burst_len(exact ? gb : sqrt(v + 2.0 * noise));
            mu = kMu; --k; j = 0;
          }
        } else {
          ++j;                                           // below the probe's noise floor: one more plain step
        }
      }
    }
    if (mu >= 0.999 * kMu) ++lifts;
    plain_prev = was_plain;
    if (!LAG) gprev = g_now;              // only where it was formed; the one-wavefront kernels never read it
    if (MEGA) gbl = (mu == 1.0 && cm == 0.0) ? gb_now : -1.0;
    G *= (MEGA && cm > 0.0) ? 1.0 + cm : ((MEGA && half == 1) ? 1.0 + cmc : 1.5 * mu);
    ++steps;
    if (steps >= kCap) last = true;
    return mu;
  }
};

using SignSched = SignSchedT<true>;
using SignSchedPlain = SignSchedT<false>;

// Scalar model of the iteration on a spectrum (the iteration acts on eigenvalues independently): used by the CPU tests
// and by tools/sign_schedule_sim.py.  s[i] = |lambda_i| / ||X||_1 on entry, the sign estimates on exit.
// lag: 0 = decisions from the current iterate (statistics skipped where needs_stats() says so, as the kernels do), 1 = lagged
// (g of the previous iterate), 2 = deferred ((a, b, g) of the previous one: measured and rejected), 3 = as 0 but with the
// statistics passed on every step (the reference for the claim that skipping them changes nothing)
inline int sign_sched_simulate(double* s, int n, int lag, double* err_out, int lift0 = 0, int* lifts_out = nullptr) {
  SignSched st;
  if (lag & 16) { st.clean = true; lag &= 15; }
  if (lag & 8) { st.mega_on = false; lag &= 7; }
  if (lift0 > 0) st.lift0 = lift0;
  double orig_max_err = 0.0;
  double* s0 = new double[n > 0 ? n : 1];
  for (int i = 0; i < n; ++i) s0[i] = s[i];
  bool last = false;
  double g2_prev = 0.0, a_prev = 0.0, b_prev = 0.0;
  while (!last) {
    double a = 0, b = 0, g2 = 0;
    for (int i = 0; i < n; ++i) {
      const double y = s[i] * s[i], r = s[i] * (1.0 - y);
      a += y; b += y * y; g2 += r * r;
    }
    double mu;
    if (lag == 3) {
      mu = st.decide<false>(n, a, b, g2, last);
    } else if (lag) {
      st.gprev = st.steps == 0 ? -1.0 : sqrt(g2_prev);
      if (lag == 2 && st.steps > 0) mu = st.decide<true>(n, a_prev, b_prev, 0.0, last);
      else mu = st.decide<true>(n, a, b, 0.0, last);
    } else {
      // as the one-wavefront kernels call it: without statistics where needs_stats() says they are not read
      mu = st.needs_stats() ? st.decide<false>(n, a, b, g2, last) : st.decide<false>(n, 0.0, 0.0, 0.0, last);
    }
    g2_prev = g2; a_prev = a; b_prev = b;
    double al, be;
    st.coefs(mu, al, be);
    if (st.half == 1) continue;                     // R is formed; the iterate moves in the second slot
    if (st.half == 2) {
      for (int i = 0; i < n; ++i) { const double w = 1.0 - s[i] * s[i]; s[i] += st.cmc * s[i] * w * w * w; }
      continue;
    }
    for (int i = 0; i < n; ++i) s[i] = be * s[i] + al * s[i] * s[i] * s[i];
  }
  for (int i = 0; i < n; ++i) {
    const double e = s0[i] * fabs(1.0 - s[i]) * 0.5;
    orig_max_err = e > orig_max_err ? e : orig_max_err;
  }
  delete[] s0;
  if (err_out) *err_out = orig_max_err;
  if (lifts_out) *lifts_out = st.lifts;
  return st.steps;
}

}  // namespace cuadmm
