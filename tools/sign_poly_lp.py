#!/usr/bin/env python3
"""Round 5: would higher-degree steps cut the matrix products of the sign iteration?  (CPU only, seconds.)

    python tools/sign_poly_lp.py > profiles/r05_sign_poly_lp.txt

1. LIFT PHASE.  A lift step is an odd polynomial q of degree d = 2k+1 applied to the iterate; with Y = S^2 and Paterson-Stockmeyer
   evaluation it costs P(d) symmetric products (3: 2, 5: 3, 7: 4, 9: 4, 11: 5, 13: 5, 15: 6).  What a step buys is its slope at the
   origin (the growth of an unresolved eigenvalue) under the constraints that make the per-block adaptive schedule possible: q maps
   the basin [l0, 1] into itself and [0, 1] into [0, 1] (sign_sched.h: resolved eigenvalues stay resolved, the exit tests stay
   rigorous without knowing the smallest eigenvalue).  The largest such slope is a linear program in the coefficients; the figure
   of merit is slope ** (1 / products), the growth per matrix product.
   ("Polar Express"-type quintics quote a slope of ~8.3: their image of [0, 1] reaches ~2, so the NORMALISED slope is ~4.2 for three
   products -- 1.61 per product, the optimally scaled cubic's sqrt(3 sqrt 3 / 2) = 1.61.)
2. KNOWN SPECTRUM.  Steps of the optimally scaled cubic when the smallest eigenvalue l is KNOWN (alpha = sqrt(3 / (1 + l + l^2)),
   Chen & Chow), against the schedule of sign_sched.h with a perfect hint (lifts with mu = 1.53 to the basin, two probes, three
   plain steps): what clairvoyance is worth.
3. TERMINAL PHASE from the basin [0.5, 1] to 2e-14: products of cubic / quintic / heptic Pade-type chains.
"""
import numpy as np
from scipy.optimize import linprog

PRODUCTS = {3: 2, 5: 3, 7: 4, 9: 4, 11: 5, 13: 5, 15: 6}


def best_slope(deg, l0, ngrid=4000):
    k = (deg + 1) // 2
    xs = np.linspace(0, 1, ngrid + 1)[1:]
    P = np.stack([xs ** (2 * i + 1) for i in range(k)], 1)
    A = [P]
    b = [np.ones(len(xs))]                                   # q <= 1 on (0, 1]
    m = xs >= l0
    A.append(-P[m]); b.append(-l0 * np.ones(m.sum()))        # q >= l0 on [l0, 1]
    m2 = xs < l0
    A.append(-P[m2]); b.append(-xs[m2])                      # q(x) >= x below the basin
    c = np.zeros(k); c[0] = -1
    r = linprog(c, A_ub=np.vstack(A), b_ub=np.concatenate(b), bounds=[(None, None)] * k)
    return r.x if r.status == 0 else None


def steps_known(l, tol=4e-15):
    k = 0
    while 1 - l > tol and k < 100:
        a = np.sqrt(3 / (1 + l + l * l))
        p = lambda x: 1.5 * a * x - 0.5 * a ** 3 * x ** 3
        l = min(p(l), p(1.0)); k += 1
    return k


def steps_sched(l):
    k, x = 0, l
    while x < 0.5:
        x = 1.5 * 1.53 * x - 0.5 * 1.53 ** 3 * x ** 3; k += 1
    return k + 5


def chain(e, order_seq):
    """error after a chain of Pade-type steps: order 2 (cubic, 1.5 e^2), 3 (quintic, 2.5 e^3), 4 (heptic, 4.375 e^4)"""
    c = {2: 1.5, 3: 2.5, 4: 4.375}
    for o in order_seq:
        e = c[o] * e ** o
    return e


def main():
    print("1. lift phase: largest slope at the origin of an odd polynomial that keeps [l0, 1] and [0, 1] invariant")
    print("   degree products  l0    slope   growth per product   (production: cubic, l0 = 0.5, slope 2.295, 1.515)")
    for l0 in (0.5, 0.3, 0.1):
        for deg, prods in PRODUCTS.items():
            x = best_slope(deg, l0)
            if x is None:
                continue
            print("   %6d %8d  %.2f  %6.3f   %.4f" % (deg, prods, l0, x[0], x[0] ** (1.0 / prods)))
    print()
    print("2. steps (2 products each) to 4e-15 from a smallest eigenvalue l: optimally scaled cubic with l KNOWN / sign_sched.h with a perfect hint")
    for l in (0.3, 0.1, 0.05, 0.02, 0.01, 3e-3, 1e-3, 1e-4, 1e-6):
        print("   l = %-7g  %2d / %2d" % (l, steps_known(l), steps_sched(l)))
    print()
    print("3. terminal phase from the basin (error 0.5 after the lifts; two optimally scaled cubic probes leave 0.011): products to <= 2e-14")
    for name, e0, seq, prods in (("production: probes + 3 cubic", 0.011, (2, 2, 2), 4 + 6),
                                 ("probes + quintic + quintic", 0.011, (3, 3), 4 + 6),
                                 ("probes + heptic + cubic", 0.011, (4, 2), 4 + 4 + 2),
                                 ("probes + cubic + heptic", 0.011, (2, 4), 4 + 2 + 4)):
        print("   %-32s error %.1e, %d products" % (name, chain(e0, seq), prods))


if __name__ == "__main__":
    main()
