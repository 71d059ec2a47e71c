import ctypes as C, numpy as np, sys
lib = C.CDLL("/tmp/cm/libshim.so")
lib.ss_new.restype = C.c_void_p; lib.ss_new.argtypes = [C.c_int, C.c_int]
lib.ss_free.argtypes = [C.c_void_p]
lib.ss_cont.argtypes = [C.c_void_p]; lib.ss_steps.argtypes = [C.c_void_p]; lib.ss_megas.argtypes = [C.c_void_p]
lib.ss_decide.restype = C.c_double
lib.ss_decide.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double] + [C.POINTER(C.c_double)] * 2 + [C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]

def sym(A):           # the mirrored product: the upper triangle decides
    U = np.triu(A)
    return U + np.triu(A, 1).T

def project(X, clean, mega_on):
    n = X.shape[0]
    nrm = np.abs(X).sum(axis=0).max()
    S = X / nrm
    h = lib.ss_new(clean, mega_on)
    gprev = -1.0
    last = C.c_int(0); half = C.c_int(0); al = C.c_double(); be = C.c_double(); cmc = C.c_double()
    R = None; Y = None
    while not last.value:
        if lib.ss_cont(h):
            M = R - R @ Y                      # FULL product
            lib.ss_decide(h, n, 0.0, 0.0, gprev, C.byref(al), C.byref(be), C.byref(half), C.byref(cmc), C.byref(last))
            assert half.value == 2
            S = sym(S + cmc.value * (M - Y @ M))
            continue
        Y = sym(S @ S)
        a = np.trace(Y); b = (Y * Y).sum()
        lib.ss_decide(h, n, a, b, gprev, C.byref(al), C.byref(be), C.byref(half), C.byref(cmc), C.byref(last))
        SY = sym(S @ Y)
        gprev = np.sqrt(((S - SY) ** 2).sum())
        if half.value == 1:
            R = S - SY
        else:
            S = al.value * SY + be.value * S
    steps = lib.ss_steps(h); megas = lib.ss_megas(h)
    lib.ss_free(h)
    P = 0.5 * (X + sym(X @ S))
    return P, steps, megas

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def make(n, lam):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    X = (Q * lam) @ Q.T
    X = 0.5 * (X + X.T)
    Pex = (Q * np.maximum(lam, 0)) @ Q.T
    return X, Pex

def spectrum(n, r, lo, hi, zeros=0):
    lam = np.zeros(n)
    lam[:r] = rng.uniform(0.2, 1.0, r) * rng.choice([-1, 1], r)
    m = n - r - zeros
    lam[r:r + m] = 10.0 ** rng.uniform(lo, hi, m) * rng.choice([-1, 1], m)
    return lam

cases = [("gap 1e-6..1e-8, r=6", 120, 6, -8, -6, 0), ("gap 1e-9..1e-11, r=10", 120, 10, -11, -9, 0), ("gap 1e-4..1e-6 r=3", 100, 3, -6, -4, 0),
         ("gap + exact zeros", 120, 8, -9, -7, 30), ("two clusters 1e-3 / 1e-9", 120, 5, -9, -3, 0), ("no gap (continuous)", 96, 0, -3, 0, 0),
         ("gap 1e-12..1e-13 (at resolution)", 120, 6, -13, -12, 0)]
for name, n, r, lo, hi, z in cases:
    lam = spectrum(n, r, lo, hi, z)
    X, Pex = make(n, lam)
    nrm = np.abs(X).sum(axis=0).max()
    out = []
    for clean, mega in ((0, 0), (0, 1), (1, 1)):
        P, steps, megas = project(X, clean, mega)
        # the contract: eigenvalues below 1e-13 ||X||_1 may be lost
        err = np.abs(P - Pex).max() / nrm
        e2 = np.linalg.norm(P - Pex, 2) / nrm
        out.append("%2d steps (%d mega) err %.1e |2 %.1e" % (steps, megas, err, e2))
    print("%-34s none: %s | capped: %s | clean: %s" % (name, out[0], out[1], out[2]))
