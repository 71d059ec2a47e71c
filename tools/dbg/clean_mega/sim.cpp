#include "sign_sched.h"
extern "C" int sim(double* s, int n, int lag, double* err) { return cuadmm::sign_sched_simulate(s, n, lag, err); }
