#include "sign_sched.h"
using S = cuadmm::SignSched;
extern "C" {
void* ss_new(int clean, int mega_on) { S* s = new S(); s->clean = clean != 0; s->mega_on = mega_on != 0; return s; }
void ss_free(void* h) { delete (S*)h; }
int ss_cont(void* h) { return ((S*)h)->cont ? 1 : 0; }
int ss_steps(void* h) { return ((S*)h)->steps; }
int ss_megas(void* h) { return ((S*)h)->megas; }
double ss_decide(void* h, int n, double a, double b, double gprev, double* alpha, double* beta, int* half, double* cmc, int* last) {
  S* s = (S*)h;
  s->gprev = gprev;
  bool l = false;
  const double mu = s->decide<true>(n, a, b, 0.0, l);
  s->coefs(mu, *alpha, *beta);
  *half = s->half; *cmc = s->cmc; *last = l ? 1 : 0;
  return mu;
}
}
