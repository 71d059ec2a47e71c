// LD_PRELOAD helper (development aid): native backtrace on SIGABRT, before Python's faulthandler gets it.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
static void on_abort(int sig) {
  void* bt[64];
  const int n = backtrace(bt, 64);
  const char msg[] = "\n[abort_bt] native backtrace:\n";
  write(2, msg, sizeof msg - 1);
  backtrace_symbols_fd(bt, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_abort;
  sigaction(SIGABRT, &sa, NULL);
}
