// LD_PRELOAD helper: print a native backtrace when the process aborts (glibc heap checks at exit).  Debug tool only.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void on_abrt(int sig) {
  void *bt[64];
  int n = backtrace(bt, 64);
  backtrace_symbols_fd(bt, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void init(void) { signal(SIGABRT, on_abrt); }
