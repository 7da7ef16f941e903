/* LD_PRELOAD diagnostic: native backtrace of the thread that raises SIGABRT (glibc's heap checks abort without saying who called free()).
   gcc -O1 -g -shared -fPIC -o libabort_backtrace.so abort_backtrace.c   — used by tools/repro_loop.sh, not by the product */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
static void on_abort(int sig) {
  void* frames[64];
  const char msg[] = "\n==== SIGABRT: native backtrace of the aborting thread ====\n";
  write(2, msg, sizeof(msg) - 1);
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  FILE* f = fopen("/proc/self/maps", "r");
  if (f) {  /* the mappings, to turn addresses into library offsets */
    char line[512];
    while (fgets(line, sizeof line, f)) if (strstr(line, "r-xp")) write(2, line, strlen(line));
    fclose(f);
  }
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void install(void) {
  void* warm[4];
  backtrace(warm, 4); /* loads libgcc now, not inside the handler */
  signal(SIGABRT, on_abort);
}
