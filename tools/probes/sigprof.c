// A sampling profiler for the host side of an un-replayed step (no perf in this image): ITIMER_PROF at a few kHz, the
// handler keeps the call stack's return addresses; sigprof_dump() writes them as library+offset for nm to resolve.
// build: gcc -O2 -g -shared -fPIC -o sigprof.so sigprof.c -ldl      (loaded by tools/direct_issue_profile.py)
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#define MAX_SAMPLES 400000
#define DEPTH 48
static void* g_stacks[MAX_SAMPLES][DEPTH];
static unsigned char g_depth[MAX_SAMPLES];
static volatile int g_n = 0, g_on = 0;

static void on_prof(int sig, siginfo_t* si, void* uc) {
  (void)sig; (void)si; (void)uc;
  if (!g_on) return;
  int i = g_n;
  if (i >= MAX_SAMPLES) return;
  int d = backtrace(g_stacks[i], DEPTH);
  g_depth[i] = (unsigned char)d;
  g_n = i + 1;
}

int sigprof_start(int hz) {
  void* warm[4];
  backtrace(warm, 4);   // (loads libgcc's unwinder outside the handler)
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_prof;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, NULL);
  g_n = 0;
  g_on = 1;
  struct itimerval it;
  it.it_interval.tv_sec = 0; it.it_interval.tv_usec = 1000000 / hz;
  it.it_value = it.it_interval;
  return setitimer(ITIMER_PROF, &it, NULL);
}

int sigprof_stop(void) {
  struct itimerval it;
  memset(&it, 0, sizeof it);
  setitimer(ITIMER_PROF, &it, NULL);
  g_on = 0;
  return g_n;
}

// one line per sample, innermost frame first: `<library path>+<offset in hex>` separated by ';' (frames `skip`.. of the
// stack: 0 and 1 are the handler and the signal trampoline).  tools/direct_issue_profile.py resolves them with nm.
void sigprof_dump(const char* path, int skip) {
  FILE* f = fopen(path, "w");
  if (!f) return;
  for (int s = 0; s < g_n; ++s) {
    for (int d = skip; d < g_depth[s]; ++d) {
      Dl_info di;
      if (dladdr(g_stacks[s][d], &di) && di.dli_fname)
        fprintf(f, "%s+%lx;", di.dli_fname, (unsigned long)((char*)g_stacks[s][d] - (char*)di.dli_fbase));
      else
        fprintf(f, "?+%lx;", (unsigned long)g_stacks[s][d]);
    }
    fputc('\n', f);
  }
  fclose(f);
}
