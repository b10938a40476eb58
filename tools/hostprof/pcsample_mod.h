// HOST-SIDE PROFILING AID: pcsample.h for programs linked against the REAL library -- every sample is resolved with dladdr()
// at dump time: "<module>\t<offset in module>\t<nearest exported symbol>" (the main program's samples go through addr2line).
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>

static unsigned long g_pcs[1 << 20];
static volatile unsigned g_npcs = 0;
static void pcsample_handler(int, siginfo_t *, void *uc) {
  if (g_npcs < (1u << 20)) g_pcs[g_npcs++] = (unsigned long)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
}
static inline void pcsample_start() {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = pcsample_handler;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction(SIGPROF, &sa, nullptr);
  struct itimerval it = {{0, 200}, {0, 200}};
  setitimer(ITIMER_PROF, &it, nullptr);
}
static inline void pcsample_dump(const char *path) {
  struct itimerval it = {{0, 0}, {0, 0}};
  setitimer(ITIMER_PROF, &it, nullptr);
  FILE *f = fopen(path, "w");
  for (unsigned i = 0; i < g_npcs; ++i) {
    Dl_info di;
    if (dladdr((void *)g_pcs[i], &di) && di.dli_fname)
      fprintf(f, "%s\t0x%lx\t%s\n", di.dli_fname, g_pcs[i] - (unsigned long)di.dli_fbase, di.dli_sname ? di.dli_sname : "?");
    else
      fprintf(f, "?\t0x%lx\t?\n", g_pcs[i]);
  }
  fclose(f);
  fprintf(stderr, "pcsample: %u samples -> %s\n", g_npcs, path);
}
