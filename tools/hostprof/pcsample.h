// HOST-SIDE PROFILING AID: a PC sampler for the header's inlined code (gprof sees one big flush()): SIGPROF every 200 us of
// CPU time, the interrupted program counter goes into a table, pcsample_dump() writes the addresses (relative to the
// executable's load base) for `addr2line -f -i -e <binary>` -- see tools/hostprof/run.sh profile.
#pragma once
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#include <link.h>

static unsigned long g_pcs[1 << 20];
static volatile unsigned g_npcs = 0;
static void pcsample_handler(int, siginfo_t *, void *uc) {
  if (g_npcs < (1u << 20)) g_pcs[g_npcs++] = (unsigned long)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
}
static int pcsample_base_cb(struct dl_phdr_info *info, size_t, void *data) {
  if (!*(unsigned long *)data && info->dlpi_name[0] == 0) *(unsigned long *)data = info->dlpi_addr + 1;  // the main program
  return 0;
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
  unsigned long base = 0;
  dl_iterate_phdr(pcsample_base_cb, &base);
  base -= 1;
  FILE *f = fopen(path, "w");
  for (unsigned i = 0; i < g_npcs; ++i) fprintf(f, "0x%lx\n", g_pcs[i] - base);
  fclose(f);
  fprintf(stderr, "pcsample: %u samples -> %s\n", g_npcs, path);
}
