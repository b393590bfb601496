/* LD_PRELOAD helper: print a native backtrace when the process receives SIGABRT (a runtime library aborting without a message).
 * Build: gcc -shared -fPIC -O1 -o build/abort_bt.so tools/diag/abort_bt.c ; run pytest with -p no:faulthandler so this handler stays installed. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_abort(int sig) {
  void* frames[64];
  const char msg[] = "\n[abort_bt] SIGABRT, native backtrace:\n";
  (void)!write(2, msg, sizeof msg - 1);
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void install(void) {
  void* warm[2]; backtrace(warm, 2);          /* loads libgcc now, not inside the handler */
  struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = on_abort;
  sigaction(SIGABRT, &sa, 0);
}
