// LD_PRELOAD helper for debugging: prints a native backtrace when the process receives SIGABRT or SIGSEGV (e.g. glibc's heap checks at exit,
// after Python's faulthandler is already gone).  gcc -shared -fPIC -o tools/_build/abort_bt.so tools/abort_bt.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig) {
    void *bt[64];
    const char msg[] = "---- SIGABRT / SIGSEGV backtrace ----\n";
    (void) !write(2, msg, sizeof msg - 1);
    int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    void *warm[2];
    backtrace(warm, 2);                       // loads libgcc now, not inside the handler
    static char alt[1 << 16];                 // its own stack: a fault from stack exhaustion still gets here
    stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_abort;
    sa.sa_flags = SA_ONSTACK;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGSEGV, &sa, 0);
}
