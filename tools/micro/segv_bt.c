// LD_PRELOAD-free native backtrace on SIGSEGV / SIGBUS / SIGABRT (debugging aid: tools/dbg_e2e.py loads it with ctypes and calls
// segv_bt_install() after the imports).  gcc -shared -fPIC -O1 -o build/segv_bt.so tools/micro/segv_bt.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig, siginfo_t* si, void* uc) {
    void* bt[96];
    const int n = backtrace(bt, 96);
    const char msg[] = "=== native backtrace of the faulting thread\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
void segv_bt_install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
}
