/* LD_PRELOAD-able: on SIGSEGV / SIGABRT print the native backtrace (function names of exported symbols) to stderr, then die by the
 * default action.  For scripts/capture_probe.py: where inside libamdhip64 does hipStreamEndCapture crash? */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_signal(int sig) {
    void* frames[64];
    const char msg[] = "\n== native backtrace (segv_trace) ==\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    void* warm[4];
    backtrace(warm, 4);                     /* loads libgcc now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_signal;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_NODEFER | SA_ONSTACK;
    static char stack[1 << 16];
    stack_t ss = {.ss_sp = stack, .ss_size = sizeof(stack), .ss_flags = 0};
    sigaltstack(&ss, NULL);
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
}
