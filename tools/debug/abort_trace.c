/* LD_PRELOAD helper for the GPU box: prints the NATIVE backtrace of the thread that raised SIGABRT / SIGSEGV / SIGBUS
 * before handing the signal on (Python's faulthandler, which pytest enables, only shows the Python frames of each thread and
 * chains to whatever handler was installed before it — this one).
 *   gcc -shared -fPIC -O1 -o gpurun_out/abort_trace.so tools/debug/abort_trace.c
 *   LD_PRELOAD=$PWD/gpurun_out/abort_trace.so python -m pytest tests -m gpu -q */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_fatal(int sig) {
    static const char head[] = "\n==== abort_trace: native backtrace of the signalling thread ====\n";
    void* frames[96];
    (void)!write(2, head, sizeof(head) - 1);
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    void* warm[2];
    backtrace(warm, 2); /* loads libgcc now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_fatal;
    sa.sa_flags = SA_NODEFER;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}
