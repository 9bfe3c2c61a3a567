/* Diagnostic aid (tools/thread_dump.py): the NATIVE call stack of any thread of this process, without a debugger.
 * nbt_install() sets a SIGUSR2 handler that prints the interrupted thread's backtrace (module + offset per frame) to
 * stderr; nbt_signal(tid) delivers the signal to that thread.  Built by hand: gcc -shared -fPIC -O1 -o libnative_bt.so native_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static void handler(int sig) {
    void *buf[48];
    char hdr[64];
    (void)sig;
    int n = backtrace(buf, 48);
    int l = snprintf(hdr, sizeof hdr, "== native stack of tid %ld ==\n", (long)syscall(SYS_gettid));
    if (write(2, hdr, (size_t)l) < 0) return;
    backtrace_symbols_fd(buf, n, 2);
}

int nbt_install(void) {
    struct sigaction sa;
    void *warm[2];
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sa.sa_flags = SA_RESTART;
    backtrace(warm, 2); /* loads libgcc's unwinder now, not inside the handler */
    return sigaction(SIGUSR2, &sa, NULL);
}

int nbt_signal(int tid) { return (int)syscall(SYS_tgkill, getpid(), tid, SIGUSR2); }
