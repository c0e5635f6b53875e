/* abort_trace.c -- LD_PRELOAD shim for debugging: records the NATIVE backtrace of whoever calls abort()
 * (and of any SIGABRT) so that an abort inside a runtime library can be attributed.  The trace goes to a
 * FILE opened in the constructor (ABORT_TRACE_FILE, default /tmp/abort_trace.<pid>.log) and to the
 * process's ORIGINAL stderr (duplicated in the constructor): pytest's fd capture replaces fd 2 during a
 * test, and whatever is written there is lost when the process dies.
 *   gcc -shared -fPIC -O1 -o tools/_bin/libaborttrace.so tools/abort_trace.c -ldl
 *   ABORT_TRACE_FILE=gpurun_out/abort.log LD_PRELOAD=tools/_bin/libaborttrace.so python -m pytest ...
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int g_fd = -1, g_err = -1;

static void put(int fd, const char *s)
{
    if (fd >= 0 && write(fd, s, strlen(s)) < 0) { /* nothing to do */ }
}

static void dump(const char *why)
{
    void *frames[64];
    int n = backtrace(frames, 64);
    const int fds[2] = {g_fd, g_err};
    for (int i = 0; i < 2; ++i) {
        if (fds[i] < 0) continue;
        put(fds[i], "\n==== abort_trace: ");
        put(fds[i], why);
        put(fds[i], " ====\n");
        backtrace_symbols_fd(frames, n, fds[i]);
        put(fds[i], "==== end ====\n");
        fsync(fds[i]);
    }
}

void abort(void)
{
    dump("abort() called");
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
    _exit(134);
}

static void on_abrt(int sig)
{
    (void)sig;
    dump("SIGABRT delivered");
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}

__attribute__((constructor)) static void init(void)
{
    char path[512];
    const char *p = getenv("ABORT_TRACE_FILE");
    if (p && *p) snprintf(path, sizeof(path), "%s.%d", p, (int)getpid());
    else snprintf(path, sizeof(path), "/tmp/abort_trace.%d.log", (int)getpid());
    g_fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    g_err = dup(2);
    signal(SIGABRT, on_abrt);
}
