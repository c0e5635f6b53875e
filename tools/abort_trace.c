/* abort_trace.c -- LD_PRELOAD shim for debugging: prints the NATIVE backtrace of whoever calls abort()
 * (and of any SIGABRT), so that a silent abort inside a runtime library can be attributed.
 *   gcc -shared -fPIC -O1 -o tools/_bin/libaborttrace.so tools/abort_trace.c -ldl
 *   LD_PRELOAD=tools/_bin/libaborttrace.so python -m pytest ...
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void dump(const char *why)
{
    void *frames[64];
    int n = backtrace(frames, 64);
    const char *hdr = "\n==== abort_trace: ";
    write(2, hdr, strlen(hdr));
    write(2, why, strlen(why));
    write(2, " ====\n", 6);
    backtrace_symbols_fd(frames, n, 2);
    write(2, "==== end ====\n", 14);
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
    signal(SIGABRT, on_abrt);
}
