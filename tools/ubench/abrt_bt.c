/* LD_PRELOAD helper (developer tool): native backtrace on SIGABRT / SIGSEGV, to find which runtime call aborts a test process. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void h(int sig)
{
    void* bt[64];
    int n = backtrace(bt, 64);
    dprintf(2, "\n==== native backtrace on signal %d ====\n", sig);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void init(void)
{
    signal(SIGABRT, h);
    signal(SIGSEGV, h);
}
